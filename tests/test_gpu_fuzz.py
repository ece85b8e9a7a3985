"""GPU suite: a seeded slice of the randomised differential campaign (tools/fuzz_parity.py): random symmetric count maps
-- sizes 3..4000, 1.5..400 contacts per row, hub rows, empty rows, zero diagonals, 16- and 32-bit counts, random masks --
through KR, the balanced map, compress, the fused sub-map, Louvain, the permutation, max_offdiag and downsample, each
compared with the CPU oracle (bit-exact except KR's x and the float reductions)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('seed,env', [(7, {}), (8, {}), (9, {'B3C_KR_TCD_XT': '1'}), (10, {'B3C_KR_NO_TCD': '1'}),
                                      (11, {'B3C_KR_TCD_XT': '1', 'B3C_KR_HUB_SPLIT': '256', 'B3C_KR_NT_MIN': '0', 'B3C_LV_NT_MIN': '0'})])
def test_random_maps_against_the_oracle(seed, env):
    """env: the tail pass of the dictionary SpMV forced on for every map that takes the dictionary form (it is automatic only
    beyond 2^19 contigs) / the CSR row kernels for all maps / the non-temporal stream policy of maps beyond the caches forced on
    every map (tail words, hub rows in pieces, the Louvain scan's entries)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_parity.py'), '80', str(seed)],
                         env=dict(os.environ, PYTHONPATH=ROOT, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0 and 'fuzz: 80 cases, 0 failures' in text, text[-3000:]
