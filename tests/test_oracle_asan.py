"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (gcc; GPU sanitizers are unavailable on the pool): the
checker itself must not read or write out of bounds while it restates the reference.  A sanitizer build of oracle/*.c
(`make -C oracle asan`; one thread) runs the KR fixtures, the CSR helpers and both Louvain restatements in a child
process with libasan preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, glob, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import oracle
oracle._LIB = ctypes.CDLL(os.path.join(%(root)r, 'oracle', '_build', 'liboracle_asan.so'))
lib = oracle._LIB
lib.orc_max_value.restype = ctypes.c_double
lib.orc_modularity.restype = ctypes.c_double
lib.orc_compress.restype = ctypes.c_int64
lib.orc_quant_shift.restype = ctypes.c_int32
n_kr = 0
for f in sorted(glob.glob(os.path.join(%(root)r, 'tests', 'golden', 'kr_*.npz'))):
    g = np.load(f)
    if 'in_indptr' not in g.files:
        continue
    kw = {}
    for k in ('tol', 'delta', 'Delta', 'max_iter'):
        if k in g.files:
            kw[k] = g[k].item()
    rc, x, st = oracle.kr_balance(g['in_indptr'], g['in_indices'], g['in_data'], **kw)
    if rc == oracle.OK and 'x' in g.files:
        assert np.allclose(x, g['x'], rtol=1e-9), f
    oracle.scale_sym(g['in_indptr'], g['in_indices'], g['in_data'], x)
    n_kr += 1
assert n_kr >= 10, n_kr
# CSR helpers + both Louvain restatements on a random symmetric map with a hub, empty rows and a zero diagonal
rs = np.random.RandomState(5)
n = 700
import scipy.sparse as sp
m = sp.random(n, n, 0.02, random_state=rs, format='coo')
m = sp.coo_matrix((rs.randint(1, 50, m.nnz).astype(np.float64), (m.row, m.col)), shape=(n, n))
m = (m + m.T).tolil()
m[3, :] = 0; m[:, 3] = 0
m[7, :400] = 5; m[:400, 7] = 5
m = m.tocsr(); m.sort_indices(); m.eliminate_zeros()
ip, ix, a = m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64)
cnt = a.astype(np.uint32)
oracle.max_offdiag_u32(ip, ix, cnt)
sites = rs.randint(0, 9, n).astype(np.float64)
nd = oracle.site_norm(ip, ix, a, sites)
mask = rs.rand(n) > 0.2
k, sip, six, sa = oracle.compress(ip, ix, nd, mask)
scl = 1.0 / oracle.max_value(sip, sa)
w = oracle.graph_weights(sip, six, sa, scl)
seq = oracle.louvain_seq(sip, six, w)
q = oracle.quantise(w, oracle.quant_shift(sip, six, w))
bsp = oracle.louvain_bsp(sip, six, q)
assert seq['labels'].shape[0] == k == bsp['labels'].shape[0]
print('asan child ok', n_kr)
'''


def test_oracle_under_address_and_ub_sanitizers():
    asan = None
    for cand in ('/usr/lib/gcc/x86_64-linux-gnu/11/libasan.so', '/usr/lib/x86_64-linux-gnu/libasan.so.6'):
        if os.path.exists(cand):
            asan = cand
            break
    if asan is None:
        out = subprocess.run(['gcc', '-print-file-name=libasan.so'], stdout=subprocess.PIPE).stdout.decode().strip()
        if os.path.isabs(out) and os.path.exists(out):
            asan = out
    if asan is None:
        pytest.skip('no libasan on this box')
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'asan'])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=97',
               UBSAN_OPTIONS='halt_on_error=1:exitcode=98', PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0 and 'asan child ok' in text and 'ERROR: AddressSanitizer' not in text and 'runtime error' not in text, text[-4000:]
