"""
GPU suite: BASELINE.json's full sizes through size-independent properties (the oracle is too slow to run there):
balanced rows sum to one, the planted genomes are recovered, results do not depend on the HBM layout, and two runs
are bit-identical.  C2 = 100k contigs / 5M edges; C3 = 1M contigs / 100M edges (about 30 s including generation).
"""
import numpy as np
import pytest
import scipy.sparse as sp

import graphs

pytestmark = pytest.mark.gpu


def _run(ctx, cfg, reorder=True):
    from bin3c_amd.pipeline import run_hot_path
    from bin3c_amd.synth import CONFIGS, make_edges
    n, e, seed, law = CONFIGS[cfg]
    lo, hi, c, diag, lengths, sites, genome = make_edges(n, e, seed, law)
    counts, n_edges = ctx.coo_symmetrise(n, lo, hi, c, diag)
    res = run_hot_path(ctx, counts, sites, lengths, 1000, 5, reorder=reorder)
    return counts, n_edges, sites, lengths, genome, res


@pytest.mark.parametrize('cfg', ['C2', 'C3'])
def test_full_size_properties(ctx, cfg):
    counts, n_edges, sites, lengths, genome, res = _run(ctx, cfg)
    n = counts.n
    assert abs(n_edges - {'C2': 5000000, 'C3': 100000000}[cfg]) < 0.002 * n_edges
    # (1) KR: diag(x) A' diag(x) is doubly stochastic, A' = A with a unit diagonal where A's is zero
    ip, ix, data = counts.download()
    s = np.where(sites == 0, 1.0, sites)
    c = sp.csr_matrix((data.astype(np.float64), ix, ip), shape=(n, n))
    d = (c.diagonal() == 0).astype(np.float64)
    u = res.x / s
    rows = res.x * ((c @ u) / s) + d * res.x * res.x
    assert np.abs(rows - 1.0).max() < 1e-5, np.abs(rows - 1.0).max()
    assert res.kr_info.bitwise_symmetric == 1 and res.kr_info.n_zero_diag == int(d.sum())
    # (2) acceptance mask as the reference defines it
    sig = np.zeros(n, dtype=np.uint32)
    off = c.copy()
    off.setdiag(0)
    sig = np.asarray(off.max(axis=0).todense()).ravel().astype(np.uint32)
    assert np.array_equal(res.signal, sig)
    assert np.array_equal(res.mask, (lengths >= 1000) & (sig >= 5))
    # (3) Louvain: every accepted contig labelled, planted genomes recovered, modularity high
    assert res.labels.min() >= 0 and res.labels.shape[0] == int(res.mask.sum())
    assert graphs.same_partition(res.labels, genome[res.mask])
    assert res.lv_info.modularity > 0.9
    # (4) determinism and layout independence
    from bin3c_amd.pipeline import run_hot_path
    again = run_hot_path(ctx, counts, sites, lengths, 1000, 5, keep=True)
    assert np.array_equal(again.x, res.x) and np.array_equal(again.labels, res.labels)
    # (5) heat-map block reduction of the balanced map: mass is conserved, the maximum survives
    b = 250 if cfg == 'C3' else 25
    ds_max = ctx.downsample(again.processed, b, 'max')
    assert ctx.max_value(ds_max) == ctx.max_value(again.processed)
    ds_max.free()
    ds = ctx.downsample(again.processed, b, 'mean')
    assert ds.n == -(-n // b)
    if cfg == 'C2':
        total = again.processed.download(structure=False)[2].sum()
        np.testing.assert_allclose(ds.download(structure=False)[2].sum() * b * b, total, rtol=1e-11)
    ds.free()
    again.free()
    if cfg == 'C2':
        plain = run_hot_path(ctx, counts, sites, lengths, 1000, 5, reorder=False)
        assert plain.kr_info.n_iter == res.kr_info.n_iter
        np.testing.assert_allclose(plain.x, res.x, rtol=1e-10)
        assert graphs.same_partition(plain.labels, res.labels)
    counts.free()
