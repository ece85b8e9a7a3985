"""
GPU suite: shapes that steer libb3c onto its less travelled kernels, each against the oracle --
rows long enough for the 4-entries-per-lane SpMV (average > 128) with 16-bit and with 32-bit counts, rows past one
pipelined chunk (> 256) and past the register sort of the permutation (> 512), and degenerate maps.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _counts_case(n, e, seed, scale=1):
    from bin3c_amd.synth import make_synth
    sm = make_synth(n, e, seed)
    counts = (sm.counts.astype(np.uint64) * scale).astype(np.uint32)
    return sm, counts


@pytest.mark.parametrize('n,e,seed,scale,bits', [
    (3000, 300000, 31, 1, 16),       # average row 201: kr_kb_vec<8,8,uint16>
    (3000, 300000, 32, 3000, 32),    # the same with counts beyond 65535: kr_kb_vec<8,8,uint32>
    (1200, 400000, 33, 1, 16),       # average row 667: tail loop of the SpMV, global-sort path of the permutation
])
def test_long_rows_counts_path_matches_oracle(ctx, oracle_lib, n, e, seed, scale, bits):
    from bin3c_amd.pipeline import run_hot_path
    sm, cnt = _counts_case(n, e, seed, scale)
    assert (cnt.max() > 0xFFFF) == (bits == 32)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, cnt.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    assert rc == oracle_lib.OK
    counts = ctx.upload(sm.indptr, sm.indices, cnt)
    for reorder in (False, True):
        # (strict_kr=False: the COUNT form is what this test is about; maps of this size take the fp64 form by default)
        res = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5, reorder=reorder, cluster=False, strict_kr=False)
        assert res.kr_info.value_bits == bits and res.kr_info.n_iter == st.n_iter
        np.testing.assert_allclose(res.x, xo, rtol=1e-9)
    # the permutation itself, rows past the register network included
    perm = np.random.RandomState(seed).permutation(n).astype(np.int32)
    pm = ctx.permute(counts, perm)
    ip, ix, d = pm.download()
    c = sp.csr_matrix((cnt, sm.indices, sm.indptr), shape=(n, n))
    want = c[perm][:, perm]
    want.sort_indices()
    assert np.array_equal(ip, want.indptr) and np.array_equal(ix, want.indices) and np.array_equal(d, want.data)
    pm.free()
    counts.free()


def test_degenerate_maps(ctx):
    """1 x 1, diagonal-only and everything-masked maps"""
    from bin3c_amd.pipeline import run_hot_path
    # 1 x 1: x = 1 / sqrt(a / s^2)
    one = ctx.upload(np.array([0, 1]), np.array([0]), np.array([7], dtype=np.uint32))
    _, x, info = ctx.kr_balance_counts(one, np.array([2.0]))
    np.testing.assert_allclose(x, [1.0 / np.sqrt(7.0 / 4.0)], rtol=1e-5)      # KR stops at its residual tolerance
    assert info.n_zero_diag == 0
    one.free()
    # diagonal only, one zero on the diagonal (treated as one, sparse_utils.py:110-115)
    n = 50
    d = np.arange(1, n + 1, dtype=np.uint32)
    d[7] = 0
    keep = d > 0
    ip = np.concatenate([[0], np.cumsum(keep)]).astype(np.int64)
    diag = ctx.upload(ip, np.flatnonzero(keep).astype(np.int32), d[keep])
    _, x, info = ctx.kr_balance_counts(diag, np.ones(n))
    want = 1.0 / np.sqrt(np.where(d > 0, d, 1).astype(np.float64))
    np.testing.assert_allclose(x, want, rtol=1e-5)
    assert info.n_zero_diag == 1
    # nothing passes the acceptance mask (no off-diagonal signal): empty sub-map, no labels, no crash
    res = run_hot_path(ctx, diag, np.ones(n), np.full(n, 5000), 1000, 5)
    assert res.mask.sum() == 0 and res.labels.shape[0] == 0
    diag.free()


def test_cluster_map_raises_when_everything_is_filtered(ctx, tmp_path):
    from bin3c_amd.mzd.cluster import cluster_map
    from bin3c_amd.mzd.contact_map import ContactMap
    from bin3c_amd.mzd.exceptions import NoneAcceptedException
    n = 30
    m = sp.identity(n, dtype=np.uint32, format='csr') * 9
    cm = ContactMap.from_arrays(m, np.full(n, 5000), np.ones(n), min_len=1000, min_sig=5)
    cm._ctx = ctx
    with pytest.raises(NoneAcceptedException):
        cluster_map(cm, seed=1, method='louvain', work_dir=str(tmp_path))


def test_max_offdiag_integer_dtypes_and_refusal(ctx):
    """sparse_utils.max_offdiag: integer maps go to the device as uint32, anything else is refused (no host path)"""
    from bin3c_amd.mzd import sparse_utils
    rs = np.random.RandomState(4)
    d = rs.poisson(0.3, size=(40, 40))
    d = d + d.T
    want = d.copy()
    np.fill_diagonal(want, 0)
    want = want.max(axis=0)
    for dt in (np.uint32, np.int64, np.int32):
        got = sparse_utils.max_offdiag(sp.lil_matrix(d.astype(dt)), ctx=ctx)
        assert np.array_equal(got, want)
    with pytest.raises(TypeError):
        sparse_utils.max_offdiag(sp.csr_matrix(d.astype(np.float64)), ctx=ctx)
    with pytest.raises(TypeError):
        sparse_utils.max_offdiag(sp.csr_matrix(-d.astype(np.int64)), ctx=ctx)



def test_row_maxima_left_by_the_balance_and_their_invalidation(ctx, oracle_lib):
    """b3c_kr_balance_counts leaves the largest off-diagonal count of every row with the matrix (kr_prep_counts) and
    b3c_max_offdiag_u32(assume_symmetric) answers from it; b3c_mat_device_ptrs hands out writable pointers and must drop that
    by-product, after which the call makes its own pass again -- same answer either way."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(20000, 600000, 13)
    ref = oracle_lib.max_offdiag_u32(sm.indptr, sm.indices, sm.counts)
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    assert np.array_equal(ctx.max_offdiag_u32(cm, assume_symmetric=True), ref)      # own pass
    ctx.kr_balance_counts(cm, sm.sites)
    assert np.array_equal(ctx.max_offdiag_u32(cm, assume_symmetric=True), ref)      # by-product of the balance
    cm.device_ptrs()                                                                # writable pointers: by-products dropped
    assert np.array_equal(ctx.max_offdiag_u32(cm, assume_symmetric=True), ref)      # own pass again
    assert np.array_equal(ctx.max_offdiag_u32(cm, assume_symmetric=False), ref)
    cm.free()


@pytest.mark.parametrize('graph_form', [False, True])
def test_fused_sub_map_is_bit_identical(ctx, graph_form):
    """b3c_normalise_scale_compress == b3c_normalise_scale followed by b3c_compress, on a permuted (id-carrying) matrix"""
    from bin3c_amd.synth import make_synth
    sm = make_synth(4000, 150000, 41)
    sm.sites[::53] = 0.0
    rs = np.random.RandomState(6)
    perm = rs.permutation(sm.n).astype(np.int32)
    counts = ctx.permute(ctx.upload(sm.indptr, sm.indices, sm.counts), perm)
    sites = np.ascontiguousarray(sm.sites[perm])
    x = rs.uniform(0.5, 2.0, sm.n)
    mask = rs.rand(sm.n) < 0.8
    full = ctx.normalise_scale(counts, sites, x, graph_form=graph_form)
    want = ctx.compress(full, mask)
    got = ctx.normalise_scale_compress(counts, sites, x, mask, graph_form=graph_form)
    for a, b in zip(got.download(), want.download()):
        assert np.array_equal(a, b)
    assert np.array_equal(ctx.mat_ids(got), ctx.mat_ids(want))
    for m in (full, want, got, counts):
        m.free()


def test_very_long_hub_row(ctx, oracle_lib):
    """a contig in contact with 200k others (power-law maps have such rows): kr_kb_hub and the long-row sort of the
    permutation, against the oracle"""
    from bin3c_amd.pipeline import run_hot_path
    rs = np.random.RandomState(17)
    n, hub, m = 250000, 123456, 750000
    i = rs.randint(0, n, m)
    j = rs.randint(0, n, m)
    keep = i != j
    others = rs.choice(n, 200000, replace=False)
    others = others[others != hub]
    rows = np.concatenate([i[keep], np.full(others.shape[0], hub)])
    cols = np.concatenate([j[keep], others])
    vals = np.concatenate([rs.geometric(0.3, int(keep.sum())), rs.geometric(0.5, others.shape[0])]).astype(np.uint64)
    a = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    a = (a + a.T + sp.diags(rs.geometric(0.1, n).astype(np.uint64))).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    ip, ix, cnt = a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.astype(np.uint32)
    assert np.diff(ip).max() > 190000
    sites = rs.poisson(20, n).astype(np.float64)
    nd = oracle_lib.site_norm(ip, ix, cnt.astype(np.float64), sites)
    rc, xo, st = oracle_lib.kr_balance(ip, ix, nd)
    assert rc == oracle_lib.OK
    counts = ctx.upload(ip, ix, cnt)
    for reorder in (False, True):
        res = run_hot_path(ctx, counts, sites, np.full(n, 5000), 1000, 1, reorder=reorder, cluster=False)
        assert res.kr_info.n_iter == st.n_iter
        np.testing.assert_allclose(res.x, xo, rtol=1e-8)
    counts.free()


def test_locality_order_shortcut_equals_the_full_gather(ctx):
    """k_best_external fetches cluster labels for the few heaviest entries of a row first and for the rest only when none of
    those leaves the cluster; B3C_ORDER_SAMPLE=-1 forces the label gather for every entry (the loop of rounds 1-2).  Same
    heaviest leaving edge for every row, hence the same order, on flat, power-law and all-ones (every weight tied) maps."""
    import scipy.sparse as sp
    from bin3c_amd.synth import make_synth
    maps = []
    for args in ((60000, 3000000, 4), (40000, 1500000, 5, 'powerlaw')):
        sm = make_synth(*args)
        maps.append((sm.indptr, sm.indices, sm.counts))
    rs = np.random.RandomState(2)
    # (sixty random columns per row; scipy's sp.random draws WITHOUT replacement from all 1.6e9 cells: a minute of host time)
    r_, c_ = np.repeat(np.arange(40000), 60), rs.randint(0, 40000, 40000 * 60)
    a = sp.coo_matrix((np.ones(r_.shape[0]), (r_, c_)), shape=(40000, 40000)).tocsr()
    a = ((a + a.T) > 0).astype(np.uint32).tocsr()
    a.sort_indices()
    maps.append((a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.astype(np.uint32)))
    for ip, ix, cnt in maps:
        m = ctx.upload(ip, ix, cnt)
        os.environ.pop('B3C_ORDER_SAMPLE', None)
        ctx.reload_env()
        fast = ctx.locality_order(m)
        os.environ['B3C_ORDER_SAMPLE'] = '-1'
        ctx.reload_env()
        try:
            full = ctx.locality_order(m)
        finally:
            os.environ.pop('B3C_ORDER_SAMPLE', None)
            ctx.reload_env()
        assert np.array_equal(fast[0], full[0]) and np.array_equal(fast[1], full[1])
        assert np.array_equal(np.sort(fast[0]), np.arange(ip.shape[0] - 1))
        m.free()
