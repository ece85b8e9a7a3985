"""
GPU suite: the round-2 golden vectors (tools/make_golden2.py, outputs of the reference's own function text) and the
round-1 helper fixtures through the C ABI entry points that had only been reached by fuzzing:
b3c_coo_symmetrise vs Sparse2DAccumulator.get_coo, b3c_mat_permute vs ContactMap._reorder_seq, b3c_kr_balance on an input
that turns the scale vector into NaNs, b3c_max_offdiag_u32 (both assume_symmetric values) and plain b3c_compress.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_coo_symmetrise_matches_the_reference_accumulator(ctx):
    """Sparse2DAccumulator.get_coo (sparse_utils.py:246-266): the ingest counts every observed pair (lo <= hi) once; self
    pairs stay on the diagonal.  The device takes the off-diagonal pairs as a stream and the diagonal as a vector."""
    g = np.load(os.path.join(GOLDEN, 'accumulate.npz'))
    n, lo, hi = int(g['n']), g['lo'], g['hi']
    own = lo == hi
    diag = np.bincount(lo[own], minlength=n).astype(np.uint32)
    mat, n_edges = ctx.coo_symmetrise(n, lo[~own], hi[~own], np.ones(int((~own).sum()), dtype=np.uint32), diag)
    ip, ix, data = mat.download()
    assert str(g['out_dtype']) == 'uint32' and data.dtype == np.uint32
    assert np.array_equal(ip, g['out_indptr']) and np.array_equal(ix, g['out_indices']) and np.array_equal(data, g['out_data'])
    rows = np.repeat(np.arange(n), np.diff(ip))
    assert n_edges == int(np.count_nonzero(rows < ix))
    mat.free()


def test_permute_matches_reference_reorder_seq(ctx):
    """ContactMap._reorder_seq (contact_map.py:1066-1085): P A P^T with P[i, order[i]] = 1, i.e. row k of the result is
    row order[k] of A -- b3c_mat_permute with new2old = order."""
    g = np.load(os.path.join(GOLDEN, 'reorder_seq.npz'))
    a = ctx.upload(g['indptr'], g['indices'], g['data'])
    pm = ctx.permute(a, g['order'].astype(np.int32))
    ip, ix, data = pm.download()
    assert np.array_equal(ip, g['out_indptr']) and np.array_equal(ix, g['out_indices'])
    assert np.array_equal(data, g['out_data'])                       # a permutation moves values, it does not round them
    pm.free()
    a.free()


def test_kr_on_an_input_that_turns_nan_behaves_like_the_reference(ctx):
    """sparse_utils.py:192-193 can never raise: once x holds a NaN, v = x * m.dot(x), rk and rout are NaN as well, `rout > rt`
    (:146) is False and the outer loop ends before the inner loop could test x again.  The reference therefore RETURNS a
    NaN scale vector without raising or warning; so does the device (same iteration count, same NaN pattern), and
    B3C_E_NAN stays reserved for the check should it ever fire."""
    g = np.load(os.path.join(GOLDEN, 'nan_input_kr.npz'))
    assert str(g['error']) == ''
    rc, x, info = ctx.kr_balance(ctx.upload(g['in_indptr'], g['in_indices'], g['in_data']), check=False)
    assert rc == 0 and info.n_iter == int(g['n_iter'])
    assert np.array_equal(np.isnan(x), np.isnan(g['x']))
    ok = ~np.isnan(g['x'])
    np.testing.assert_allclose(x[ok], g['x'][ok], rtol=1e-9)


def test_max_offdiag_fixture_both_paths(ctx):
    """sparse_utils.py:269-281 on the reference's own outputs: the general column-maximum kernel and the row kernel that a
    symmetric map allows, including rows whose only entry is the diagonal (maximum 0)."""
    g = np.load(os.path.join(GOLDEN, 'max_offdiag.npz'))
    for sfx in ('', '2'):
        m = ctx.upload(g['indptr' + sfx], g['indices' + sfx], g['data' + sfx].astype(np.uint32))
        want = np.asarray(g['out' + sfx]).ravel().astype(np.uint32)
        assert np.array_equal(ctx.max_offdiag_u32(m, assume_symmetric=False), want)
        assert np.array_equal(ctx.max_offdiag_u32(m, assume_symmetric=True), want)
        m.free()
    assert (np.asarray(g['out2']).ravel()[[5, 77, 300]] == 0).all()


@pytest.mark.parametrize('which', ['random', 'all_true', 'one_false', 'few_true'])
def test_compress_fixture(ctx, which):
    """sparse_utils.py:284-314 on the reference's own outputs, through plain b3c_compress."""
    g = np.load(os.path.join(GOLDEN, 'compress.npz'))
    a = ctx.upload(g['indptr'], g['indices'], g['data'])
    sub = ctx.compress(a, g['mask_' + which])
    ip, ix, data = sub.download()
    assert sub.n == int(g['out_n_' + which])
    assert np.array_equal(ip, g['out_indptr_' + which]) and np.array_equal(ix, g['out_indices_' + which])
    assert np.array_equal(data, g['out_data_' + which])
    sub.free()
    a.free()


def test_acceptance_mask_and_vec_permute(ctx):
    """b3c_acceptance_mask (contact_map.py:888-905) against its definition on the pipeline fixture's signal, and b3c_vec_permute
    (gather / scatter of 1-, 4- and 8-byte per-contig vectors) against numpy, host and device pointers."""
    import torch
    rs = np.random.RandomState(3)
    n = 5000
    lengths = rs.randint(200, 20000, n).astype(np.int64)
    signal = rs.randint(0, 40, n).astype(np.uint32)
    signal[::97] = 0xFFFFFFF0                                  # beyond int32: the comparison is unsigned
    for min_len, min_sig in ((1000, 5), (0, 0), (10 ** 9, 1)):
        want = ((lengths >= min_len) & (signal.astype(np.int64) >= min_sig)).astype(np.uint8)
        assert np.array_equal(ctx.acceptance_mask(lengths, signal, min_len, min_sig), want)
        dev = torch.empty(n, dtype=torch.uint8, device='cuda')
        ctx.acceptance_mask(torch.as_tensor(lengths, device='cuda'), torch.as_tensor(signal.view(np.int32), device='cuda'), min_len, min_sig, out=dev)
        ctx.sync()
        assert np.array_equal(dev.cpu().numpy(), want)
    perm = rs.permutation(n).astype(np.int32)
    for dt in (np.uint8, np.int32, np.float64):
        src = (rs.rand(n) * 200).astype(dt)
        assert np.array_equal(ctx.vec_permute(src, perm, np.empty_like(src)), src[perm])
        out = np.empty_like(src)
        out[perm] = src
        assert np.array_equal(ctx.vec_permute(src, perm, np.empty_like(src), scatter=True), out)
    t = torch.as_tensor(rs.rand(n), device='cuda')
    got = ctx.vec_permute(t, torch.as_tensor(perm, device='cuda'), torch.empty_like(t))
    ctx.sync()
    assert torch.equal(got, t[torch.as_tensor(perm, device='cuda').long()])
