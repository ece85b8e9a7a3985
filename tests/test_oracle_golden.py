"""
CPU suite: pin the C oracle against the golden vectors that the reference's own function text
produced (tools/make_golden.py).  KR: identical iteration counts, x within 1e-12 relative.
Integer/byte helpers: bit-exact.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

KR_CASES = sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, 'kr_*.npz')))


@pytest.mark.parametrize('case', KR_CASES)
def test_kr_oracle_matches_reference(oracle_lib, case):
    g = np.load(os.path.join(GOLDEN, 'kr_{}.npz'.format(case)))
    rc, x, st = oracle_lib.kr_balance(g['in_indptr'], g['in_indices'], g['in_data'], tol=float(g['tol']),
                                      max_iter=int(g['max_iter']))
    err = str(g['error'])
    assert st.n_zero_diag > 0 if int(g['warn_zero_diag']) else st.n_zero_diag == 0
    assert bool(st.is_symmetric) == (not int(g['warn_asym']))
    if err:
        assert rc == oracle_lib.E_NOCONV
        assert err == 'matrix balancing failed to converge in {} iterations'.format(st.n_iter)
        return
    assert rc == oracle_lib.OK
    assert st.n_iter == int(g['n_iter'])
    assert bool(st.hit_max_iter) == bool(int(g['warn_maxiter']))
    np.testing.assert_allclose(x, g['x'], rtol=1e-12, atol=0)
    out = oracle_lib.scale_sym(g['in_indptr'], g['in_indices'], g['in_data'], g['x'])
    if out.shape == g['out_data'].shape:
        assert np.array_equal(out, g['out_data'])   # same x in -> bit-identical X A X


def test_kr_bound_cases_hit_their_branches(oracle_lib):
    g = np.load(os.path.join(GOLDEN, 'kr_delta_bound.npz'))
    _, _, st = oracle_lib.kr_balance(g['in_indptr'], g['in_indices'], g['in_data'])
    assert st.hit_delta > 0
    g = np.load(os.path.join(GOLDEN, 'kr_bigDelta_bound.npz'))
    _, _, st = oracle_lib.kr_balance(g['in_indptr'], g['in_indices'], g['in_data'])
    assert st.hit_Delta > 0


def test_max_offdiag_bit_exact(oracle_lib):
    g = np.load(os.path.join(GOLDEN, 'max_offdiag.npz'))
    assert np.array_equal(oracle_lib.max_offdiag_u32(g['indptr'], g['indices'], g['data']), g['out'])
    assert np.array_equal(oracle_lib.max_offdiag_u32(g['indptr2'], g['indices2'], g['data2']), g['out2'])


@pytest.mark.parametrize('name', ['random', 'all_true', 'one_false', 'few_true'])
def test_compress_bit_exact(oracle_lib, name):
    g = np.load(os.path.join(GOLDEN, 'compress.npz'))
    m, ip, ix, a = oracle_lib.compress(g['indptr'], g['indices'], g['data'], g['mask_' + name])
    assert m == int(g['out_n_' + name])
    assert np.array_equal(ip, g['out_indptr_' + name])
    assert np.array_equal(ix, g['out_indices_' + name])
    assert np.array_equal(a, g['out_data_' + name])


def test_cluster_pipeline_stages_bit_exact(oracle_lib):
    """mask -> site norm -> KR -> X A X -> compress -> scl -> graph weights (cluster.py:301-321)."""
    g = np.load(os.path.join(GOLDEN, 'pipeline.npz'))
    sig = oracle_lib.max_offdiag_u32(g['indptr'], g['indices'], g['counts'])
    assert np.array_equal(sig, g['signal'])
    mask = (g['lengths'] >= int(g['min_len'])) & (sig >= int(g['min_sig']))
    assert np.array_equal(mask, g['mask'])
    nd = oracle_lib.site_norm(g['indptr'], g['indices'], g['counts'].astype(np.float64), g['sites'])
    assert np.array_equal(nd, g['normed_data'])
    rc, x, st = oracle_lib.kr_balance(g['indptr'], g['indices'], nd)
    np.testing.assert_allclose(x, g['x'], rtol=1e-12)
    proc = oracle_lib.scale_sym(g['indptr'], g['indices'], nd, g['x'])
    assert np.array_equal(proc, g['proc_data'])
    m, ip, ix, a = oracle_lib.compress(g['proc_indptr'], g['proc_indices'], g['proc_data'], g['mask'])
    assert np.array_equal(ip, g['sub_indptr']) and np.array_equal(ix, g['sub_indices']) and np.array_equal(a, g['sub_data'])
    scl = 1.0 / oracle_lib.max_value(ip, a)
    assert scl == float(g['scl'])
    w = oracle_lib.graph_weights(ip, ix, a, scl)
    rows = np.repeat(np.arange(m), np.diff(ip))
    up = rows <= ix
    o = np.lexsort((ix[up], rows[up]))
    assert np.array_equal(rows[up][o], g['graph_u'])
    assert np.array_equal(ix[up][o], g['graph_v'])
    assert np.array_equal(w[up][o], g['graph_w'])
    # both directions of an edge carry the same weight (nx.Graph holds one value per edge)
    import scipy.sparse as sp
    W = sp.csr_matrix((w, ix, ip), shape=(m, m))
    assert (W != W.T).nnz == 0


def test_downsample_restatement_known_answers(oracle_lib):
    oracle = oracle_lib
    """oracle.downsample_dense (sparse_utils.py:48-87) on a case worked by hand: 3x3 -> padded 4x4 -> 2x2"""
    m = np.array([[1.0, 2.0, 0.0], [3.0, 4.0, 5.0], [0.0, 6.0, -7.0]])
    np.testing.assert_array_equal(oracle.downsample_dense(m, 2, 'mean'), [[10 / 4.0, 5 / 4.0], [6 / 4.0, -7 / 4.0]])
    np.testing.assert_array_equal(oracle.downsample_dense(m, 2, 'max'), [[4.0, 5.0], [6.0, 0.0]])
    assert oracle.downsample_dense(np.ones((5, 7)), 3, 'mean').shape == (2, 3)
