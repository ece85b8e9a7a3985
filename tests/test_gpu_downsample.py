"""
Heat-map step after clustering: sparse_utils.downsample (sparse_utils.py:48-87),
ContactMap._reorder_seq (contact_map.py:1066-1085), the raster of ContactMap.plot (:1316-1357) and
enable_clusters / plot_clusters (cluster.py:327-449), device path against numpy/scipy restatements.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from bin3c_amd.mzd import sparse_utils
from bin3c_amd.mzd.cluster import cluster_map, enable_clusters, plot_clusters
from bin3c_amd.mzd.contact_map import ContactMap
from bin3c_amd.synth import make_synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def oracle(oracle_lib):
    return oracle_lib


def _rand(rows, cols, density, seed, signed=False):
    """random sparse matrix from directly drawn coordinates (scipy.sparse.random with a legacy RandomState
    permutes range(rows * cols): hundreds of GB at the sizes used below)"""
    rs = np.random.RandomState(seed)
    k = int(round(rows * cols * density))
    m = sp.coo_matrix((rs.gamma(2.0, 3.0, k), (rs.randint(0, rows, k), rs.randint(0, cols, k))), shape=(rows, cols)).tocsr()
    m.sum_duplicates()
    m.sort_indices()
    if signed:
        m.data -= 4.0
    return m


@pytest.mark.parametrize('n,b', [(37, 2), (37, 3), (40, 5), (9, 9), (9, 20), (64, 64), (130, 7), (130, 8), (500, 16)])
@pytest.mark.parametrize('method', ['mean', 'max'])
def test_downsample_small(ctx, oracle, n, b, method):
    m = _rand(n, n, 0.2, n * 31 + b, signed=(method == 'max'))
    got = sparse_utils.downsample(m, b, method, ctx=ctx)
    exp = oracle.downsample_dense(m, b, method)
    assert got.shape == exp.shape and sp.isspmatrix_csr(got)
    if method == 'max':
        np.testing.assert_array_equal(got.toarray(), exp)
    else:
        np.testing.assert_allclose(got.toarray(), exp, rtol=1e-14, atol=0)
    assert got.has_sorted_indices


def test_downsample_full_blocks_max_negative(ctx):
    """a block whose b*b cells are all stored keeps its (negative) maximum; one missing cell brings in the zero"""
    d = -np.arange(1.0, 17.0).reshape(4, 4)
    got = sparse_utils.downsample(sp.csr_matrix(d), 2, 'max', ctx=ctx).toarray()
    np.testing.assert_array_equal(got, [[-1.0, -3.0], [-9.0, -11.0]])
    d[0, 0] = 0.0
    m = sp.csr_matrix(d)
    m.eliminate_zeros()
    got = sparse_utils.downsample(m, 2, 'max', ctx=ctx).toarray()
    np.testing.assert_array_equal(got, [[0.0, -3.0], [-9.0, -11.0]])


def test_downsample_ndarray_nonsquare_and_dtype(ctx, oracle):
    rs = np.random.RandomState(5)
    d = rs.poisson(0.4, size=(23, 50)).astype(np.uint32)
    for method in ('mean', 'max'):
        got = sparse_utils.downsample(d, 4, method, ctx=ctx)
        exp = oracle.downsample_dense(d, 4, method)
        assert got.shape == (6, 13)
        np.testing.assert_allclose(got.toarray(), exp, rtol=1e-14)
    assert sparse_utils.downsample(d, 4, 'max', ctx=ctx).dtype == np.uint32


def test_downsample_empty_and_asserts(ctx):
    got = sparse_utils.downsample(sp.csr_matrix((10, 10), dtype=np.float64), 3, ctx=ctx)
    assert got.shape == (4, 4) and got.nnz == 0
    with pytest.raises(AssertionError):
        sparse_utils.downsample(sp.eye(4, format='csr'), 1, ctx=ctx)
    with pytest.raises(AssertionError):
        sparse_utils.downsample(sp.eye(4, format='csr'), 2.0, ctx=ctx)
    with pytest.raises(AssertionError):
        sparse_utils.downsample([[1.0]], 2, ctx=ctx)


def _coo_reduce(m, b):
    c = m.tocoo()
    nb = -(-m.shape[0] // b)
    r = sp.coo_matrix((c.data, (c.row // b, c.col // b)), shape=(nb, nb)).tocsr()
    r.sum_duplicates()
    return r


@pytest.mark.parametrize('n,b,density', [(20000, 7, 2e-3), (300000, 2, 3e-5), (50000, 250, 1e-3)])
def test_downsample_large_sparse(ctx, n, b, density):
    """sizes past one bitmap chunk (n/b > 131072) and plot-like reduction factors, against a scipy COO reduction"""
    m = _rand(n, n, density, n + b)
    got = sparse_utils.downsample(m, b, 'mean', ctx=ctx)
    exp = _coo_reduce(m, b) * (1.0 / b ** 2)
    assert got.nnz == exp.nnz
    np.testing.assert_array_equal(got.indptr, exp.indptr)
    np.testing.assert_array_equal(got.indices, exp.indices)
    np.testing.assert_allclose(got.data, exp.data, rtol=1e-13)
    again = sparse_utils.downsample(m, b, 'mean', ctx=ctx)
    np.testing.assert_array_equal(got.data, again.data)          # fixed summation order


def _small_map(ctx):
    sm = make_synth(1500, 60000, 11)
    cm = ContactMap.from_arrays(sm.to_scipy(), sm.lengths, sm.sites, min_len=1000, min_sig=5, min_extent=20000)
    cm._ctx = ctx
    cm.set_primary_acceptance_mask()
    return cm


def test_reorder_seq_matches_scipy(ctx):
    cm = _small_map(ctx)
    cm.prepare_seq_map(norm=True, bisto=True)
    sub = cm.get_subspace(permute=False)
    rs = np.random.RandomState(2)
    acc = cm.order.accepted()
    cm.order.set_order_only(rs.permutation(acc), implicit_excl=True)
    order = cm.order.gapless_positions()
    got = cm._reorder_seq(sub)
    exp = sub.tocsr()[order][:, order]
    exp.sort_indices()
    np.testing.assert_array_equal(got.indptr, exp.indptr)
    np.testing.assert_array_equal(got.indices, exp.indices)
    np.testing.assert_array_equal(got.data, exp.data)


def test_plot_clusters_raster(ctx, oracle, tmp_path):
    cm = _small_map(ctx)
    clustering = cluster_map(cm, seed=1, method='louvain', work_dir=str(tmp_path))
    captured = {}

    def fake_plot(fname, **kw):
        captured['fname'] = fname
        captured['kw'] = kw
        captured['img'], captured['ticks'] = cm.plot_matrix(simple=kw['simple'], permute=kw['permute'],
                                                            max_image_size=kw['max_image_size'], tick_locs=kw['tick_locs'])
    cm.plot = fake_plot
    plot_clusters(cm, str(tmp_path / 'cluster_plot.png'), clustering, max_image_size=100, ordered_only=False,
                  simple=False, permute=True)
    cl_list = sorted(k for k in clustering if clustering[k]['extent'] >= cm.min_extent)
    members = np.hstack([clustering[k]['seq_ids'] for k in cl_list])
    # restatement: rows/cols of the processed map in cluster order, x100, zero diagonal, block mean, log(x + alpha)
    full = cm.processed_map.tocsr()[members][:, members].toarray()
    np.fill_diagonal(full, 0)
    full *= 100
    rf = int(np.ceil(len(members) / 100.0))
    exp = np.log(oracle.downsample_dense(full, rf, 'mean') + 0.01)
    np.testing.assert_allclose(captured['img'], exp, rtol=1e-12, atol=1e-12)
    ticks = np.cumsum([0] + [len(clustering[k]['seq_ids']) for k in cl_list])
    np.testing.assert_array_equal(captured['ticks'], np.floor(ticks / float(rf)))
    assert captured['kw']['tick_labs'] == [clustering[k]['name'] for k in cl_list]
    assert cm.order.count_accepted() == len(members)


def test_plot_writes_file(ctx, tmp_path):
    cm = _small_map(ctx)
    clustering = cluster_map(cm, seed=1, method='louvain', work_dir=str(tmp_path))
    out = tmp_path / 'cluster_plot.png'
    plot_clusters(cm, str(out), clustering, max_image_size=200, ordered_only=False, simple=False, permute=True,
                  width=4, height=4, dpi=50)
    assert out.exists() or (tmp_path / 'cluster_plot.png.npy').exists()


def test_downsample_skip_diagonal(ctx, oracle):
    """B3C_DS_SKIP_DIAG == setdiag(0) before the reduction (contact_map.py:1324), both kernel forms"""
    for n, b in ((130, 3), (130, 16), (700, 50)):
        m = _rand(n, n, 0.15, n + b)
        m = (m + sp.diags(np.arange(1.0, n + 1.0))).tocsr()
        z = m.tolil()
        z.setdiag(0)
        z = z.tocsr()
        mat = ctx.upload(m.indptr, m.indices, m.data)
        for method in ('mean', 'max'):
            ds = ctx.downsample(mat, b, method, skip_diag=True)
            ip, ix, d = ds.download()
            got = sp.csr_matrix((d, ix, ip), shape=(ds.n, ds.n)).toarray()
            np.testing.assert_allclose(got, oracle.downsample_dense(z, b, method), rtol=1e-14, atol=0)
            ds.free()
        mat.free()


def test_raster_device_path_equals_host_path(ctx):
    """plot_matrix from the device-resident balanced map against the host route through get_subspace/downsample"""
    cm = _small_map(ctx)
    cm.prepare_seq_map(norm=True, bisto=True)
    rs = np.random.RandomState(3)
    cm.order.set_order_only(rs.permutation(cm.order.accepted()), implicit_excl=True)
    ticks = np.array([0, 100, 700])
    for mis in (None, 90):
        img_d, t_d = cm.plot_matrix(permute=True, max_image_size=mis, tick_locs=ticks)
        hot, cm._hot = cm._hot, None                 # force the host route on the same (materialised) map
        cm._processed_host = None
        from bin3c_amd.pipeline import processed_to_host
        ip, ix, data = processed_to_host(ctx, hot)
        cm._processed_host = sp.csr_matrix((data, ix, ip), shape=(cm.total_seq, cm.total_seq))
        img_h, t_h = cm.plot_matrix(permute=True, max_image_size=mis, tick_locs=ticks)
        cm._hot = hot
        cm._processed_host = None
        np.testing.assert_allclose(img_d, img_h, rtol=1e-11, atol=1e-11)
        assert (t_d is None and t_h is None) or np.array_equal(t_d, t_h)
