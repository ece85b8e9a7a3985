"""
CPU suite: the host-side mirror of the reference's Python surface (integer / label work must be identical).
Expected values in tests/golden/seqorder_cluster.npz were produced by the reference's own SeqOrder class,
add_cluster_names and the standardisation block of cluster_map (tools/make_golden.py).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN


class _SI(object):
    def __init__(self, length):
        self.length = length


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(GOLDEN, 'seqorder_cluster.npz'))


def _order(g):
    from bin3c_amd.mzd.contact_map import SeqOrder
    o = SeqOrder([_SI(int(v)) for v in g['lengths']])
    o.set_mask_only(g['mask'])
    return o


def test_seqorder_mask_positions_and_remaps(g):
    from bin3c_amd.mzd.contact_map import SeqOrder
    o = _order(g)
    assert o.order.dtype == SeqOrder.STRUCT_TYPE
    assert np.array_equal(o.order['pos'], g['pos_after_mask'])
    assert np.array_equal(o.all_positions(), g['positions'])
    assert np.array_equal(o.gapless_positions(), g['gapless'])
    assert np.array_equal(o.remap_gapless(g['dense']), g['remap_plain'])
    r = o.remap_gapless(SeqOrder.asindex(np.sort(g['dense'])))
    assert r.dtype == SeqOrder.INDEX_TYPE
    assert np.array_equal(r['index'], g['remap_index_index']) and np.array_equal(r['ori'], g['remap_index_ori'])
    assert o.count_accepted() == int(g['mask'].sum()) and o.count_excluded() == int((~g['mask']).sum())
    assert np.array_equal(o.accepted(), np.flatnonzero(g['mask']))
    assert np.array_equal(o.lengths(exclude_masked=True), g['lengths'][g['mask']])


def test_cluster_map_standardisation(g):
    from bin3c_amd.mzd import cluster

    class _CM(object):
        pass

    cm = _CM()
    cm.order = _order(g)
    lab = g['labels']
    cl_to_ids = {int(c): np.flatnonzero(lab == c) for c in np.unique(lab)}
    out = cluster.standardise(cm, cl_to_ids)
    assert len(out) == int(g['n_clusters'])
    for k in out:
        assert np.array_equal(out[k]['seq_ids'], g['cl{}_seq_ids'.format(k)])
        assert int(out[k]['extent']) == int(g['cl{}_extent'.format(k)])
        assert out[k]['name'] == str(g['cl{}_name'.format(k)])
    extents = [out[k]['extent'] for k in sorted(out)]
    assert extents == sorted(extents, reverse=True)


@pytest.mark.parametrize('k', [1, 9, 10, 11, 99, 100, 101, 1000])
def test_cluster_names_width(g, k):
    from bin3c_amd.mzd.cluster import add_cluster_names
    d = {i: {} for i in range(k)}
    add_cluster_names(d)
    assert d[0]['name'] == str(g['names_{}_first'.format(k)])
    assert d[k - 1]['name'] == str(g['names_{}_last'.format(k)])


def test_solver_wire_formats(tmp_path):
    from bin3c_amd.mzd import cluster
    p = tmp_path / 'x.mcl'
    p.write_text('3 1 2\n7 5\n\n9\n')
    m = cluster.read_mcl(str(p))
    assert list(m) == [0, 1] and m[0].tolist() == [1, 2, 3] and m[1].tolist() == [5, 7]
    p = tmp_path / 'x.tab'
    p.write_text('0 4\n1 4\n2 9\n')
    t = cluster.read_table(str(p))
    assert t[4].tolist() == [0, 1] and t[9].tolist() == [2]
    p.write_text('4\n4\n9\n')
    t = cluster.read_table(str(p), seq_col=None, cl_col=0)
    assert t[4].tolist() == [0, 1] and t[9].tolist() == [2]
    p = tmp_path / 'x.tree'
    p.write_text('# comment\n1:1 0.1 "a" 5\n1:2 0.1 "b" 6\n2:1 0.1 "c" 7\n1:3 0.1 "d" 8\n')
    tr = cluster.read_tree(str(p))
    assert tr[0].tolist() == [5, 6, 8] and tr[1].tolist() == [7]          # renumbered by descending size
    ip = np.array([0, 2, 4, 5], dtype=np.int64)
    ix = np.array([0, 1, 0, 2, 1], dtype=np.int32)
    w = np.array([0.5, 0.25, 0.25, 1.0, 1.0])
    f = cluster.write_edges((ip, ix, w), str(tmp_path), 'cm_graph')
    assert open(f).read().split('\n')[:3] == ['0 0 0.5', '0 1 0.25', '1 2 1.0']


def test_write_edges_is_byte_identical_to_networkx(tmp_path):
    """cluster.py:139-151: the file networkx.write_edgelist wrote for the graph of cluster.py:316-321 (bytes captured by
    tools/make_golden2.py), and -- where networkx is installed -- the same comparison live on a second map."""
    from bin3c_amd.mzd import cluster
    import scipy.sparse as sp
    g = np.load(os.path.join(GOLDEN, 'edges_nx.npz'))
    c = sp.csr_matrix((g['data'], g['indices'], g['indptr']))
    # the graph triple as to_graph forms it: the lower triangle mirrored (the later add_edge(v, u) overwrites (u, v))
    low = sp.tril(c, k=0, format='csr')
    sym = (low + sp.tril(c, k=-1, format='csr').T).tocsr()
    sym.sort_indices()
    f = cluster.write_edges((sym.indptr.astype(np.int64), sym.indices.astype(np.int32), sym.data * float(g['scl'])),
                            str(tmp_path), 'cm_graph')
    assert open(f, 'rb').read() == g['edge_file'].tobytes()
    nx = pytest.importorskip('networkx')
    rs = np.random.RandomState(3)
    a = sp.random(70, 70, density=0.08, random_state=rs, format='coo')
    a = (a + a.T + sp.diags(rs.uniform(0, 1, 70) * (rs.uniform(0, 1, 70) < 0.5))).tocsr()
    a.eliminate_zeros()
    a.sort_indices()
    coo = a.tocoo()
    gr = nx.Graph()
    for u, v, w in zip(coo.row, coo.col, coo.data):
        gr.add_edge(u, v, weight=w)
    ref = tmp_path / 'ref.edges'
    nx.write_edgelist(gr, str(ref), data=['weight'], delimiter='\t')
    f = cluster.write_edges((a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data), str(tmp_path), 'mine', sep='\t')
    assert open(f, 'rb').read() == ref.read_bytes()


def test_object_io_roundtrip_and_suffixes(tmp_path):
    from bin3c_amd.mzd import io_utils
    obj = {'a': np.arange(5), 'b': 'x'}
    io_utils.save_object(str(tmp_path / 'thing.p'), obj)
    assert (tmp_path / 'thing.p.gz').exists()                               # .gz appended (io_utils.py:73-76)
    back = io_utils.load_object(str(tmp_path / 'thing.p.gz'))
    assert back['b'] == 'x' and np.array_equal(back['a'], obj['a'])
    import bz2
    import pickle
    with bz2.BZ2File(str(tmp_path / 'o.bz2'), 'wb') as h:
        pickle.dump([1, 2], h)
    assert io_utils.load_object(str(tmp_path / 'o.bz2')) == [1, 2]


def test_exception_family_and_utils(tmp_path):
    from bin3c_amd.mzd import exceptions as ex, utils
    assert issubclass(ex.NoneAcceptedException, ex.ApplicationException)
    assert ex.NoneAcceptedException().message == 'all sequences were excluded'
    assert ex.ZeroLengthException('c1').message == 'Sequence [c1] has zero length'
    assert 1000000 <= utils.make_random_seed() < 10000000
    d = tmp_path / 'o'
    utils.make_dir(str(d))
    with pytest.raises(IOError):
        utils.make_dir(str(d))
    utils.make_dir(str(d), exist_ok=True)


def test_contact_map_surface_pickles_without_device(tmp_path):
    """The attribute surface of contact_map.py:492-518 survives a save/load cycle (no GPU needed for that)."""
    from bin3c_amd.mzd.contact_map import ContactMap
    from bin3c_amd.mzd import io_utils
    from bin3c_amd.synth import make_synth
    sm = make_synth(300, 6000, 3)
    cm = ContactMap.from_arrays(sm.to_scipy(), sm.lengths, sm.sites)
    for attr in ('strong bam_file bin_size min_mapq min_insert min_len min_sig min_extent min_size max_fold '
                 'random_state seq_info seq_map seq_file grouping extent_map order tip_size precount total_reads '
                 'cov_info processed_map primary_acceptance_mask bisto_scale seq_analyzer enzymes total_len '
                 'total_seq current_mask').split():
        assert hasattr(cm, attr), attr
    assert cm.seq_map.dtype == np.uint32 and cm.total_seq == 300 and not cm.is_empty() and not cm.is_tipbased()
    io_utils.save_object(str(tmp_path / 'contact_map.p'), cm)
    back = io_utils.load_object(str(tmp_path / 'contact_map.p.gz'))
    assert (back.seq_map != cm.seq_map).nnz == 0 and back.processed_map is None
    assert np.array_equal(back.order.order, cm.order.order)


def test_reference_module_paths_resolve_on_load(tmp_path):
    """A pickle that names the reference's classes (mzd.contact_map.*) loads onto the mirror."""
    import gzip
    import pickle
    import pickletools
    from bin3c_amd.mzd import io_utils
    from bin3c_amd.mzd.contact_map import ContactMap, SeqOrder
    from bin3c_amd.synth import make_synth
    sm = make_synth(200, 3000, 4)
    cm = ContactMap.from_arrays(sm.to_scipy(), sm.lengths, sm.sites)
    raw = pickle.dumps(cm, protocol=2)
    assert b'bin3c_amd.mzd.contact_map' in raw
    # rewrite the module path the way the reference would have written it
    raw = raw.replace(b'cbin3c_amd.mzd.contact_map\n', b'cmzd.contact_map\n')
    assert b'cmzd.contact_map\n' in raw and b'bin3c_amd.mzd.contact_map' not in raw
    with gzip.open(str(tmp_path / 'ref_map.p.gz'), 'wb') as h:
        h.write(raw)
    back = io_utils.load_object(str(tmp_path / 'ref_map.p.gz'))
    assert isinstance(back, ContactMap) and isinstance(back.order, SeqOrder)
    assert back.total_seq == 200 and (back.seq_map != cm.seq_map).nnz == 0


def test_loads_a_python2_protocol0_contact_map(tmp_path):
    """a ContactMap in the byte format the Python-2 reference writes (cPickle protocol 0: old-style class INST
    records, S'...' strings with raw numpy buffers, copy_reg._reconstructor for the SeqInfo namedtuples and for the scipy
    matrices of scipy < 1.8, numpy.random.__RandomState_ctor) -- assembled opcode by opcode by tools/make_py2_pickle.py
    because no Python 2 exists here -- loads into the mirror classes with its content intact and round-trips."""
    import warnings
    import scipy.sparse as sp
    from bin3c_amd.mzd import io_utils
    from bin3c_amd.mzd.contact_map import ContactMap, SeqInfo, SeqOrder
    import gzip
    raw = gzip.open(os.path.join(GOLDEN, 'contact_map_py2.p.gz')).read()
    assert raw.startswith(b'(imzd.contact_map\nContactMap\n') and raw.endswith(b'.')          # protocol 0, old-style instance
    assert b'ccopy_reg\n_reconstructor\n' in raw and b'cnumpy.core.multiarray\n_reconstruct\n' in raw
    exp = np.load(os.path.join(GOLDEN, 'contact_map_py2_expect.npz'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                    # numpy.core / scipy.sparse.coo shims announce their deprecation
        cm = io_utils.load_object(os.path.join(GOLDEN, 'contact_map_py2.p.gz'))
    assert isinstance(cm, ContactMap) and isinstance(cm.order, SeqOrder) and isinstance(cm.seq_info[0], SeqInfo)
    n = exp['lengths'].shape[0]
    assert (cm.total_seq, cm.min_len, cm.min_sig, cm.enzymes, cm.bam_file) == (n, 1000, 5, ['MluCI', 'Sau3AI'], 'hic2ctg.bam')
    assert [s.length for s in cm.seq_info] == exp['lengths'].tolist() and [s.sites for s in cm.seq_info] == exp['sites'].astype(int).tolist()
    assert sp.isspmatrix_coo(cm.seq_map) and cm.seq_map.dtype == np.uint32
    c = cm.seq_map.tocsr()
    c.sort_indices()
    assert np.array_equal(c.indptr, exp['indptr']) and np.array_equal(c.indices, exp['indices']) and np.array_equal(c.data, exp['counts'])
    assert cm.order.order.dtype == SeqOrder.STRUCT_TYPE and np.array_equal(cm.order.order['length'], exp['lengths'])
    assert np.array_equal(cm.order.lengths(), exp['lengths']) and cm.order.count_accepted() == n
    assert np.array_equal(cm.primary_acceptance_mask, exp['mask']) and cm.processed_map is None
    assert np.array_equal(cm.random_state.randint(0, 1 << 30, 5), exp['rs_next'])        # the generator resumes where it was
    io_utils.save_object(str(tmp_path / 'again.p'), cm)
    back = io_utils.load_object(str(tmp_path / 'again.p.gz'))
    assert back.total_seq == n and (back.seq_map != cm.seq_map).nnz == 0
