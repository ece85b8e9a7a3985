"""
GPU suite: the reference's Python call surface end to end on the device -- ContactMap hot methods,
cluster_map, louvain.cluster and the bin3C.py cluster command.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import graphs
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _cm_from_fixture(g):
    from bin3c_amd.mzd.contact_map import ContactMap
    n = g['indptr'].shape[0] - 1
    seq_map = sp.csr_matrix((g['counts'], g['indices'], g['indptr']), shape=(n, n))
    return ContactMap.from_arrays(seq_map, g['lengths'], g['sites'], min_len=int(g['min_len']), min_sig=int(g['min_sig']))


def test_contact_map_hot_methods_match_reference_fixture(ctx):
    g = np.load(os.path.join(GOLDEN, 'pipeline.npz'))
    cm = _cm_from_fixture(g)
    cm._ctx = ctx
    mask = cm.set_primary_acceptance_mask()
    assert np.array_equal(mask, g['mask'])
    cm.prepare_seq_map(norm=True, bisto=True)
    np.testing.assert_allclose(cm.bisto_scale, g['x'], rtol=1e-9)
    pm = cm.processed_map
    assert sp.isspmatrix_csr(pm)
    assert np.array_equal(pm.indptr, g['proc_indptr']) and np.array_equal(pm.indices, g['proc_indices'])
    np.testing.assert_allclose(pm.data, g['proc_data'], rtol=1e-9)
    sub = cm.get_subspace(marginalise=True, flatten=False).tocsr()
    sub.sort_indices()
    assert np.array_equal(sub.indptr, g['sub_indptr']) and np.array_equal(sub.indices, g['sub_indices'])
    np.testing.assert_allclose(sub.data, g['sub_data'], rtol=1e-9)
    cm._ctx = None


def test_cluster_map_louvain_and_to_graph(ctx, oracle_lib, tmp_path):
    from bin3c_amd.mzd import cluster
    g = np.load(os.path.join(GOLDEN, 'pipeline.npz'))
    cm = _cm_from_fixture(g)
    cm._ctx = ctx
    clustering = cluster.cluster_map(cm, seed=1, method='louvain', work_dir=str(tmp_path))
    # expectation: the oracle's schedule on the fixture's graph, standardised with the same integer rules
    w = oracle_lib.graph_weights(g['sub_indptr'], g['sub_indices'], g['sub_data'], float(g['scl']))
    ref = oracle_lib.louvain_bsp(g['sub_indptr'], g['sub_indices'],
                                 oracle_lib.quantise(w, oracle_lib.quant_shift(g['sub_indptr'], g['sub_indices'], w)))
    gapped = np.flatnonzero(g['mask'])
    got = {frozenset(v['seq_ids'].tolist()) for v in clustering.values()}
    want = {frozenset(gapped[ref['labels'] == c].tolist()) for c in range(ref['n_comm'])}
    assert got == want
    ext = [clustering[k]['extent'] for k in sorted(clustering)]
    assert ext == sorted(ext, reverse=True) and clustering[0]['name'].startswith('CL')
    for v in clustering.values():
        assert v['extent'] == g['lengths'][v['seq_ids']].sum() and np.all(np.diff(v['seq_ids']) > 0)
    # to_graph: same weights as the reference's nx.Graph would hold
    ip, ix, ww = cluster.to_graph(cm, norm=True, bisto=True, scale=True)
    rows = np.repeat(np.arange(ip.shape[0] - 1), np.diff(ip))
    up = rows <= ix
    o = np.lexsort((ix[up], rows[up]))
    assert np.array_equal(rows[up][o], g['graph_u']) and np.array_equal(ix[up][o], g['graph_v'])
    np.testing.assert_allclose(ww[up][o], g['graph_w'], rtol=1e-9)
    cluster.write_mcl(cm, str(tmp_path / 'clustering.mcl'), clustering)
    lines = open(str(tmp_path / 'clustering.mcl')).read().strip().split('\n')
    assert len(lines) == len(clustering) + int((~g['mask']).sum())          # lost sequences become singletons
    assert sum(len(l.split()) for l in lines) == cm.total_seq
    cm._ctx = None


def test_louvain_cluster_on_networkx_graph(ctx):
    import networkx as nx
    from bin3c_amd.mzd import louvain
    n, edges = graphs.two_cliques(5, 0.1)
    gx = nx.Graph()
    for u, v, w in edges:
        gx.add_edge('n%d' % u, 'n%d' % v, weight=w)
    gx.add_edge('iso', 'iso', weight=1.0)                                    # isolated apart from a self-loop
    com = louvain.cluster(gx, no_iso=False, ragbag=False, ctx=ctx)
    groups = sorted(sorted(v) for v in com.values())
    assert groups == [['iso'], ['n%d' % i for i in range(5)], ['n%d' % i for i in range(5, 10)]]
    assert all(w == 1.0 for v in com.values() for w in v.values())
    com = louvain.cluster(gx, no_iso=True, ctx=ctx)
    assert sorted(len(v) for v in com.values()) == [5, 5]
    com = louvain.cluster(gx, no_iso=False, ragbag=True, ctx=ctx)
    assert {'iso': 1.0} in com.values() and len(com) == 3
    soft = louvain.cluster(gx, no_iso=True, method='simple', ctx=ctx)
    assert any(0.5 in v.values() for v in soft.values())                    # bridge endpoints join both sides
    with pytest.raises(AssertionError):
        louvain.cluster(gx, no_iso=True, ragbag=True, ctx=ctx)


def test_cli_mksynth_then_cluster(tmp_path):
    """config #1 of BASELINE.json: 1k contigs / 50k edges through the cluster command."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    d1, d2 = str(tmp_path / 'map'), str(tmp_path / 'out')
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'bin3C.py'), 'mksynth', '-n', '1000', '-e', '50000', '-s', '1', d1],
                          env=env)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'bin3C.py'), 'cluster', '-s', '7', os.path.join(d1, 'contact_map.p.gz'), d2],
                          env=env)
    from bin3c_amd.mzd.io_utils import load_object
    from bin3c_amd.synth import make_synth
    clustering = load_object(os.path.join(d2, 'clustering.p.gz'))
    sm = make_synth(1000, 50000, 1)
    lab = np.full(1000, -1)
    for k, v in clustering.items():
        lab[v['seq_ids']] = k
    acc = lab >= 0
    assert acc.sum() > 900 and graphs.same_partition(lab[acc], sm.genome[acc])
    assert os.path.exists(os.path.join(d2, 'clustering.mcl')) and os.path.exists(os.path.join(d2, 'bin3C.log'))
    assert os.path.exists(os.path.join(d2, 'cluster_plot.png')) or os.path.exists(os.path.join(d2, 'cluster_plot.png.npy'))
    log = open(os.path.join(d2, 'bin3C.log')).read()
    assert 'Clustering contact graph using method: louvain' in log and 'User set random seed: 7' in log


def test_cluster_map_on_prepared_and_on_reloaded_maps(ctx, tmp_path):
    """Three routes to the same clustering: fresh map; map prepared first (balanced map stays in HBM); map pickled after
    preparation and reloaded (balanced map comes back from the host cache, contact_map.py:941-945 / cluster.py:308-309)."""
    from bin3c_amd.mzd import cluster, io_utils
    g = np.load(os.path.join(GOLDEN, 'pipeline.npz'))

    def as_sets(cl):
        return {frozenset(v['seq_ids'].tolist()) for v in cl.values()}

    cm1 = _cm_from_fixture(g)
    cm1._ctx = ctx
    fresh = cluster.cluster_map(cm1, seed=1, method='louvain', work_dir=str(tmp_path))

    cm2 = _cm_from_fixture(g)
    cm2._ctx = ctx
    cm2.set_primary_acceptance_mask()
    cm2.prepare_seq_map(norm=True, bisto=True)
    assert cm2._hot is not None and cm2._processed_host is None            # still only in HBM
    prepared = cluster.cluster_map(cm2, seed=1, method='louvain', work_dir=str(tmp_path))
    assert as_sets(prepared) == as_sets(fresh)

    cm2._ctx = None
    io_utils.save_object(str(tmp_path / 'prepared.p'), cm2)               # materialises processed_map on the host
    cm3 = io_utils.load_object(str(tmp_path / 'prepared.p.gz'))
    assert cm3.processed_map is not None and cm3._hot is None
    np.testing.assert_allclose(cm3.bisto_scale, g['x'], rtol=1e-9)
    cm3._ctx = ctx
    reloaded = cluster.cluster_map(cm3, seed=1, method='louvain', work_dir=str(tmp_path))
    assert as_sets(reloaded) == as_sets(fresh)
    # a stricter mask on the same prepared map re-uses the balanced matrix (no second KR)
    stricter = cluster.cluster_map(cm3, seed=1, method='louvain', min_len=4000, min_sig=int(g['min_sig']),
                                   work_dir=str(tmp_path))
    kept = np.concatenate([v['seq_ids'] for v in stricter.values()])
    assert np.all(g['lengths'][kept] >= 4000)
    for c in (cm1, cm2, cm3):
        c._ctx = None
