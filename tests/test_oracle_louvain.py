"""
CPU suite: the two Louvain oracles on hand-checkable graphs.  python-louvain itself is absent
(PARITY UNPINNED, see oracle/louvain_seq.c), so these pin what CAN be pinned: the obvious partitions,
modularity against networkx's independent `modularity` implementation, exactness of the fixed-point
schedule, and agreement between the sequential and the bulk-synchronous schedule on easy graphs.
"""
import numpy as np
import pytest

import graphs


def _nx_modularity(n, edges, labels, loops=None):
    import networkx as nx
    g = nx.Graph()
    g.add_nodes_from(range(n))
    for u, v, w in edges:
        g.add_edge(u, v, weight=w)
    for u, w in (loops or {}).items():
        g.add_edge(u, u, weight=w)
    comms = {}
    for i, l in enumerate(labels):
        comms.setdefault(int(l), set()).add(i)
    return nx.algorithms.community.modularity(g, list(comms.values()), weight='weight')


def test_two_cliques_sequential(oracle_lib):
    n, edges = graphs.two_cliques()
    ip, ix, w = graphs.csr_from_edges(n, edges)
    r = oracle_lib.louvain_seq(ip, ix, w)
    assert r['n_comm'] == 2
    assert graphs.same_partition(r['labels'], [0] * 5 + [1] * 5)
    assert abs(r['Q'] - _nx_modularity(n, edges, r['labels'])) < 1e-12
    assert abs(oracle_lib.modularity(ip, ix, w, r['labels']) - r['Q']) < 1e-12


@pytest.mark.parametrize('maker', [graphs.ring_of_cliques, graphs.karate])
def test_sequential_modularity_matches_networkx(oracle_lib, maker):
    n, edges = maker()
    ip, ix, w = graphs.csr_from_edges(n, edges)
    r = oracle_lib.louvain_seq(ip, ix, w)
    assert abs(r['Q'] - _nx_modularity(n, edges, r['labels'])) < 1e-12
    assert r['Q'] > 0.35
    assert r['labels'].min() == 0 and r['labels'].max() == r['n_comm'] - 1


def test_ring_of_cliques_multilevel(oracle_lib):
    """30 cliques of 5 on a ring: level 0 finds the cliques, a second level merges neighbours (resolution limit)."""
    n, edges = graphs.ring_of_cliques(30, 5, 1.0, 1.0)
    ip, ix, w = graphs.csr_from_edges(n, edges)
    r = oracle_lib.louvain_seq(ip, ix, w)
    assert r['n_levels'] >= 2 and r['n_comm'] < 30
    # cliques are never split
    lab = r['labels'].reshape(30, 5)
    assert (lab == lab[:, :1]).all()
    S = oracle_lib.quant_shift(ip, ix, w)
    b = oracle_lib.louvain_bsp(ip, ix, oracle_lib.quantise(w, S), n_sub=4)
    lab = b['labels'].reshape(30, 5)
    assert (lab == lab[:, :1]).all() and b['n_levels'] >= 2


def test_self_loops_and_isolated_rows(oracle_lib):
    n, edges = graphs.two_cliques(4, 0.05)
    n += 2                                   # node 8: only a self-loop; node 9: empty row
    ip, ix, w = graphs.csr_from_edges(n, edges, loops={0: 2.0, 8: 1.0})
    r = oracle_lib.louvain_seq(ip, ix, w)
    assert r['labels'][9] == -1              # never added to the graph (cluster.py:321)
    assert r['labels'][8] >= 0 and (r['labels'] == r['labels'][8]).sum() == 1
    assert abs(r['Q'] - _nx_modularity(9, edges, r['labels'][:9], loops={0: 2.0, 8: 1.0})) < 1e-12
    S = oracle_lib.quant_shift(ip, ix, w)
    b = oracle_lib.louvain_bsp(ip, ix, oracle_lib.quantise(w, S))
    assert graphs.same_partition(b['labels'], r['labels'])


def test_no_edges(oracle_lib):
    ip = np.zeros(6, dtype=np.int64)
    r = oracle_lib.louvain_seq(ip, np.zeros(0, np.int32), np.zeros(0))
    assert (r['labels'] == -1).all() and r['n_comm'] == 0
    b = oracle_lib.louvain_bsp(ip, np.zeros(0, np.int32), np.zeros(0, np.int64))
    assert (b['labels'] == -1).all()


def test_quantisation_is_exact_and_bounded(oracle_lib):
    rs = np.random.RandomState(3)
    n, edges = graphs.ring_of_cliques(20, 8)
    edges = [(u, v, w * rs.uniform(1e-6, 1.0)) for u, v, w in edges]
    ip, ix, w = graphs.csr_from_edges(n, edges, loops={3: 0.7})
    S = oracle_lib.quant_shift(ip, ix, w)
    q = oracle_lib.quantise(w, S)
    rows = np.repeat(np.arange(n), np.diff(ip))
    total = int(q.sum()) + int(q[rows == ix].sum())          # 2m with loops counted twice
    assert 2 ** 59 <= total < 2 ** 61                         # uses the headroom, never overflows
    np.testing.assert_allclose(q / 2.0 ** S, w, rtol=0, atol=2.0 ** -S)
    assert np.array_equal(q, np.rint(np.ldexp(w, S)).astype(np.int64))


@pytest.mark.parametrize('n_sub', [1, 2, 4, 8])
def test_bsp_close_to_sequential_on_synthetic_map(oracle_lib, n_sub):
    gr = graphs.synth_graph(oracle_lib, 3000, 90000, 5)
    seq = oracle_lib.louvain_seq(gr['indptr'], gr['indices'], gr['w'])
    S = oracle_lib.quant_shift(gr['indptr'], gr['indices'], gr['w'])
    bsp = oracle_lib.louvain_bsp(gr['indptr'], gr['indices'], oracle_lib.quantise(gr['w'], S), n_sub=n_sub)
    qb = oracle_lib.modularity(gr['indptr'], gr['indices'], gr['w'], bsp['labels'])
    assert abs(qb - bsp['Q']) < 1e-9           # exact-integer Q equals the fp64 Q of the same partition
    assert qb > seq['Q'] - 0.01                # schedule changes the optimum found only marginally
    # planted genomes are recovered by both
    gen = gr['sm'].genome[gr['mask']]
    assert graphs.same_partition(seq['labels'], gen)
    assert graphs.same_partition(bsp['labels'], gen)


# three hard maps: 40-50 % of the edges join random contigs and carry the same counts as the intra-genome ones, so the
# modularity landscape is flat (Q ~ 0.48-0.57), Louvain needs several levels and the schedule matters
HARD_MAPS = [(3000, 90000, 5, 0.4), (6000, 120000, 7, 0.5), (20000, 600000, 9, 0.5)]


@pytest.mark.parametrize('n,e,seed,noise', HARD_MAPS)
def test_bsp_against_sequential_on_hard_maps(oracle_lib, n, e, seed, noise):
    """Where the two schedules CAN differ they do: the bulk-synchronous schedule the device runs and the sequential
    python-louvain-style sweep end in different local optima of comparable modularity.  This pins how far apart: both
    need more than one level or a dozen sweeps, the exact-integer modularity of the BSP result equals the fp64 modularity
    of its partition, and the two optima lie within 0.03 of each other on these maps (reported: ARI between them)."""
    gr = graphs.synth_graph(oracle_lib, n, e, seed, noise=noise, intra_boost=1)
    ip, ix, w = gr['indptr'], gr['indices'], gr['w']
    seq = oracle_lib.louvain_seq(ip, ix, w)
    bsp = oracle_lib.louvain_bsp(ip, ix, oracle_lib.quantise(w, oracle_lib.quant_shift(ip, ix, w)))
    assert 0.3 < seq['Q'] < 0.7 and 0.3 < bsp['Q'] < 0.7
    assert bsp['n_levels'] >= 2 and bsp['n_sweeps'] >= 12
    assert abs(oracle_lib.modularity(ip, ix, w, bsp['labels']) - bsp['Q']) < 1e-9
    assert abs(oracle_lib.modularity(ip, ix, w, seq['labels']) - seq['Q']) < 1e-9
    ari = graphs.adjusted_rand(seq['labels'], bsp['labels'])
    print('hard map n={} noise={}: Q seq {:.4f} / bsp {:.4f}, levels {} / {}, communities {} / {}, ARI {:.3f}'.format(
        n, noise, seq['Q'], bsp['Q'], seq['n_levels'], bsp['n_levels'], seq['n_comm'], bsp['n_comm'], ari))
    assert abs(bsp['Q'] - seq['Q']) < 0.03
    assert ari > 0.5


@pytest.mark.parametrize('maker', [graphs.two_cliques, graphs.ring_of_cliques, graphs.karate])
def test_against_python_louvain(oracle_lib, maker):
    """The pin for `community.best_partition(g)` (mzd/louvain.py:70; python-louvain 0.11 in Pipfile.lock): runs the moment
    the package is installed (it is absent from this image and from the reference tree, hence PARITY UNPINNED elsewhere).
    The graph is built the way cluster.py:316-321 builds it: nx.Graph, add_edge in row-major entry order."""
    com = pytest.importorskip('community')
    import networkx as nx
    n, edges = maker()
    ip, ix, w = graphs.csr_from_edges(n, edges)
    g = nx.Graph()
    rows = np.repeat(np.arange(n), np.diff(ip))
    for u, v, ww in zip(rows.tolist(), ix.tolist(), w.tolist()):
        g.add_edge(u, v, weight=ww)
    part = com.best_partition(g)
    ref = np.array([part[i] for i in range(n)])
    r = oracle_lib.louvain_seq(ip, ix, w)
    assert abs(com.modularity(part, g) - r['Q']) < 1e-12
    assert graphs.same_partition(r['labels'], ref)


def _nx_best_q(ip, ix, w, seeds=5):
    """modularity of the best of `seeds` runs of networkx's louvain_communities (an independent third-party Louvain: seeded
    shuffling, its own gain bookkeeping) on the graph as cluster.py:319-321 builds it"""
    import networkx as nx
    n = ip.shape[0] - 1
    g = nx.Graph()
    g.add_nodes_from(range(n))
    rows = np.repeat(np.arange(n), np.diff(ip))
    for u, v, ww in zip(rows.tolist(), ix.tolist(), w.tolist()):
        if u <= v:
            g.add_edge(u, v, weight=ww)
    g.remove_nodes_from([u for u in range(n) if g.degree(u) == 0])
    best = -1.0
    for sd in range(seeds):
        parts = nx.community.louvain_communities(g, weight='weight', resolution=1.0, threshold=1e-7, seed=sd)
        best = max(best, nx.community.modularity(g, parts, weight='weight'))
    return best


NX_GRAPHS = [('two_cliques', None), ('ring_of_cliques', None), ('karate', None), ('synth', (3000, 90000, 5, 0.1, 5)),
             ('hard', (3000, 90000, 5, 0.4, 1)), ('hard', (6000, 120000, 7, 0.5, 1))]


@pytest.mark.parametrize('kind,args', NX_GRAPHS)
def test_quality_against_networkx_louvain(oracle_lib, kind, args):
    """python-louvain is absent from this image (test_against_python_louvain stays staged for a box that has it); what IS
    obtainable is an independent quality bar: networkx 3.x's `louvain_communities` -- another implementation of the same
    algorithm (seeded shuffling) -- on the suite's graphs.  Both restatements, the sequential python-louvain-style one and
    the bulk-synchronous schedule the device runs, must reach its best-of-5 modularity to within 0.02."""
    pytest.importorskip('networkx')
    if args is None:
        n, edges = getattr(graphs, kind)()
        ip, ix, w = graphs.csr_from_edges(n, edges)
    else:
        nn, e, seed, noise, boost = args
        gr = graphs.synth_graph(oracle_lib, nn, e, seed, noise=noise, intra_boost=boost)
        ip, ix, w = gr['indptr'], gr['indices'], gr['w']
    q_nx = _nx_best_q(ip, ix, w)
    seq = oracle_lib.louvain_seq(ip, ix, w)
    bsp = oracle_lib.louvain_bsp(ip, ix, oracle_lib.quantise(w, oracle_lib.quant_shift(ip, ix, w)))
    print('{} {}: Q networkx {:.4f} / sequential {:.4f} / BSP {:.4f}'.format(kind, args, q_nx, seq['Q'], bsp['Q']))
    assert seq['Q'] > q_nx - 0.02
    assert bsp['Q'] > q_nx - 0.02
