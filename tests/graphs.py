"""Small hand-checkable graphs and pipeline helpers shared by the Louvain tests (CSR, symmetric, fp64)."""
# genome-size laws of the synthetic generator: the contract workload (the default) and the variant of rounds 1-4
LAWS = ('survey', 'sqrt')
import numpy as np
import scipy.sparse as sp


def csr_from_edges(n, edges, loops=None):
    """edges: iterable of (u, v, w) with u != v, stored in both directions; loops: {u: w}."""
    r, c, w = [], [], []
    for u, v, ww in edges:
        r += [u, v]
        c += [v, u]
        w += [ww, ww]
    for u, ww in (loops or {}).items():
        r.append(u)
        c.append(u)
        w.append(ww)
    m = sp.coo_matrix((w, (r, c)), shape=(n, n)).tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64)


def two_cliques(k=5, bridge=0.1):
    edges = []
    for base in (0, k):
        for i in range(k):
            for j in range(i + 1, k):
                edges.append((base + i, base + j, 1.0))
    edges.append((k - 1, k, bridge))
    return 2 * k, edges


def ring_of_cliques(n_cliques=12, k=6, w_in=1.0, w_ring=0.3):
    edges = []
    for c in range(n_cliques):
        base = c * k
        for i in range(k):
            for j in range(i + 1, k):
                edges.append((base + i, base + j, w_in * (1.0 + 0.01 * ((i * 7 + j * 3 + c) % 5))))
        nxt = ((c + 1) % n_cliques) * k
        edges.append((base + k - 1, nxt, w_ring))
    return n_cliques * k, edges


def karate():
    import networkx as nx
    g = nx.karate_club_graph()
    edges = [(u, v, float(d.get('weight', 1.0))) for u, v, d in g.edges(data=True)]
    return g.number_of_nodes(), edges


def same_partition(a, b):
    """labelings equal up to renaming (and -1 in the same places)"""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a < 0, b < 0):
        return False
    keep = a >= 0
    pairs = set(zip(a[keep].tolist(), b[keep].tolist()))
    return len(pairs) == len(set(p[0] for p in pairs)) == len(set(p[1] for p in pairs))


def synth_graph(oracle, n, n_edges, seed, min_len=1000, min_sig=5, degree_law='flat', noise=0.1, intra_boost=5, genome_law=None):
    """Synthetic contact map pushed through the oracle's normalise/KR/compress steps -> graph CSR + weights."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(n, n_edges, seed, degree_law, noise, intra_boost, genome_law=genome_law)
    sig = oracle.max_offdiag_u32(sm.indptr, sm.indices, sm.counts)
    mask = (sm.lengths >= min_len) & (sig >= min_sig)
    nd = oracle.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, x, st = oracle.kr_balance(sm.indptr, sm.indices, nd)
    proc = oracle.scale_sym(sm.indptr, sm.indices, nd, x)
    m, ip, ix, a = oracle.compress(sm.indptr, sm.indices, proc, mask)
    scl = 1.0 / oracle.max_value(ip, a)
    return dict(sm=sm, mask=mask, n=m, indptr=ip, indices=ix, data=a, scl=scl,
                w=oracle.graph_weights(ip, ix, a, scl))


def adjusted_rand(a, b):
    """Adjusted Rand index of two labelings (entries < 0 in either are left out)."""
    a = np.asarray(a)
    b = np.asarray(b)
    keep = (a >= 0) & (b >= 0)
    a, b = a[keep], b[keep]
    _, ai = np.unique(a, return_inverse=True)
    _, bi = np.unique(b, return_inverse=True)
    m = sp.coo_matrix((np.ones(a.shape[0]), (ai, bi))).tocsr()
    comb = lambda x: x * (x - 1.0) / 2.0
    sum_ij = comb(m.data).sum()
    sa = comb(np.asarray(m.sum(axis=1)).ravel()).sum()
    sb = comb(np.asarray(m.sum(axis=0)).ravel()).sum()
    tot = comb(float(a.shape[0]))
    exp = sa * sb / tot
    mx = 0.5 * (sa + sb)
    return 1.0 if mx == exp else (sum_ij - exp) / (mx - exp)
