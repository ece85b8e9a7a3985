"""
Host-side mirror of the reference's ``mzd/louvain.py``: ``cluster(g, no_iso, method, ragbag, verbose)`` with the
same return shape ``{community: {node: weight}}`` (louvain.py:36-124).  ``community.best_partition``
(python-louvain, louvain.py:70) is replaced by libb3c's Louvain on the GPU; the graph may be a
networkx.Graph (the reference's type) or the CSR triple of mzd.cluster.to_graph.
"""
import logging

import numpy as np
import scipy.sparse as sp

from .. import _native

logger = logging.getLogger(__name__)


def _as_csr(g):
    """-> (nodes, indptr, indices, weights); nodes maps row -> node label"""
    if isinstance(g, tuple):
        ip, ix, w = g
        return list(range(ip.shape[0] - 1)), np.asarray(ip), np.asarray(ix), np.asarray(w, dtype=np.float64)
    nodes = list(g.nodes())
    pos = {u: i for i, u in enumerate(nodes)}
    r, c, w = [], [], []
    for u, v, d in g.edges(data=True):
        ww = float(d.get('weight', 1))
        r.append(pos[u])
        c.append(pos[v])
        w.append(ww)
        if u != v:
            r.append(pos[v])
            c.append(pos[u])
            w.append(ww)
    m = sp.coo_matrix((w, (r, c)), shape=(len(nodes), len(nodes))).tocsr()
    m.sort_indices()
    return nodes, m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data


def best_partition(g, ctx=None, n_sub=0):
    """node -> community, the role of community.best_partition (louvain.py:21,70)."""
    nodes, ip, ix, w = _as_csr(g)
    ctx = ctx or _native.default_context()
    mat = ctx.upload(ip, ix, w)
    try:
        labels, info = ctx.louvain(mat, 1.0, n_sub=n_sub)
    finally:
        mat.free()
    return {nodes[i]: int(l) for i, l in enumerate(labels) if l >= 0}


def cluster(g, no_iso, method=None, ragbag=False, verbose=False, ctx=None):
    """louvain.py:36-124 (soft-membership options 'maxaff'/'simple' included)."""
    assert not (no_iso and ragbag), 'options no_iso and ragbag are mutually exclusive'
    nodes, ip, ix, w = _as_csr(g)
    n = len(nodes)
    rows = np.repeat(np.arange(n), np.diff(ip))
    offdiag = rows != ix
    has_nb = np.zeros(n, dtype=bool)
    has_nb[rows[offdiag]] = True
    in_graph = np.diff(ip) > 0
    singletons = np.flatnonzero(in_graph & ~has_nb)     # isolated once self-loops are ignored (louvain.py:44-48)
    ragbag_group = None
    keep = np.ones(n, dtype=bool)
    if no_iso:
        logger.info('Removed {0} isolated nodes from graph'.format(len(singletons)))
        keep[singletons] = False
    elif ragbag:
        if len(singletons) == 0:
            logger.info('Ragbag cluster would be empty, so excluded')
            ragbag_group = {}
        else:
            logger.info('Ragbag cluster will contain {0} nodes'.format(len(singletons)))
            keep[singletons] = False
            ragbag_group = dict((nodes[i], 1.0) for i in singletons)
    m = sp.csr_matrix((w, ix, ip), shape=(n, n))
    if not keep.all():
        m = m[keep][:, keep].tocsr()
        m.sort_indices()
    kept = np.flatnonzero(keep)

    logger.info('Determining best partitioning')
    ctx = ctx or _native.default_context()
    mat = ctx.upload(m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64))
    try:
        labels, info = ctx.louvain(mat, 1.0)
    finally:
        mat.free()
    partitions = {nodes[kept[i]]: int(l) for i, l in enumerate(labels) if l >= 0}
    com_ids = set(partitions.values())
    logger.info('There were {0} communities in decomposition'.format(len(com_ids)))

    communities = {}
    for ni, ci in partitions.items():
        communities.setdefault(ci, {})[ni] = 1.0
    if verbose:
        logger.info('Communities with more than one node:')
        for ci in com_ids:
            if len(communities[ci]) > 1:
                logger.info('\tcommunity {0}: {1} nodes'.format(ci, len(communities[ci])))

    if method in ('maxaff', 'simple'):
        coo = m.tocoo()
        lab = np.asarray(labels)
        rowmax = np.zeros(m.shape[0])
        np.maximum.at(rowmax, coo.row, coo.data)
        for u, v, wv in zip(coo.row.tolist(), coo.col.tolist(), coo.data.tolist()):
            if u == v or lab[u] == lab[v] or lab[u] < 0 or lab[v] < 0:
                continue
            nu, nv = nodes[kept[u]], nodes[kept[v]]
            if method == 'simple':
                communities[int(lab[u])][nv] = 0.5
                communities[int(lab[v])][nu] = 0.5
            else:
                if wv == rowmax[u]:
                    communities[int(lab[v])][nu] = 0.5
                if wv == rowmax[v]:
                    communities[int(lab[u])][nv] = 0.5

    if ragbag and len(ragbag_group) > 0:
        rb_id = max(communities) + 1 if communities else 0
        communities[rb_id] = ragbag_group
    return communities
