#!/usr/bin/env python3
"""
Round-2 golden vectors, again produced by EXECUTING the reference's own function text (see tools/make_golden.py for the
slicing rules and shims) or, for the edge list, by the library call the reference makes (networkx.write_edgelist).
Build container only (needs /root/reference).  Adds, without touching the round-1 fixtures:

  accumulate.npz     Sparse2DAccumulator.get_coo (sparse_utils.py:227-266) on a random pair stream
  reorder_seq.npz    ContactMap._reorder_seq (contact_map.py:1066-1085): P A P^T for a gapless order
  nan_input_kr.npz   kr_biostochastic on an input that drives the scale vector to NaN: what the reference RETURNS
                     (sparse_utils.py:192-193 never fires: a NaN in x makes v, rk and rout NaN, and `rout > rt` is then
                     False, so the outer loop ends first)
  edges_nx.npz       the bytes networkx.write_edgelist(g, f, data=['weight'], delimiter=' ') writes (cluster.py:139-151)
                     for the graph cluster.py:316-321 builds from a small balanced sub-map

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden2.py
"""
import os
import sys
import tempfile

import numpy as np
import scipy.sparse as scisp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as mg  # noqa: E402

from bin3c_amd.synth import make_synth  # noqa: E402


def main():
    ns, cap = mg.build_namespace()
    su = mg._read('mzd/sparse_utils.py')
    cm = mg._read('mzd/contact_map.py')

    # ---------------- Sparse2DAccumulator.get_coo ----------------
    np.int = int                                   # the class text asserts isinstance(value, (int, np.int)) (numpy < 1.24)
    exec(mg.slice_block(su, r'class Sparse2DAccumulator'), ns)
    rs = np.random.RandomState(31)
    n = 400
    acc = ns['Sparse2DAccumulator'](n)
    m = 6000
    a = rs.randint(0, n, m)
    b = rs.randint(0, n, m)
    for i, j in zip(a.tolist(), b.tolist()):
        # the ingest orders every pair (contact_map.py:786-792: ix1 <= ix2) and counts it once per observation
        lo, hi = (i, j) if i <= j else (j, i)
        acc[lo, hi] = acc[lo, hi] + 1
    coo = acc.get_coo()
    c = coo.tocsr()
    c.sum_duplicates()
    c.sort_indices()
    lo = np.minimum(a, b).astype(np.int64)
    hi = np.maximum(a, b).astype(np.int64)
    mg.save('accumulate', n=n, lo=lo, hi=hi, out_indptr=c.indptr.astype(np.int64), out_indices=c.indices.astype(np.int32),
            out_data=c.data.astype(np.uint32), out_dtype=str(coo.dtype))

    # ---------------- _reorder_seq ----------------
    # the method body with `self` reduced to what it touches: order.gapless_positions() and is_tipbased()
    txt = mg.slice_block(cm, r'class ContactMap')
    lines = txt.split('\n')
    s0 = next(k for k, ln in enumerate(lines) if ln.startswith('    def _reorder_seq('))
    s1 = next(k for k in range(s0 + 1, len(lines)) if lines[k].startswith('    def '))
    body = '\n'.join(ln[4:] for ln in lines[s0:s1])
    exec(body, ns)

    class _Order(object):
        def __init__(self, o):
            self.o = o

        def gapless_positions(self):
            return self.o

    class _Self(object):
        pass

    sm = make_synth(300, 4000, 41, genome_law='sqrt')
    amat = sm.to_scipy(np.float64)
    order = np.random.RandomState(42).permutation(300)
    me = _Self()
    me.order = _Order(order)
    me.is_tipbased = lambda: False
    out = ns['_reorder_seq'](me, amat).tocsr()
    out.sort_indices()
    mg.save('reorder_seq', indptr=amat.indptr.astype(np.int64), indices=amat.indices.astype(np.int32), data=amat.data,
            order=order.astype(np.int64), out_indptr=out.indptr.astype(np.int64), out_indices=out.indices.astype(np.int32),
            out_data=out.data)

    # ---------------- KR on an input that produces NaNs ----------------
    # row 0 sums to zero (a negative contact): v_0 = 0, Z_0 = rk_0 / v_0 = inf, and the first CG step poisons everything
    mm = mg.sym_random(60, 0.2, 77).tolil()
    mm[0, :] = 0
    mm[:, 0] = 0
    mm[0, 0] = 1.0
    mm[0, 1] = -1.0
    mm[1, 0] = -1.0
    mm = mm.tocsr()
    mm.sort_indices()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = mg.run_kr(ns, cap, mm, {})
    print('nan_input_kr: n_iter={} err={!r} nan in x: {}'.format(res['n_iter'], res['error'], int(np.isnan(res['x']).sum())))
    mg.save('nan_input_kr', in_indptr=mm.indptr.astype(np.int64), in_indices=mm.indices.astype(np.int32), in_data=mm.data,
            x=res['x'], n_iter=res['n_iter'], error=res['error'])

    # ---------------- edge list as networkx writes it ----------------
    import networkx as nx
    g0 = np.load(os.path.join(mg.OUT, 'pipeline.npz'))
    sub = scisp.csr_matrix((g0['sub_data'], g0['sub_indices'], g0['sub_indptr']))
    sub = sub[:120][:, :120].tocoo()                       # a corner of the balanced sub-map, entries in row-major order
    scl = 1.0 / sub.max()
    g = nx.Graph(name='contact_graph')
    for u, v, w in zip(sub.row, sub.col, sub.data):        # cluster.py:319-321
        g.add_edge(u, v, weight=w * scl)
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, 'cm_graph.edges')
        nx.write_edgelist(g, f, data=['weight'], delimiter=' ')   # cluster.py:150
        raw = open(f, 'rb').read()
    c = sub.tocsr()
    c.sort_indices()
    mg.save('edges_nx', indptr=c.indptr.astype(np.int64), indices=c.indices.astype(np.int32), data=c.data, scl=scl,
            edge_file=np.frombuffer(raw, dtype=np.uint8), networkx=nx.__version__)


if __name__ == '__main__':
    main()
