#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by EXECUTING the reference's own function text.

Runs only in the build container (needs /root/reference); the GPU box never sees the reference.
The reference modules are Python 2 and cannot be imported (print statements, cPickle, numba, pysam),
so individual functions are sliced out of the source files as text and exec'd in a namespace
holding numpy/scipy and the smallest possible shims:

  * ``xrange``/``zip``          -> py3 equivalents (loop syntax only)
  * ``is_hermitian``            -> sparse test of the same predicate; the original (sparse_utils.py:10-18)
                                   densifies n x n and uses ``.H`` which scipy >= 1.8 removed.  Warn-only.
  * ``@jit(...)`` decorator     -> dropped (numba absent); the loop body below it runs as plain Python.

Nothing from the reference is written into the repo: only inputs and the outputs it computed.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import builtins
import io
import logging
import os
import re
import sys

import numpy as np
import scipy.sparse as scisp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)

from bin3c_amd.synth import make_synth  # noqa: E402


# ----------------------------------------------------------------------------------------------
# reference text slicing
# ----------------------------------------------------------------------------------------------
def _read(rel):
    with open(os.path.join(REF, rel), 'r') as fh:
        return fh.read().split('\n')


def slice_block(lines, header_regex):
    """Return the text of the top-level def/class whose header matches, up to the next top-level stmt."""
    start = None
    for n, ln in enumerate(lines):
        if re.match(header_regex, ln):
            start = n
            break
    assert start is not None, header_regex
    end = len(lines)
    for n in range(start + 1, len(lines)):
        ln = lines[n]
        if ln and not ln[0].isspace() and not ln.startswith('#') and not ln.startswith(')'):
            end = n
            break
    return '\n'.join(lines[start:end])


class _Capture(logging.Handler):
    def __init__(self):
        logging.Handler.__init__(self)
        self.records = []

    def emit(self, record):
        self.records.append((record.levelname, record.getMessage()))


def build_namespace():
    cap = _Capture()
    logger = logging.getLogger('golden.ref')
    logger.setLevel(logging.DEBUG)
    logger.handlers = [cap]
    logger.propagate = False

    def is_hermitian(m, tol=1e-6):
        # same predicate as sparse_utils.py:18, evaluated sparsely
        d = abs(m - m.T.conj())
        return (d >= tol).nnz == 0

    ns = {'np': np, 'scisp': scisp, 'sp': scisp, 'logger': logger, 'xrange': range,
          'zip': lambda *a: list(builtins.zip(*a)), 'is_hermitian': is_hermitian}
    su = _read('mzd/sparse_utils.py')
    for hdr in (r'def kr_biostochastic\(', r'def max_offdiag\(', r'def compress\('):
        exec(slice_block(su, hdr), ns)
    cm = _read('mzd/contact_map.py')
    txt = slice_block(cm, r'def fast_norm_fullseq_bysite\(')
    exec(txt, ns)
    exec(slice_block(cm, r'class SeqOrder'), ns)
    cl = _read('mzd/cluster.py')
    exec(slice_block(cl, r'def add_cluster_names\('), ns)
    # the standardisation block of cluster_map (cluster.py:205-226) lives inside the function body
    s0 = next(n for n, ln in enumerate(cl) if ln.strip().startswith('clustering = {}'))
    s1 = next(n for n, ln in enumerate(cl) if ln.strip().startswith('add_cluster_names(clustering)'))
    block = [ln[4:] for ln in cl[s0:s1 + 1]]
    ns['_standardise_src'] = '\n'.join(block).replace('.iteritems()', '.items()')
    return ns, cap


# ----------------------------------------------------------------------------------------------
# case builders (inputs are ours; outputs are the reference's)
# ----------------------------------------------------------------------------------------------
def sym_random(n, density, seed, diag=True, scale=1.0):
    rs = np.random.RandomState(seed)
    m = scisp.random(n, n, density=density, random_state=rs, format='coo', dtype=np.float64)
    m = scisp.triu(m, k=1)
    m = m + m.T
    if diag:
        m = m + scisp.diags(rs.uniform(0.5, 2.0, n))
    return (m * scale).tocsr()


def planted(n, n_blocks, p_in, p_out, seed):
    rs = np.random.RandomState(seed)
    lab = rs.randint(0, n_blocks, n)
    rows, cols, vals = [], [], []
    m_in = int(p_in * n * n / n_blocks / 2)
    for _ in range(3):
        a = rs.randint(0, n, m_in)
        # partner inside the same block: pick a random member by rejection-free trick
        order = np.argsort(lab, kind='stable')
        start = np.searchsorted(lab[order], np.arange(n_blocks))
        cnt = np.bincount(lab, minlength=n_blocks)
        b = order[start[lab[a]] + (rs.random_sample(m_in) * cnt[lab[a]]).astype(np.int64)]
        rows.append(a)
        cols.append(b)
        vals.append(rs.geometric(0.2, m_in).astype(np.float64))
    m_out = int(p_out * n * n / 2)
    rows.append(rs.randint(0, n, m_out))
    cols.append(rs.randint(0, n, m_out))
    vals.append(rs.geometric(0.5, m_out).astype(np.float64))
    r = np.concatenate(rows)
    c = np.concatenate(cols)
    v = np.concatenate(vals)
    keep = r != c
    m = scisp.coo_matrix((v[keep], (r[keep], c[keep])), shape=(n, n)).tocsr()
    m = m + m.T
    m = m + scisp.diags(rs.geometric(0.05, n).astype(np.float64))
    m.sum_duplicates()
    return m.tocsr()


def kr_cases():
    cases = {}
    rs = np.random.RandomState(7)
    a = rs.uniform(0.1, 1.0, (64, 64))
    a = a + a.T + 64 * np.eye(64)
    a[rs.random_sample((64, 64)) < 0.3] = 0
    a = np.triu(a) + np.triu(a, 1).T
    cases['dense64'] = (scisp.csr_matrix(a), {})
    cases['random1000'] = (sym_random(1000, 0.05, 42), {})
    cases['planted5000'] = (planted(5000, 25, 0.02, 0.0004, 3), {})
    # 5 % zero diagonals -> working-copy substitution, _orig scaled on return
    m = sym_random(800, 0.03, 11).tolil()
    z = np.random.RandomState(12).choice(800, 40, replace=False)
    m[z, z] = 0
    m = m.tocsr()
    m.eliminate_zeros()
    cases['zero_diag'] = (m, {})
    # delta bound: weak diagonal, strong off-diagonal, widely spread row sums
    rs = np.random.RandomState(21)
    m = sym_random(400, 0.05, 21, diag=False)
    s = scisp.diags(np.exp(rs.normal(0, 3.0, 400)))
    m = (s @ m @ s + scisp.diags(np.full(400, 1e-6))).tocsr()
    cases['delta_bound'] = (m, {})
    # Delta bound: tiny row sums need y to grow past Delta=3 in the first outer iteration
    m = (sym_random(300, 0.08, 31) * 1e-3).tocsr()
    cases['bigDelta_bound'] = (m, {})
    # already bistochastic: permutation-like + identity mix
    n = 128
    p = np.roll(np.arange(n), 1)
    m = scisp.coo_matrix((np.full(n, 0.25), (np.arange(n), p)), shape=(n, n))
    m = (m + m.T + scisp.diags(np.full(n, 0.5))).tocsr()
    cases['already_bisto'] = (m, {})
    # two disconnected blocks
    m = scisp.block_diag((sym_random(200, 0.1, 51), sym_random(300, 0.05, 52) * 7.0)).tocsr()
    cases['two_blocks'] = (m, {})
    # slightly asymmetric (warning path only)
    m = sym_random(500, 0.04, 61).tolil()
    m[3, 7] = m[3, 7] + 0.25
    cases['asym'] = (m.tocsr(), {})
    # max_iter split: raise (n_iter > max_iter) vs warn (n_iter == max_iter)
    cases['maxiter_raise'] = (planted(2000, 10, 0.03, 0.001, 71), {'max_iter': 5})
    cases['maxiter_small'] = (sym_random(600, 0.05, 81), {'max_iter': 3})
    # warn-only: a max_iter that lands exactly on an outer-iteration boundary (found by search in main())
    cases['maxiter_warn'] = (sym_random(600, 0.05, 81), {'max_iter': None})
    cases['tol1e-10'] = (sym_random(1000, 0.05, 42), {'tol': 1e-10})
    # power-law degrees
    sm = make_synth(6000, 100000, 5, 'powerlaw', genome_law='sqrt')
    cases['powerlaw6000'] = (sm.to_scipy(np.float64), {})
    # COO input (the CLI path hands COO: contact_map.py:929)
    cases['coo_input'] = (sym_random(300, 0.05, 91).tocoo(), {})
    return cases


def run_kr(ns, cap, m, kwargs):
    del cap.records[:]
    out = {}
    try:
        b, x = ns['kr_biostochastic'](m, **kwargs)
        b = b.tocsr()
        b.sort_indices()
        out['x'] = x
        out['out_indptr'] = b.indptr.astype(np.int64)
        out['out_indices'] = b.indices.astype(np.int32)
        out['out_data'] = b.data
        out['error'] = ''
    except RuntimeError as ex:
        out['error'] = str(ex)
    msgs = cap.records[:]
    n_iter = -1
    for lvl, msg in msgs:
        mm = re.match(r'It took (\d+) iterations', msg)
        if mm:
            n_iter = int(mm.group(1))
    out['n_iter'] = n_iter
    out['warn_zero_diag'] = int(any('zeros on diagonal' in msg for _, msg in msgs))
    out['warn_asym'] = int(any('expected to be fully symmetric' in msg for _, msg in msgs))
    out['warn_maxiter'] = int(any('maximum number of iterations' in msg for _, msg in msgs))
    return out


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote {} ({:.1f} kB)'.format(path, os.path.getsize(path) / 1e3))


def main():
    os.makedirs(OUT, exist_ok=True)
    ns, cap = build_namespace()

    # ---------------- KR ----------------
    for name, (m, kwargs) in kr_cases().items():
        mc = m.tocsr().copy()
        mc.sort_indices()
        if 'max_iter' in kwargs and kwargs['max_iter'] is None:
            for mi in range(1, 60):
                res = run_kr(ns, cap, m, {'max_iter': mi})
                if res['error'] == '' and res['warn_maxiter']:
                    kwargs = {'max_iter': mi}
                    break
            else:
                raise RuntimeError('no warn-only max_iter found')
        res = run_kr(ns, cap, m, kwargs)
        print('kr_{}: n={} nnz={} n_iter={} err={!r} warns z{} a{} m{}'.format(
            name, mc.shape[0], mc.nnz, res['n_iter'], res['error'], res['warn_zero_diag'], res['warn_asym'],
            res['warn_maxiter']))
        save('kr_' + name, in_indptr=mc.indptr.astype(np.int64), in_indices=mc.indices.astype(np.int32),
             in_data=mc.data.astype(np.float64), is_coo=int(scisp.isspmatrix_coo(m)),
             tol=kwargs.get('tol', 1e-6), max_iter=kwargs.get('max_iter', 1000), **res)

    # ---------------- max_offdiag / compress ----------------
    sm = make_synth(1500, 30000, 17, genome_law='sqrt')
    a = sm.to_scipy(np.uint32)
    mo = ns['max_offdiag'](a.tocoo())
    # rows whose only entry is the diagonal
    a2 = a.tolil()
    for r in (5, 77, 300):
        a2[r, :] = 0
        a2[:, r] = 0
        a2[r, r] = 9
    a2 = a2.tocsr().astype(np.uint32)
    a2.eliminate_zeros()
    mo2 = ns['max_offdiag'](a2.tocoo())
    save('max_offdiag', indptr=a.indptr.astype(np.int64), indices=a.indices, data=a.data, out=mo,
         indptr2=a2.indptr.astype(np.int64), indices2=a2.indices, data2=a2.data, out2=mo2)

    rs = np.random.RandomState(5)
    af = a.astype(np.float64)
    masks = {'random': rs.random_sample(a.shape[0]) < 0.7, 'all_true': np.ones(a.shape[0], dtype=bool),
             'one_false': np.arange(a.shape[0]) != 42, 'few_true': rs.random_sample(a.shape[0]) < 0.01}
    arrs = {'indptr': af.indptr.astype(np.int64), 'indices': af.indices, 'data': af.data}
    for k, msk in masks.items():
        c = ns['compress'](af.tocoo(), msk).tocsr()
        c.sort_indices()
        arrs['mask_' + k] = msk
        arrs['out_indptr_' + k] = c.indptr.astype(np.int64)
        arrs['out_indices_' + k] = c.indices.astype(np.int32)
        arrs['out_data_' + k] = c.data
        arrs['out_n_' + k] = c.shape[0]
    save('compress', **arrs)

    # ---------------- cluster pipeline: mask -> site norm -> KR -> compress -> scl ----------------
    # follows cluster.py:301-316 / contact_map.py:888-945,966-981 call by call with the sliced functions
    sm = make_synth(1500, 40000, 23, genome_law='sqrt')
    seq_map = sm.to_scipy(np.uint32).tocoo()
    min_len, min_sig = 2500, 40
    signal = ns['max_offdiag'](seq_map)
    mask = (sm.lengths >= min_len) & (signal >= min_sig)
    _map = seq_map.astype(np.float64)
    sites = sm.sites.copy()
    sites[np.where(sites == 0)] = 1
    ns['fast_norm_fullseq_bysite'](_map.row, _map.col, _map.data, sites)
    normed = _map.tocsr()
    normed.sort_indices()
    del cap.records[:]
    proc, x = ns['kr_biostochastic'](_map)
    proc = proc.tocsr()
    proc.sort_indices()
    sub = ns['compress'](proc.astype(np.float64).tocoo(), mask)
    scl = 1.0 / sub.max()
    # graph weights: later (v,u) overwrites (u,v) in nx.Graph.add_edge (cluster.py:319-321); emulate with a dict
    wmap = {}
    for u, v, w in zip(sub.row, sub.col, sub.data):
        wmap[(min(u, v), max(u, v))] = w * scl
    gu = np.array([k[0] for k in wmap], dtype=np.int32)
    gv = np.array([k[1] for k in wmap], dtype=np.int32)
    gw = np.array([wmap[k] for k in wmap])
    o = np.lexsort((gv, gu))
    subc = sub.tocsr()
    subc.sort_indices()
    save('pipeline', indptr=sm.indptr, indices=sm.indices, counts=sm.counts, lengths=sm.lengths, sites=sm.sites,
         min_len=min_len, min_sig=min_sig, signal=signal, mask=mask,
         normed_data=normed.data, x=x, proc_indptr=proc.indptr.astype(np.int64), proc_indices=proc.indices,
         proc_data=proc.data, sub_indptr=subc.indptr.astype(np.int64), sub_indices=subc.indices, sub_data=subc.data,
         scl=scl, graph_u=gu[o], graph_v=gv[o], graph_w=gw[o])

    # ---------------- SeqOrder + cluster_map standardisation ----------------
    SeqOrder = ns['SeqOrder']

    class _SI(object):
        def __init__(self, length):
            self.length = length

    n = 60
    rs = np.random.RandomState(99)
    lens = rs.randint(1000, 50000, n).astype(np.int32)
    order = SeqOrder([_SI(int(v)) for v in lens])
    msk = rs.random_sample(n) < 0.75
    order.set_mask_only(msk)
    n_acc = int(order.count_accepted())
    pos_after_mask = order.order['pos'].copy()
    positions = order.all_positions().copy()
    gapless = order.gapless_positions().copy()
    dense = rs.permutation(n_acc)[:20]
    remap_plain = order.remap_gapless(dense)
    remap_index = order.remap_gapless(SeqOrder.asindex(np.sort(dense)))
    # a made-up solver result in gapless ids: {cl: ids}
    lab = rs.randint(0, 6, n_acc)
    cl_to_ids = {int(c): np.flatnonzero(lab == c) for c in np.unique(lab)}

    class _CM(object):
        pass

    cmock = _CM()
    cmock.order = order
    loc = dict(ns)
    loc.update({'cl_to_ids': cl_to_ids, 'contact_map': cmock})
    exec(ns['_standardise_src'], loc)
    clustering = loc['clustering']
    arrs = {'lengths': lens, 'mask': msk, 'pos_after_mask': pos_after_mask, 'positions': positions,
            'gapless': gapless, 'dense': dense, 'remap_plain': remap_plain,
            'remap_index_index': remap_index['index'], 'remap_index_ori': remap_index['ori'],
            'labels': lab, 'n_clusters': len(clustering)}
    for k in clustering:
        arrs['cl{}_seq_ids'.format(k)] = np.asarray(clustering[k]['seq_ids'])
        arrs['cl{}_extent'.format(k)] = clustering[k]['extent']
        arrs['cl{}_name'.format(k)] = clustering[k]['name']
    # name widths at the 9/10/11 and 99/100/101 boundaries (cluster.py:23-41)
    for k in (1, 9, 10, 11, 99, 100, 101, 1000):
        d = {i: {} for i in range(k)}
        ns['add_cluster_names'](d)
        arrs['names_{}_first'.format(k)] = d[0]['name']
        arrs['names_{}_last'.format(k)] = d[k - 1]['name']
    save('seqorder_cluster', **arrs)


if __name__ == '__main__':
    sys.dont_write_bytecode = True
    main()
