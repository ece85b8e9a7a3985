"""
How far apart are two CORRECT evaluations of the reference's Knight-Ruiz iteration on ill-conditioned maps?

VERDICT (round 2) asked that the 1e-6 agreement of the scale vector x not be relaxed for maps that need >= 100
iterations without knowing what the reference's own spread is.  This tool (build container only: it executes the
reference's function text from /root/reference) takes the random maps of tools/fuzz_parity.py, keeps those on which
the oracle needs >= `min_iter` iterations, and evaluates x

  ref-a   the reference text under /usr/bin/python3        (scipy 1.15, numpy 2.2)
  ref-b   the reference text under /opt/conda/bin/python3.9 (scipy 1.7.1, numpy 1.26)
  oracle  oracle/kr.c (scalar C restatement, explicit normalised matrix)
  perm    the reference text (ref-a) on P A P^T for a random permutation P, mapped back: the SAME sums in another order

and prints, per case, the largest relative difference of x between them and each vector's residual max|row sum - 1|.
`--save` writes the cases whose ref-vs-ref or ref-vs-permuted spread exceeds 1e-6 to tests/golden/spread_kr.npz
(inputs, both x, the residuals): the committed evidence for the tolerance rule of tests/tools.

usage: PYTHONDONTWRITEBYTECODE=1 python tools/kr_spread.py [n_cases] [seed] [--save]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
import oracle  # noqa: E402

MAX_ITER = 300


def random_map(rs):
    """tools/fuzz_parity.py's generator restricted to sizes the reference's LIL conversion handles in seconds"""
    n = int(rs.choice([3, 17, 64, 65, 200, 513, 1500]))
    deg = float(rs.choice([1.5, 6, 30, 150, 400]))
    deg = min(deg, n - 1)
    m = int(n * deg / 2)
    i = rs.randint(0, n, m)
    j = rs.randint(0, n, m)
    keep = i != j
    i, j = i[keep], j[keep]
    law = rs.choice(['geom', 'big', 'ones'])
    if law == 'geom':
        c = rs.geometric(0.2, i.shape[0]).astype(np.uint32)
    elif law == 'big':
        c = rs.randint(1, 200000, i.shape[0]).astype(np.uint32)
    else:
        c = np.ones(i.shape[0], dtype=np.uint32)
    if n > 64 and rs.rand() < 0.4:
        hub = rs.randint(0, n)
        others = rs.choice(n, min(n - 1, int(rs.choice([300, 700, 3000]))), replace=False)
        others = others[others != hub]
        i = np.concatenate([i, np.full(others.shape[0], hub)])
        j = np.concatenate([j, others])
        c = np.concatenate([c, rs.geometric(0.3, others.shape[0]).astype(np.uint32)])
    a = sp.coo_matrix((c.astype(np.uint64), (i, j)), shape=(n, n)).tocsr()
    a = a + a.T
    d = rs.geometric(0.05, n).astype(np.uint64)
    d[rs.rand(n) < 0.1] = 0
    a = (a + sp.diags(d)).tocsr()
    if rs.rand() < 0.3:
        dead = rs.rand(n) < 0.05
        k = sp.diags((~dead).astype(np.uint64))
        a = (k @ a @ k).tocsr()
    a.eliminate_zeros()
    a.sum_duplicates()
    a.sort_indices()
    a.data = np.minimum(a.data, 0xFFFFFFFF)
    sites = rs.poisson(20, n).astype(np.float64)
    return n, a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.astype(np.uint32), sites


def ill_map(rs):
    """tools/fuzz_ill.py's generator (tiny maps, 1.2 - 3 contacts per row)"""
    n = int(rs.choice([33, 64, 65, 120, 200]))
    deg = float(rs.choice([1.2, 1.5, 2.0, 3.0]))
    m = int(n * deg / 2)
    i = rs.randint(0, n, m)
    j = rs.randint(0, n, m)
    keep = i != j
    i, j = i[keep], j[keep]
    law = rs.choice(['geom', 'big', 'ones'])
    c = (rs.geometric(0.2, i.shape[0]) if law == 'geom' else rs.randint(1, 200000, i.shape[0]) if law == 'big'
         else np.ones(i.shape[0])).astype(np.uint64)
    a = sp.coo_matrix((c, (i, j)), shape=(n, n)).tocsr()
    a = a + a.T
    d = rs.geometric(0.05, n).astype(np.uint64)
    d[rs.rand(n) < 0.1] = 0
    a = (a + sp.diags(d)).tocsr()
    a.eliminate_zeros(); a.sum_duplicates(); a.sort_indices()
    a.data = np.minimum(a.data, 0xFFFFFFFF)
    sites = rs.poisson(20, n).astype(np.float64)
    return n, a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.astype(np.uint32), sites


def residual(ip, ix, a, x):
    n = ip.shape[0] - 1
    m = sp.csr_matrix((a, ix, ip), shape=(n, n))
    dz = (m.diagonal() == 0).astype(np.float64)
    return float(np.abs(x * (m @ x) + dz * x * x - 1.0).max())


def run_ref(python, cases_path):
    out = tempfile.mktemp(suffix='.npz')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    subprocess.check_call([python, '-W', 'ignore', os.path.join(HERE, 'kr_spread_ref.py'), cases_path, out], env=env)
    z = dict(np.load(out))
    os.unlink(out)
    return z


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n_cases = int(args[0]) if args else 300
    seed0 = int(args[1]) if len(args) > 1 else 1
    min_iter = 100
    ill = '--ill' in sys.argv                       # the maps of tools/fuzz_ill.py (same seeds -> same maps as on the GPU box)
    global MAX_ITER
    if ill:
        MAX_ITER = 400
    only = None
    for a in sys.argv[1:]:
        if a.startswith('--cases='):
            only = set(int(t) for t in a.split('=')[1].split(','))
    oracle.build()
    cases = {}
    meta = []
    for case in range(n_cases):
        if only is not None and case not in only:
            continue
        if ill:
            n, ip, ix, cnt, sites = ill_map(np.random.RandomState(seed0 * 1000003 + case))
        else:
            n, ip, ix, cnt, sites = random_map(np.random.RandomState(seed0 * 100003 + case))
        if ip[-1] == 0:
            continue
        nd = oracle.site_norm(ip, ix, cnt.astype(np.float64), sites)
        rc, xo, st = oracle.kr_balance(ip, ix, nd, max_iter=MAX_ITER)
        if rc != oracle.OK or st.n_iter < min_iter:
            continue
        k = len(meta)
        perm = np.random.RandomState(case).permutation(n)
        m = sp.csr_matrix((nd, ix, ip), shape=(n, n))
        pm = m[perm][:, perm].tocsr()
        pm.sort_indices()
        cases['ip_%d' % (2 * k)], cases['ix_%d' % (2 * k)], cases['a_%d' % (2 * k)] = ip, ix, nd
        cases['ip_%d' % (2 * k + 1)], cases['ix_%d' % (2 * k + 1)], cases['a_%d' % (2 * k + 1)] = pm.indptr.astype(np.int64), pm.indices.astype(np.int32), pm.data
        meta.append((case, n, int(ip[-1]), int(st.n_iter), int(st.hit_delta + st.hit_Delta), xo, perm, cnt, sites))
        if only is not None:
            print('case %d: oracle n_iter %d' % (case, st.n_iter))
    print('%d of %d maps need >= %d iterations' % (len(meta), n_cases, min_iter), flush=True)
    if not meta:
        return
    cases['max_iter'] = np.array(MAX_ITER)
    path = tempfile.mktemp(suffix='.npz')
    np.savez(path, **cases)
    za = run_ref(sys.executable, path)
    zb = run_ref('/opt/conda/bin/python3.9', path)
    os.unlink(path)
    print('ref-a = %s\nref-b = %s' % (za['env'], zb['env']))
    rel = lambda u, v: float(np.max(np.abs(u - v) / np.abs(v)))
    keep = {}
    worst = {'ab': 0.0, 'ao': 0.0, 'ap': 0.0}
    print('%5s %6s %7s %5s %4s | %9s %9s %9s | residuals: %9s %9s %9s' % ('case', 'n', 'nnz', 'iter', 'cuts', 'ref a-b', 'ref-orc', 'ref-perm', 'ref-a', 'oracle', 'perm'))
    for k, (case, n, nnz, it, cuts, xo, perm, cnt, sites) in enumerate(meta):
        if not (za['ok_%d' % (2 * k)] and zb['ok_%d' % (2 * k)] and za['ok_%d' % (2 * k + 1)]):
            continue
        xa, xb = za['x_%d' % (2 * k)], zb['x_%d' % (2 * k)]
        xp = np.empty(n)
        xp[perm] = za['x_%d' % (2 * k + 1)]
        ip, ix, a = cases['ip_%d' % (2 * k)], cases['ix_%d' % (2 * k)], cases['a_%d' % (2 * k)]
        ita, itb, itp = int(za['it_%d' % (2 * k)]), int(zb['it_%d' % (2 * k)]), int(za['it_%d' % (2 * k + 1)])
        if len({ita, itb, itp, it}) > 1:
            print('%5d   n_iter: ref-a %d  ref-b %d  ref-a on the permuted map %d  oracle %d' % (case, ita, itb, itp, it))
        dab, dao, dap = rel(xb, xa), rel(xo, xa), rel(xp, xa)
        worst['ab'], worst['ao'], worst['ap'] = max(worst['ab'], dab), max(worst['ao'], dao), max(worst['ap'], dap)
        ra, ro, rp = residual(ip, ix, a, xa), residual(ip, ix, a, xo), residual(ip, ix, a, xp)
        flag = ' <-- beyond 1e-6' if max(dab, dap) > 1e-6 else ''
        print('%5d %6d %7d %5d %4d | %9.2e %9.2e %9.2e | %20.2e %9.2e %9.2e%s' % (case, n, nnz, it, cuts, dab, dao, dap, ra, ro, rp, flag))
        if (max(dab, dap) > 1e-6 or len({ita, itb, itp}) > 1) and len(keep) < 8 * 15 and n <= 600:
            j = len(keep) // 15
            keep.update({'n_iter_ref_a_%d' % j: np.array(ita), 'n_iter_ref_b_%d' % j: np.array(itb), 'n_iter_ref_perm_%d' % j: np.array(itp)})
            keep.update({'case_%d' % j: np.array(case), 'seed_%d' % j: np.array(seed0), 'ip_%d' % j: ip, 'ix_%d' % j: ix, 'counts_%d' % j: cnt,
                         'sites_%d' % j: sites, 'x_ref_a_%d' % j: xa, 'x_ref_b_%d' % j: xb, 'x_ref_perm_%d' % j: xp,
                         'perm_%d' % j: perm, 'n_iter_%d' % j: np.array(it), 'cuts_%d' % j: np.array(cuts)})
    print('largest spread: ref-a vs ref-b %.2e, ref vs oracle %.2e, ref vs ref on the permuted map %.2e' % (worst['ab'], worst['ao'], worst['ap']))
    if '--save' in sys.argv and keep:
        keep['env_a'], keep['env_b'] = za['env'], zb['env']
        out = os.path.join(REPO, 'tests', 'golden', 'spread_kr.npz')
        np.savez_compressed(out, **keep)
        print('saved %d cases to %s' % (len(keep) // 15, out))


if __name__ == '__main__':
    main()
