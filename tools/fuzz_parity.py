"""
Development tool (GPU box): randomised differential testing of libb3c against the CPU oracle -- random symmetric maps of
random size, density, weight law, hub rows, empty rows, zero diagonals and masks -- and, a third of the cases, planted-genome maps of
the synthetic generator on the contract genome-size law or its variant -- through KR (count and fp64 forms),
compress, the fused sub-map, the permutation, downsample and Louvain.  usage: tools/fuzz_parity.py [n_cases] [seed]
Prints one line per failing case and a summary; exit code 1 if anything differed.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import scipy.sparse as sp
import oracle
from bin3c_amd.pipeline import make_context

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = make_context(0)
fails = 0
n_unstable = [0]         # maps on which x is not determined to 1e-6 (see the probe below)
# share of the cases drawn from the synthetic generator's genome-structured maps (two thirds of them on the contract law) instead
# of the unstructured random maps below
GENOME_SHARE = float(os.environ.get('B3C_FUZZ_GENOME_SHARE', '0.35'))


def genome_map(rs):
    """a contact map of the synthetic generator (bin3c_amd/synth.py): planted genomes on the CONTRACT genome-size law (SURVEY 8(d):
    n/200 genomes ~ Zipf(1.5), one of them with 38 % of the contigs) or, one time in three, on the rank^-0.5 variant; flat or
    power-law endpoints, easy or noisy"""
    from bin3c_amd.synth import make_synth
    n = int(rs.choice([400, 1000, 2500, 6000, 12000]))
    e = int(n * float(rs.choice([4, 15, 40])))
    law = 'survey' if rs.rand() < 0.67 else 'sqrt'
    noise, boost = (0.1, 5) if rs.rand() < 0.6 else (float(rs.choice([0.3, 0.45])), 1)
    sm = make_synth(n, e, int(rs.randint(1, 1 << 30)), 'powerlaw' if rs.rand() < 0.3 else 'flat', noise, boost, genome_law=law)
    return n, sm.indptr, sm.indices, sm.counts, sm.sites


def random_map(rs):
    if GENOME_SHARE > 0 and rs.rand() < GENOME_SHARE:
        return genome_map(rs)
    n = int(rs.choice([3, 17, 64, 65, 200, 513, 1500, 4000, 6000, 9000]))    # from 4096 rows on (and 16 entries per row) KR takes the dictionary form
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
    if n > 64 and rs.rand() < 0.4:                      # a hub row
        hub = rs.randint(0, n)
        others = rs.choice(n, min(n - 1, int(rs.choice([300, 700, 3000]))), replace=False)
        others = others[others != hub]
        i = np.concatenate([i, np.full(others.shape[0], hub)])
        j = np.concatenate([j, others])
        c = np.concatenate([c, rs.geometric(0.3, others.shape[0]).astype(np.uint32)])
    a = sp.coo_matrix((c.astype(np.uint64), (i, j)), shape=(n, n)).tocsr()
    a = a + a.T
    d = rs.geometric(0.05, n).astype(np.uint64)
    d[rs.rand(n) < 0.1] = 0                                # zero diagonals
    a = (a + sp.diags(d)).tocsr()
    if rs.rand() < 0.3:                                    # empty rows / columns
        dead = rs.rand(n) < 0.05
        k = sp.diags((~dead).astype(np.uint64))
        a = (k @ a @ k).tocsr()
    a.eliminate_zeros()
    a.sum_duplicates()
    a.sort_indices()
    a.data = np.minimum(a.data, 0xFFFFFFFF)
    sites = rs.poisson(20, n).astype(np.float64)
    return n, a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.astype(np.uint32), sites


def check(name, ok, case):
    global fails
    if not ok:
        fails += 1
        print('FAIL case %d: %s' % (case, name), flush=True)


only = int(os.environ['B3C_FUZZ_CASE']) if os.environ.get('B3C_FUZZ_CASE') else None     # re-run one case with diagnostics
for case in range(n_cases):
    if only is not None and case != only:
        continue
    rs = np.random.RandomState(seed0 * 100003 + case)
    n, ip, ix, cnt, sites = random_map(rs)
    if ip[-1] == 0:
        continue
    counts = ctx.upload(ip, ix, cnt)
    # --- KR on the counts against the oracle on the explicitly normalised matrix
    nd = oracle.site_norm(ip, ix, cnt.astype(np.float64), sites)
    rc_o, xo, st = oracle.kr_balance(ip, ix, nd, max_iter=300)
    rc_d, x, info = ctx.kr_balance_counts(counts, sites, max_iter=300, check=False)
    codes = {oracle.OK: 0, oracle.E_NAN: -3, oracle.E_NOCONV: -4}
    check('kr rc %d vs %d' % (rc_d, rc_o), rc_d == codes.get(rc_o, rc_o), case)
    if rc_o == oracle.OK and rc_d == 0:
        x_ok = np.allclose(x, xo, rtol=1e-6, atol=0)                # the contract; typical agreement is 1e-12
        by_balance = False
        if not x_ok or info.n_iter != st.n_iter:
            # Is x DETERMINED to 1e-6 on this map at all?  tests/golden/spread_kr.npz (tests/test_kr_spread.py) holds maps on which
            # the reference itself, run under two numpy/scipy builds or on P A P^T, differs by 1e-6 .. 9e-4 and in its iteration
            # count.  Probe: the oracle on a permuted copy of the map.  Stable there -> the 1e-6 contract stands, unconditionally;
            # unstable -> the device must return the same code and balance the map as well as the oracle does.
            prm_ = np.random.RandomState(case).permutation(n)
            pm_ = sp.csr_matrix((nd, ix, ip), shape=(n, n))[prm_][:, prm_].tocsr()
            pm_.sort_indices()
            rc_p, xp_, stp_ = oracle.kr_balance(pm_.indptr.astype(np.int64), pm_.indices.astype(np.int32), pm_.data, max_iter=300)
            back_ = np.empty(n)
            back_[prm_] = xp_
            unstable = rc_p != rc_o or stp_.n_iter != st.n_iter or not np.allclose(back_, xo, rtol=1e-8, atol=0)
            if unstable:
                am_ = sp.csr_matrix((nd, ix, ip), shape=(n, n))
                dz_ = (am_.diagonal() == 0).astype(np.float64)
                res_o = np.abs(xo * (am_ @ xo) + dz_ * xo * xo - 1.0).max()
                res_d = np.abs(x * (am_ @ x) + dz_ * x * x - 1.0).max()
                x_ok = res_d <= max(1.5 * res_o, 1e-6) and abs(info.n_iter - st.n_iter) <= 0.1 * st.n_iter
                n_unstable[0] += 1
                by_balance = True
                print('note case %d: x is not determined to 1e-6 on this map (oracle on the permuted map: n_iter %d vs %d, x off by %.1e); '
                      'device n_iter %d, x off by %.1e, residual %.1e vs %.1e' % (case, stp_.n_iter, st.n_iter, np.max(np.abs(back_ - xo) / xo),
                                                                                 info.n_iter, np.max(np.abs(x - xo) / xo), res_d, res_o), flush=True)
        check('kr n_iter %d vs %d' % (info.n_iter, st.n_iter), info.n_iter == st.n_iter or (by_balance and x_ok), case)
        check('kr x (max rel diff %.3g, n_iter %d, n %d)' % (np.max(np.abs(x - xo) / np.abs(xo)), st.n_iter, n), x_ok, case)
        check('kr zero diag', info.n_zero_diag == st.n_zero_diag, case)
        if only is not None:
            # how well do the two scale vectors balance the map?  (rows of diag(x) A' diag(x), A' = A with unit diagonal where A's is zero)
            am = sp.csr_matrix((nd, ix, ip), shape=(n, n))
            dz = (am.diagonal() == 0).astype(np.float64)
            for nm, xx in (('oracle', xo), ('device', x)):
                rows = xx * (am @ xx) + dz * xx * xx
                print('  %s: max |row sum - 1| = %.3e' % (nm, np.abs(rows - 1.0).max()))
            print('  n %d nnz %d n_iter %d/%d hit_delta %d/%d hit_Delta %d/%d zero diag %d; x range %.3e..%.3e; largest rel diff at %d' % (
                n, ip[-1], info.n_iter, st.n_iter, info.hit_delta, st.hit_delta, info.hit_Delta, st.hit_Delta, st.n_zero_diag, xo.min(), xo.max(),
                int(np.argmax(np.abs(x - xo) / np.abs(xo)))))
        # --- balanced map, mask, sub-map
        proc_o = oracle.scale_sym(ip, ix, nd, xo)
        full = ctx.normalise_scale(counts, sites, xo, graph_form=False)
        got = full.download(structure=False)[2]
        check('normalise_scale', np.allclose(got, proc_o, rtol=4e-16, atol=0), case)
        mask = rs.rand(n) < rs.choice([0.3, 0.8, 1.0])
        m_o, sip, six, sa = oracle.compress(ip, ix, got, mask)
        sub = ctx.compress(full, mask)
        dip, dix, da = sub.download()
        check('compress', np.array_equal(dip, sip) and np.array_equal(dix, six) and np.array_equal(da, sa), case)
        fused = ctx.normalise_scale_compress(counts, sites, xo, mask, graph_form=False)
        fip, fix_, fa = fused.download()
        check('fused sub-map', np.array_equal(fip, sip) and np.array_equal(fix_, six) and np.array_equal(fa, sa), case)
        # --- Louvain on the sub-map: device schedule against its scalar twin
        if sip[-1] > 0 and sa.max() > 0:
            scl = 1.0 / oracle.max_value(sip, sa)
            w = oracle.graph_weights(sip, six, sa, scl)
            q = oracle.quantise(w, oracle.quant_shift(sip, six, w))
            ref = oracle.louvain_bsp(sip, six, q)
            labels, linfo = ctx.louvain(sub, scl)
            check('louvain labels', np.array_equal(labels, ref['labels']), case)
            check('louvain Q/sweeps', linfo.modularity == ref['Q'] and linfo.n_sweeps == ref['n_sweeps'], case)
        for m in (full, sub, fused):
            m.free()
    # --- permutation and max_offdiag
    perm = rs.permutation(n).astype(np.int32)
    pm = ctx.permute(counts, perm)
    pip, pix, pd = pm.download()
    c = sp.csr_matrix((cnt, ix, ip), shape=(n, n))
    want = c[perm][:, perm]
    want.sort_indices()
    check('permute', np.array_equal(pip, want.indptr) and np.array_equal(pix, want.indices) and np.array_equal(pd, want.data), case)
    check('max_offdiag', np.array_equal(ctx.max_offdiag_u32(counts, assume_symmetric=True), oracle.max_offdiag_u32(ip, ix, cnt)), case)
    pm.free()
    # --- downsample of the normalised map
    if n <= 1500:
        b = int(rs.choice([2, 3, 8, 50]))
        nm = ctx.upload(ip, ix, nd)
        for method in ('mean', 'max'):
            ds = ctx.downsample(nm, b, method)
            dip, dix, dd = ds.download()
            dense = sp.csr_matrix((dd, dix, dip), shape=(ds.n, ds.n)).toarray()
            exp = oracle.downsample_dense(sp.csr_matrix((nd, ix, ip), shape=(n, n)), b, method)
            check('downsample ' + method, np.allclose(dense, exp, rtol=1e-13, atol=0), case)
            ds.free()
        nm.free()
    counts.free()
    if case % 20 == 19:
        print('... %d cases, %d failures' % (case + 1, fails), flush=True)
print('fuzz: %d cases, %d failures' % (n_cases, fails))
if n_unstable[0]:
    print('      (%d maps judged by balance: x not determined to 1e-6, see tests/test_kr_spread.py)' % n_unstable[0])
sys.exit(1 if fails else 0)
