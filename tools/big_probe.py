import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth import make_edges
ctx = make_context(0)
n, e = 4000000, 330000000
lo, hi, c, diag, lengths, sites, genome = make_edges(n, e, 5, 'powerlaw')
for with_hubs in (0, 1):
    l2, h2, c2 = lo, hi, c
    if with_hubs:
        rs = np.random.RandomState(7)
        el, eh, ec = [], [], []
        for h, deg in zip([11, 1234567, 3999999], (100000, 300000, 1000000)):
            o = rs.choice(n, deg, replace=False).astype(np.int64); o = o[o != h]
            el.append(np.minimum(o, h)); eh.append(np.maximum(o, h)); ec.append(rs.geometric(0.3, o.shape[0]).astype(np.uint32))
        l2 = np.concatenate([lo] + el); h2 = np.concatenate([hi] + eh); c2 = np.concatenate([c] + ec)
    counts, ne = ctx.coo_symmetrise(n, l2, h2, c2, diag)
    t = time.time()
    rc, x, info = ctx.kr_balance_counts(counts, sites, max_iter=20000, check=False)
    print('hubs', with_hubs, 'nnz', counts.nnz, 'rc', rc, 'n_iter', info.n_iter, 'outer', info.n_outer, 'rout', info.rout, 'hit', info.hit_delta, info.hit_Delta, 'x range', x.min(), x.max(), 'time %.1fs' % (time.time() - t), flush=True)
    counts.free()
