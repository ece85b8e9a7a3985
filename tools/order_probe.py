"""Development probe (GPU box only): how close to the diagonal does the locality ordering put the stored entries,
for several numbers of Boruvka rounds, against the planted-genome order; and the KR SpMV time on each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
import numpy as np
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth import make_edges, CONFIGS
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
n, E, seed, law = CONFIGS[cfg]
ctx = make_context(0)
lo, hi, c, diag, lengths, sites, genome = make_edges(n, E, seed, law)
counts, _ = ctx.coo_symmetrise(n, lo, hi, c, diag)
del lo, hi, c


def report(tag, n2o, t_order):
    work = ctx.permute(counts, np.ascontiguousarray(n2o, dtype=np.int32))
    ip, ix, _d = work.download()
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(ip).astype(np.int64))
    d = np.abs(ix.astype(np.int64) - rows)
    del rows, ix
    fr = [float((d < w).mean()) for w in (64, 512, 2048, 8192, 65536)]
    del d
    s2 = np.ascontiguousarray(sites[n2o])
    for rep in range(2):
        rc, x, info = ctx.kr_balance_counts(work, s2, max_iter=40, check=False)
    print('%-14s order %.1f ms | within 64/512/2048/8192/65536: %s | spmv %.1f us' % (
        tag, t_order * 1e3, ' '.join('%.3f' % f for f in fr), info.spmv_ms / info.n_spmv * 1e3), flush=True)
    work.free()


report('identity', np.arange(n), 0.0)
report('genome', np.argsort(genome, kind='stable'), 0.0)
for r in (3, 4, 5, 6, 8):
    os.environ['B3C_ORDER_ROUNDS'] = str(r)
    ctx.locality_order(counts)
    ctx.sync()
    t = time.time()
    n2o, o2n = ctx.locality_order(counts)
    ctx.sync()
    report('rounds=%d' % r, n2o, time.time() - t)
