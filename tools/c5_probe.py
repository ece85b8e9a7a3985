"""One-GPU attempt at BASELINE.json's C5 (5M contigs / 1B edges, power law): build, one step, timings (development probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bin3c_amd.pipeline import make_context, run_hot_path
from bin3c_amd.synth import make_edges
n, e = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000, int(sys.argv[2]) if len(sys.argv) > 2 else 1000000000
ctx = make_context(0)
t = time.time()
lo, hi, c, diag, lengths, sites, genome = make_edges(n, e, 5, 'powerlaw')
print('edges drawn in %.0f s' % (time.time() - t), flush=True)
t = time.time()
counts, ne = ctx.coo_symmetrise(n, lo, hi, c, diag)
del lo, hi, c
print('map: n %d E %d nnz %d (%.1f GB of entries) in %.1f s' % (n, ne, counts.nnz, counts.nnz * 8 / 1e9, time.time() - t), flush=True)
for rep in range(2):
    t = time.time()
    res = run_hot_path(ctx, counts, sites, lengths, 1000, 5, max_iter=20000)
    dt = time.time() - t
    print('step %d: %.1f ms  (%.3g edges/s)  stages %s  KR n_iter %d  Louvain sweeps %d levels %d communities %d Q %.4f' % (
        rep, dt * 1e3, ne / dt, {k: round(v * 1e3, 1) for k, v in res.timings.items()}, res.kr_info.n_iter, res.lv_info.n_sweeps,
        res.lv_info.n_levels, res.lv_info.n_comm, res.lv_info.modularity), flush=True)
print('HBM held by the context: %.1f GB' % (ctx.bytes() / 1e9 if hasattr(ctx, 'bytes') else -1))
