"""Power-law (hub) robustness probe at a C3-like scale on one GPU (development; GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
import numpy as np
from bin3c_amd.pipeline import make_context, run_hot_path
from bin3c_amd.synth import make_synth
n, E = int(sys.argv[1]), int(sys.argv[2])
t = time.time(); sm = make_synth(n, E, 5, 'powerlaw'); print('gen %.1fs nnz=%d maxdeg=%d' % (time.time() - t, sm.nnz, np.diff(sm.indptr).max()))
ctx = make_context(0)
counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
for rep in range(2):
    t = time.time(); res = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5); dt = time.time() - t
    print('step %.1f ms' % (dt * 1e3), {k: round(v * 1e3, 1) for k, v in res.timings.items()}, 'kr it', res.kr_info.n_iter, 'spmv avg us',
          round(res.kr_info.spmv_ms / res.kr_info.n_spmv * 1e3), 'sweeps', res.lv_info.n_sweeps, 'levels', res.lv_info.n_levels, 'comm', res.lv_info.n_comm,
          'Q', round(res.lv_info.modularity, 4))
