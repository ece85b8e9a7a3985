"""Compare the KR SpMV kernel forms on one synthetic map (development probe; GPU box only)."""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
import numpy as np
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth import make_synth, CONFIGS
n, E, seed, law = CONFIGS[cfg]
sm = make_synth(n, E, seed, law)
ctx = make_context(0)
counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
n2o, o2n = ctx.locality_order(counts)
work = ctx.permute(counts, n2o)
a = ctx.site_norm(work, sm.sites[n2o])
by = 12.0 * sm.nnz + 20.0 * n
for form in ['pipe', 'rows', 'stream', 'sell', 'wave', 'pipe']:
    os.environ['B3C_KR_KERNEL'] = form
    for rep in range(2):
        t = time.time(); _, x, info = ctx.kr_balance(a); dt = time.time() - t
    print('%-7s kr %.2f ms  n_iter %d  spmv avg %.1f us -> %.0f GB/s' % (form, dt * 1e3, info.n_iter, info.spmv_ms / info.n_spmv * 1e3,
          by / (info.spmv_ms / info.n_spmv * 1e-3) / 1e9))
