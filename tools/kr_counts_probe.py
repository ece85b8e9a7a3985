"""F64 vs raw-count KR on the same ordered map (development probe; GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
import numpy as np
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth import make_synth, CONFIGS
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C2'
n, E, seed, law = CONFIGS[cfg]
sm = make_synth(n, E, seed, law)
from bin3c_amd import _native
ctx = make_context(0) if os.environ.get('B3C_PROBE_STREAM', 'torch') == 'torch' else _native.Context(0)
counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
n2o, o2n = ctx.locality_order(counts)
work = ctx.permute(counts, n2o)
sites = sm.sites[n2o]
a = ctx.site_norm(work, sites)
for mode in ['u16', 'u32', 'u16']:
    os.environ.pop('B3C_KR_NO_U16', None)
    if mode == 'u32':
        os.environ['B3C_KR_NO_U16'] = '1'
    for rep in range(2):
        t = time.time(); rc, x2, i2 = ctx.kr_balance_counts(work, sites); d2 = time.time() - t
    print('%-9s counts: %.2f ms (spmv avg %.0f us over %d)' % (mode, d2 * 1e3, i2.spmv_ms / i2.n_spmv * 1e3, i2.n_spmv))
