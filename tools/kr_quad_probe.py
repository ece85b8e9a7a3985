"""Development probe (GPU box only): SpMV time of the count-form KR at C3 for kernel variants chosen by env switches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
import numpy as np
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth import make_edges, CONFIGS
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
variants = sys.argv[2:] or ['', 'B3C_KR_QUADPROBE=1', 'B3C_KR_VEC=-1']
n, E, seed, law = CONFIGS[cfg]
ctx = make_context(0)
lo, hi, c, diag, lengths, sites, genome = make_edges(n, E, seed, law)
counts, _ = ctx.coo_symmetrise(n, lo, hi, c, diag)
n2o, o2n = ctx.locality_order(counts)
work = ctx.permute(counts, n2o)
sites = np.ascontiguousarray(sites[n2o])
for var in variants + variants[:1]:
    keys = []
    for kv in var.split(','):
        if kv:
            k, v = kv.split('=')
            os.environ[k] = v
            keys.append(k)
    for rep in range(2):
        rc, x, info = ctx.kr_balance_counts(work, sites, max_iter=40, check=False)
    print('%-28s spmv avg %.1f us over %d (rc %d)' % (var or 'default', info.spmv_ms / info.n_spmv * 1e3, info.n_spmv, rc), flush=True)
    for k in keys:
        os.environ.pop(k)
