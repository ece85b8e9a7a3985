"""KR SpMV forms on the C5 map (development probe; GPU box only): dictionary capacity / tile rows / tail pass variants.
usage: tools/c5_probe.py [n] [edges] [variant ...]   (variant = comma-separated ENV=value list, '' = default)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
os.environ['B3C_KR_DEBUG'] = '1'
import numpy as np
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth_device import make_counts_device
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5000000
E = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000000
variants = sys.argv[3:] or ['', 'B3C_KR_TCD_DCAP=8192', 'B3C_KR_TCD_DCAP=12288', 'B3C_KR_TCD_R=256', 'B3C_KR_TCD_R=128,B3C_KR_TCD_DCAP=8192']
ctx = make_context(0)
t = time.time()
counts, e_true, lengths_h, sites_h, genome, ginfo = make_counts_device(ctx, n, E, 5, 'powerlaw')
print('map: n %d edges %d nnz %d in %.0f s' % (n, e_true, counts.nnz, time.time() - t), flush=True)
n2o, o2n = ctx.locality_order(counts)
work = ctx.permute(counts, n2o)
counts.free()
sites = sites_h[n2o]
ref = None
for var in variants:
    keys = []
    for kv in var.split(','):
        if kv:
            k, v = kv.split('=')
            os.environ[k] = v
            keys.append(k)
    for rep in range(2):
        t = time.time(); rc, x, info = ctx.kr_balance_counts(work, sites, max_iter=40, check=False); dt = time.time() - t
    print('%-44s KR %.1f ms, spmv avg %.1f us over %d; tiles %d dmax %d dmean %.0f tails %.1f%% (xt %d)' % (
        var or '(default)', dt * 1e3, info.spmv_ms / max(info.n_spmv, 1) * 1e3, info.n_spmv, info.tcd_tiles,
        info.tcd_dict_max, info.tcd_dict_total / max(1, info.tcd_tiles), 100.0 * info.tcd_tail_entries / work.nnz, info.tcd_xt_tails), flush=True)
    if ref is None:
        ref = x
    else:
        print('    vs first: max rel diff %.2e' % np.nanmax(np.abs(x - ref) / ref), flush=True)
    for k in keys:
        os.environ.pop(k, None)
