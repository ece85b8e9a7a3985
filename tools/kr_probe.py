"""Quick device timing of the KR path on a synthetic map (development probe, not the bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
import numpy as np
from bin3c_amd import _native
from bin3c_amd.synth import make_synth
n, E, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = time.time(); sm = make_synth(n, E, seed); print('gen %.1fs nnz=%d' % (time.time() - t, sm.nnz))
ctx = _native.Context(0)
counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
for rep in range(3):
    t = time.time(); a = ctx.site_norm(counts, sm.sites); t_norm = time.time() - t
    t = time.time(); _, x, info = ctx.kr_balance(a); t_kr = time.time() - t
    t = time.time(); ctx.scale_sym(a, x); t_sc = time.time() - t
    nnz = sm.nnz
    by = 12.0 * nnz + 20.0 * n
    print('rep %d norm %.2fms kr %.2fms (n_iter %d, spmv total %.2fms avg %.1fus -> %.0f GB/s) scale %.2fms' % (
        rep, t_norm * 1e3, t_kr * 1e3, info.n_iter, info.spmv_ms, info.spmv_ms / info.n_spmv * 1e3,
        by / (info.spmv_ms / info.n_spmv * 1e-3) / 1e9, t_sc * 1e3))
    a.free()
