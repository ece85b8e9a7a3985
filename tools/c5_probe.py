"""GPU box: layout parameters of the KR tail pass on BASELINE's largest map as the CONTRACT generator writes it (5M contigs, Zipf genome
sizes, 1e9 power-law draws with duplicates summed).  The map is built once; every variant (comma-separated B3C_* assignments)
runs the balancing for a bounded number of iterations and prints the mean SpMV time and the set-up time.
usage: python tools/c5_probe.py [max_iter] "<variant>" "<variant>" ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
import numpy as np
import torch
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth import CONFIGS
from bin3c_amd.synth_device import make_counts_device
max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
variants = sys.argv[2:] or ['']
n, E, seed, law = CONFIGS['C5']
ctx = make_context(0)
counts, e_true, lengths_h, sites_h, _g, sinfo = make_counts_device(ctx, n, E, seed, law, genome_law='survey', dup='sum', draws_only=True)
torch.cuda.empty_cache()
print('C5 contract: n=%d E=%d nnz=%d' % (n, e_true, counts.nnz), flush=True)
n2o, o2n = ctx.locality_order(counts)
work = ctx.permute(counts, n2o)
counts.free()
sites = sites_h[n2o.cpu().numpy()] if torch.is_tensor(n2o) else sites_h[n2o]
ref = None
for var in variants:
    keys = []
    for kv in var.split(','):
        if kv:
            k, v = kv.split('=')
            os.environ[k] = v
            keys.append(k)
    ctx.reload_env()
    for rep in range(2):
        ctx.sync(); t = time.time(); rc, x, info = ctx.kr_balance_counts(work, sites, max_iter=max_iter, check=False); dt = time.time() - t
    n_spmv = max(info.n_spmv, 1)
    per = info.spmv_ms / n_spmv
    print('%-46s call %.1f ms = %d SpMVs x %.3f ms + %.1f ms; stream %.2f GB -> %.0f GB/s; tiles %d dict %.0f tails %.1f%% xt_words %d B' % (
        var or '(default)', dt * 1e3, info.n_spmv, per, dt * 1e3 - info.spmv_ms, info.spmv_stream_bytes / 1e9,
        info.spmv_stream_bytes / 1e9 / (per * 1e-3) if per > 0 else 0, info.tcd_tiles, info.tcd_dict_total / max(1, info.tcd_tiles),
        100.0 * info.tcd_tail_entries / work.nnz, info.xt_word_bytes), flush=True)
    if ref is None:
        ref = (x, info.n_iter)
    else:
        print('    vs first: n_iter %s, max rel diff %.2e' % ('same' if info.n_iter == ref[1] else 'DIFFERENT', np.nanmax(np.abs(x - ref[0]) / ref[0])), flush=True)
    for k in keys:
        os.environ.pop(k, None)
    ctx.reload_env()
