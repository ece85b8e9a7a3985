"""GPU box: what the locality order is worth on a bench map -- run_hot_path with and without it (development probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bin3c_amd.pipeline import make_context, run_hot_path
from bin3c_amd.synth import CONFIGS, make_edges
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
law = sys.argv[2] if len(sys.argv) > 2 else 'survey'
n, E, seed, dlaw = CONFIGS[cfg]
ctx = make_context(0, pin=True)
lo, hi, cnt, diag, lengths, sites, _g = make_edges(n, E, seed, dlaw, genome_law=law)
counts, e_true = ctx.coo_symmetrise(n, lo, hi, cnt, diag)
del lo, hi, cnt
sites_t = torch.as_tensor(sites, device='cuda')
lengths_t = torch.as_tensor(lengths.astype(np.int64), device='cuda')
for reorder in (True, False):
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t = time.time()
        res = run_hot_path(ctx, counts, sites_t, lengths_t, 1000, 5, reorder=reorder)
        torch.cuda.synchronize()
        ts.append(time.time() - t)
        st = dict(res.stage_ms) if hasattr(res, 'stage_ms') else {}
        info = (res.kr_info.n_iter, res.lv_info.n_sweeps, res.lv_info.n_comm)
        res.free()
    print('reorder=%s: %.1f ms per step (best of last 3), stages %s, iterations / sweeps / communities %s' % (
        reorder, min(ts[1:]) * 1e3, {k: round(v, 1) for k, v in st.items()}, info), flush=True)
