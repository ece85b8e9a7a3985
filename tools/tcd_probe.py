"""KR SpMV forms on the bench map (development probe; GPU box only): dictionary tiles vs the CSR row-group kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
os.environ['B3C_KR_DEBUG'] = '1'
import numpy as np
import torch
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth import CONFIGS, make_edges
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
variants = sys.argv[2:] or ['B3C_KR_NO_TCD=1', '', 'B3C_KR_TCD_XT=0', 'B3C_KR_TCD_P24=0', 'B3C_KR_TCD_MINUSE8=0']
genome_law = os.environ.get('TCD_PROBE_LAW', 'survey')      # the contract workload unless told otherwise
n, E, seed, law = CONFIGS[cfg]
ctx = make_context(0)
lo, hi, cnt, diag, lengths_h, sites_h, _g = make_edges(n, E, seed, law, genome_law=genome_law)
counts, e_true = ctx.coo_symmetrise(n, lo, hi, cnt, diag)
del lo, hi, cnt
n2o, o2n = ctx.locality_order(counts)
work = ctx.permute(counts, n2o)
sites = sites_h[n2o]
ref = None
for var in variants:
    keys = []
    for kv in var.split(','):
        if kv:
            k, v = kv.split('=')
            os.environ[k] = v
            keys.append(k)
    ctx.reload_env()          # libb3c reads its switches once per context
    for rep in range(2):
        t = time.time(); rc, x, info = ctx.kr_balance_counts(work, sites, max_iter=1000, check=False); dt = time.time() - t
    print('%-40s KR %.2f ms, spmv avg %.1f us over %d, n_iter %d; tiles %d dmax %d dmean %.0f tails %.1f%%' % (
        var or '(default)', dt * 1e3, info.spmv_ms / max(info.n_spmv, 1) * 1e3, info.n_spmv, info.n_iter, info.tcd_tiles,
        info.tcd_dict_max, info.tcd_dict_total / max(1, info.tcd_tiles), 100.0 * info.tcd_tail_entries / work.nnz), flush=True)
    if ref is None:
        ref = (x, info.n_iter)
    else:
        print('    vs first: n_iter %s, max rel diff %.2e' % ('same' if info.n_iter == ref[1] else 'DIFFERENT', np.nanmax(np.abs(x - ref[0]) / ref[0])), flush=True)
    for k in keys:
        os.environ.pop(k, None)
    ctx.reload_env()
