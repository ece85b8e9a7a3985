"""Louvain kernel timing probe: delta-Q scan time per launch for several n_sub (development; GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_PROFILE'] = '1'
import numpy as np
from bin3c_amd.pipeline import make_context, run_hot_path
from bin3c_amd.synth import make_synth, CONFIGS
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
n, E, seed, law = CONFIGS[cfg]
sm = make_synth(n, E, seed, law)
ctx = make_context(0)
counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
res = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5, keep=True)
sub = res.sub if res.sub is not None else res.processed
for n_sub in (4, 1, 2, 8, 4):
    for rep in range(2):
        t = time.time(); labels, info = ctx.louvain(sub, res.scl, n_sub=n_sub); dt = time.time() - t
    print('n_sub %d: louvain %.1f ms, sweeps %d, decide total %.2f ms over %d launches = %.0f us/launch, %.2f ms/sweep, Q=%.6f' % (
        n_sub, dt * 1e3, info.n_sweeps, info.move_ms, info.n_move_launch, info.move_ms / info.n_move_launch * 1e3,
        info.move_ms / info.n_sweeps, info.modularity))
