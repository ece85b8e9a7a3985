"""Development probe (GPU box): HBM held by the libb3c pool and by torch over repeated steps (leak check)."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from bin3c_amd.pipeline import make_context, run_hot_path
from bin3c_amd.synth import make_edges, CONFIGS
ctx=make_context(0)
n,e,seed,law=CONFIGS['C2']
lo,hi,c,diag,lengths,sites,genome=make_edges(n,e,seed,law)
counts,_=ctx.coo_symmetrise(n,lo,hi,c,diag)
for i in range(40):
    r=run_hot_path(ctx,counts,sites,lengths,1000,5)
    if i%10==9: print(i+1, 'pool bytes', ctx.bytes_in_use(), 'torch alloc', torch.cuda.memory_allocated(), 'reserved', torch.cuda.memory_reserved(), flush=True)
