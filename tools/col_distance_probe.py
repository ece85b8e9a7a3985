"""How local is the ordered map?  For the C3 contract map after the locality order: the share of stored entries whose column lies within
+-W/2 positions of its row, for a range of W -- what a window of W labels held in LDS around a workgroup's rows could serve of the
Louvain scan's label gathers (development probe; GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bin3c_amd.pipeline import make_context
from bin3c_amd.synth import CONFIGS, make_edges
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
n, E, seed, law = CONFIGS[cfg]
ctx = make_context(0)
lo, hi, cnt, diag, lengths_h, sites_h, _g = make_edges(n, E, seed, law, genome_law='survey')
counts, e_true = ctx.coo_symmetrise(n, lo, hi, cnt, diag)
del lo, hi, cnt
for name, mat in (('caller order', counts), ('locality order', None)):
    if mat is None:
        n2o, o2n = ctx.locality_order(counts)
        mat = ctx.permute(counts, n2o)
    ip, ix, _ = mat.download()
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ip))
    d = np.abs(ix.astype(np.int64) - rows)
    del rows
    print('%s: n = %d, nnz = %d' % (name, n, ix.shape[0]))
    for w in (1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144):
        print('   |col - row| < %7d : %5.1f %%' % (w // 2, 100.0 * float((d < w // 2).sum()) / d.shape[0]))
    del d, ix, ip
