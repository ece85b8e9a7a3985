"""Replay fuzz cases of tools/fuzz_parity.py through KR with and without the dictionary layout (development probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B3C_KR_DEBUG'] = '1'
import numpy as np
import importlib.util
seed0 = int(sys.argv[1]); cases = [int(c) for c in sys.argv[2:]]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fuzz_parity.py')).read()
ns = {}
exec(src[src.index('def random_map'):src.index('def check')], {'np': np, 'sp': __import__('scipy.sparse', fromlist=['x'])}, ns)
from bin3c_amd.pipeline import make_context
ctx = make_context(0)
for case in cases:
    rs = np.random.RandomState(seed0 * 100003 + case)
    n, ip, ix, cnt, sites = ns['random_map'](rs)
    lens = np.diff(ip)
    print('case', case, 'n', n, 'nnz', ip[-1], 'max row', lens.max(), 'mean', lens.mean(), 'max count', cnt.max(), flush=True)
    counts = ctx.upload(ip, ix, cnt)
    os.environ['B3C_KR_NO_TCD'] = '1'
    rc0, x0, i0 = ctx.kr_balance_counts(counts, sites, max_iter=300, check=False)
    os.environ.pop('B3C_KR_NO_TCD')
    rc1, x1, i1 = ctx.kr_balance_counts(counts, sites, max_iter=300, check=False)
    print('   csr: rc %d n_iter %d   tcd: rc %d n_iter %d  max rel diff %.3g  nan %d' % (rc0, i0.n_iter, rc1, i1.n_iter, np.nanmax(np.abs(x1 - x0) / x0), np.isnan(x1).sum()), flush=True)
