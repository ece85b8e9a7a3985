"""Build container: tests/golden/illcond_kr_counts.npz -- the one map of the round-3 ill-conditioned campaign (tools/fuzz_ill.py,
6000 maps, seed 1, case 3338: 64 contigs, 126 entries, 262 iterations, 20 bound cuts) on which the COUNT form of the device
balance (b3c_kr_balance_counts) was further than 1e-6 from the reference while the explicit fp64 form (b3c_kr_balance on
the normalised matrix) and the reference's own two environments agree to 1e-8.  Inputs, the reference's x (function text
executed here, both environments) and the two device vectors recorded on the GPU box.
usage: PYTHONDONTWRITEBYTECODE=1 python tools/make_golden3.py gpurun_out/fuzz_ill/case_1_3338.npz"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
import oracle  # noqa: E402

z = np.load(sys.argv[1])
ip, ix, cnt, sites = z['ip'], z['ix'], z['counts'], z['sites']
nd = oracle.site_norm(ip, ix, cnt.astype(np.float64), sites)
tmp = tempfile.mktemp(suffix='.npz')
np.savez(tmp, ip_0=ip, ix_0=ix, a_0=nd, max_iter=np.array(400))
outs = []
for py in (sys.executable, '/opt/conda/bin/python3.9'):
    o = tempfile.mktemp(suffix='.npz')
    subprocess.check_call([py, '-W', 'ignore', os.path.join(HERE, 'kr_spread_ref.py'), tmp, o], env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    outs.append(dict(np.load(o)))
    os.unlink(o)
os.unlink(tmp)
xa, xb = outs[0]['x_0'], outs[1]['x_0']
rel = lambda u, v: float(np.max(np.abs(u - v) / np.abs(v)))
print('ref a-b %.2e  device counts vs ref %.2e  device fp64 vs ref %.2e  n_iter ref %d / %d  device %d' % (
    rel(xb, xa), rel(z['x_counts'], xa), rel(z['x_fp64'], xa), outs[0]['it_0'], outs[1]['it_0'], z['n_iter']))
np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'illcond_kr_counts.npz'), ip=ip, ix=ix, counts=cnt, sites=sites,
                    x_ref_a=xa, x_ref_b=xb, n_iter_ref=outs[0]['it_0'], x_dev_counts=z['x_counts'], x_dev_fp64=z['x_fp64'],
                    env_a=outs[0]['env'], env_b=outs[1]['env'])
