import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np
from epstan_amd.engine import HipEngine
from test_gpu_parity import _engine_with_cavity, _site_problem
X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 16, 200, 23, K=3, tight=1000.)
eng, Om_dev, mu_dev = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
seeds = np.array([101, 202, 303], dtype=np.int64)
out = {}
for nm, env in (('full', '1'), ('lean', None)):
    if env: os.environ['EPX_NO_LEAN'] = env
    else: os.environ.pop('EPX_NO_LEAN', None)
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=20, init='random', layout=7))
    out[nm] = (np.stack([eng.get_draws(k, all_params=True) for k in range(3)]), eng.get_chain_stats(4))
a, b = out['full'][0], out['lean'][0]
print('layout', eng.last_layout(), 'max diff', np.abs(a - b).max())
print('leapfrogs full', out['full'][1][:, :, 2], '\nleapfrogs lean', out['lean'][1][:, :, 2])
print('stepsize full', out['full'][1][:, :, 1], '\nstepsize lean', out['lean'][1][:, :, 1])
