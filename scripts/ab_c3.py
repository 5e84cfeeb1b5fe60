"""A/B per-leapfrog timing at the C3 site size (J sites of D=32, n=500, layout 1)."""
import subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
mod = models.MODELS['m4b'](256, 32, 500)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=16)
opts = HipEngine.sampler_opts(chains=4, iter=16, init='random', layout=1)
best = 1e9
for rep in range(2):
    stats, ms = M.engine.sample_batch(np.arange(256) + 1, opts)
    cs = M.engine.get_chain_stats(4)
    best = min(best, ms * 1e3 / cs[:, :, 3].max())
print('   C3-size m4b layout 1: %%.2f us/tick (slowest chain), %%.1f Mgrad/s' %% (best, cs[:, :, 3].sum() / ms / 1e3))
''' % root
for lib in sys.argv[1:]:
    print(lib)
    subprocess.run([sys.executable, '-c', code], env=dict(os.environ, EPX_LIB=os.path.join(root, lib)))
