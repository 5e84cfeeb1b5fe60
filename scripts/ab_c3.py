"""Per-leapfrog timing at the C3 site size (J sites of D=32, n=500) for the layouts given as
arguments `lib:layout` (default: the in-tree library with layouts 1 and 4)."""
import subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
J = int(os.environ.get('AB_SITES', '256'))
layout = int(os.environ['AB_LAYOUT'])
mod = models.MODELS['m4b'](J, 32, 500)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=16)
opts = HipEngine.sampler_opts(chains=4, iter=16, init='random', layout=layout)
best = 1e9
for rep in range(2):
    stats, ms = M.engine.sample_batch(np.arange(J) + 1, opts)
    cs = M.engine.get_chain_stats(4)
    ticks = M.engine.row_passes(4).max() if layout >= 3 else cs[:, :, 3].max()
    best = min(best, ms * 1e3 / ticks)
print('   C3-size m4b, %%d sites, layout %%d: %%.2f us/leapfrog (slowest workgroup), %%.1f Mgrad/s, %%.1f ms'
      %% (J, M.engine.last_layout(), best, cs[:, :, 3].sum() / ms / 1e3, ms))
''' % root
args = sys.argv[1:] or ['ep-stan_amd/libepx.so:1', 'ep-stan_amd/libepx.so:4']
for arg in args:
    lib, layout = arg.split(':')
    subprocess.run([sys.executable, '-c', code], env=dict(os.environ, EPX_LIB=os.path.join(root, lib), AB_LAYOUT=layout))
