"""A/B per-leapfrog timing of library variants: EPX_LIB is set per subprocess."""
import subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
for name in ('m4b', 'm1b'):
    mod = models.MODELS[name](64, 16, 200)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=40)
    opts = HipEngine.sampler_opts(chains=4, iter=40, init='random', layout=2)
    best = 1e9
    for rep in range(3):
        stats, ms = M.engine.sample_batch(np.arange(64) + 1, opts)
        cs = M.engine.get_chain_stats(4)
        best = min(best, ms * 1e3 / cs[:, :, 3].max())
    print('   %%s: %%.3f us/tick' %% (name, best))
''' % root
for lib in sys.argv[1:]:
    print(lib)
    env = dict(os.environ, EPX_LIB=os.path.join(root, lib))
    subprocess.run([sys.executable, '-c', code], env=env)
