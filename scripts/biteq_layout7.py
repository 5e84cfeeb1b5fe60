import sys, os, subprocess, json
root = '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd()
code = r'''
import sys, os, hashlib
sys.path.insert(0, %r)
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
for (J, D, n) in ((40, 32, 500), (24, 16, 200), (12, 21, 333)):
    mod = models.MODELS['m4b'](J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=40)
    opts = HipEngine.sampler_opts(chains=4, iter=40, init='random', layout=7)
    M.engine.sample_batch(np.arange(J) + 1, opts)
    dr = np.stack([M.engine.get_draws(k, True) for k in range(J)])
    print(J, D, n, M.engine.last_layout(), hashlib.sha1(dr.tobytes()).hexdigest(), float(np.abs(dr).sum()))
''' % root
for lib in sys.argv[1:]:
    print('==', lib); sys.stdout.flush()
    subprocess.run([sys.executable, '-c', code], env=dict(os.environ, EPX_LIB=os.path.join(root, lib)))
