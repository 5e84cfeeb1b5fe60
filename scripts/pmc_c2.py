"""One sampler launch at C2 for rocprofv3 --pmc runs (layout from argv)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
name = sys.argv[1] if len(sys.argv) > 1 else 'm4b'
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 2
J, D, n = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (64, 16, 200)
mod = models.MODELS[name](J, D, n)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=40)
opts = HipEngine.sampler_opts(chains=4, iter=40, init='random', layout=layout)
stats, ms = M.engine.sample_batch(np.arange(J) + 1, opts)
cs = M.engine.get_chain_stats(4)
print('ms', ms, 'total ngrad', cs[:, :, 3].sum(), 'max', cs[:, :, 3].max())
