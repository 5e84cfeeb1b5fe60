"""A/B of libepx builds on one sampling regime: C3 site size, 256 sites, EP iterations 1-3 (same draws in every
build that keeps the arithmetic).  EPX_LIB selects the build.  Prints the launch times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
J = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mod = models.m4b(J, 32, 500)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
           df0=models.default_df0(J), layout=int(os.environ.get('AB_LAYOUT', '5')), sync_sites=False)
M.run(3, verbose=False, seed=1)
print(os.environ.get('EPX_LIB', 'default'), 'launch ms', np.round(M.sampling_ms, 1), 'gradients', ['%.4g' % g for g in M.ngrad_log])
