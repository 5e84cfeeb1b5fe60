"""Time of one lock-step pass of the row team (layout 7) from whole launches, no instrumentation: 256 sites of the C3 shape,
one workgroup per CU and no piece queue, so a launch lasts as long as its slowest site: launch / passes of that site.
python3 scripts/pass_time.py [sites] [ep_iters]   (EPX_LIB selects the build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master

J = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mod = models.m4b(J, 32, 500)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
           chains=4, iter=200, df0=models.default_df0(J), layout=7, sync_sites=False, balance_sites=False)
out = []
for it in range(nit):
    M.run(1, verbose=False, seed=1 + it)
    gl = M.engine.get_chain_stats(4)[:, :, 3]
    out.append((M.sampling_ms[-1], gl.max(axis=1).max(), gl.max(axis=1).mean()))
print('launch ms, passes of the slowest site, mean passes per site, us per pass of the slowest site:')
for ms, mx, mean in out:
    print('   %8.1f %9.0f %9.0f   %.3f' % (ms, mx, mean, ms * 1e3 / mx))
