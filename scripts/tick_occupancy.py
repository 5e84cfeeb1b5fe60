"""Leapfrog time of layout 2 (one workgroup per chain) when 1, 2 or more workgroups share a CU:
J sites x 4 chains at the C2 site size on 256 CUs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
for J in [int(a) for a in sys.argv[1:]] or (64, 128):
    mod = models.MODELS['m4b'](J, 16, 200)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=40)
    for layout in (2, 1):
        opts = HipEngine.sampler_opts(chains=4, iter=40, init='random', layout=layout)
        best = 1e9
        for rep in range(3):
            stats, ms = M.engine.sample_batch(np.arange(J) + 1, opts)
            cs = M.engine.get_chain_stats(4)[:, :, 3]
            best = min(best, ms)
        print('J=%d layout %d: %.1f ms, slowest chain %d leapfrogs -> %.2f us each; %.1f M leapfrogs/s overall'
              % (J, layout, best, cs.max(), best * 1e3 / cs.max(), cs.sum() / best / 1e3))
