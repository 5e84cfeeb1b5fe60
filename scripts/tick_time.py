"""Per-leapfrog ("tick") latency of the sampler kernel at the BASELINE site sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master

def run(name, J, D, n, layouts=(1, 2), it=40):
    mod = models.MODELS[name](J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    eng = M.engine
    for layout in layouts:
        opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=layout)
        seeds = np.arange(J) + 1
        stats, ms = eng.sample_batch(seeds, opts)
        stats, ms = eng.sample_batch(seeds, opts)
        cs = eng.get_chain_stats(4)
        ng = cs[:, :, 3]
        print('%s J=%d D=%d n=%d layout=%d: %.2f ms, ngrad/chain max %d mean %.0f -> %.2f us/tick (max chain), total grads %.3g, %.1f Mgrad/s'
              % (name, J, D, n, layout, ms, ng.max(), ng.mean(), ms * 1e3 / ng.max(), ng.sum(), ng.sum() / ms / 1e3))

if __name__ == '__main__':
    run('m4b', 64, 16, 200)
    run('m1b', 64, 16, 200)
    run('m4b', 512, 32, 500, layouts=(1,), it=20)
    run('m4b', 4, 4, 50)
