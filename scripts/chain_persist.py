"""Are the slow chains of one EP iteration the slow chains of the next one?  (decides whether
the previous iteration's leapfrog counts can steer a per-site choice of layout)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
J, D, n = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 32, 500)))
mod = models.MODELS['m4b'](J, D, n)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200)
prev = None
for it in range(6):
    M.run(1, verbose=False)
    cs = M.engine.get_chain_stats(4)
    lf = cs[:, :, 3]
    site_max = lf.max(axis=1)
    if prev is not None:
        slow = np.where(site_max > 100000)[0]
        order = np.argsort(-prev)
        rank = np.empty(J, int); rank[order] = np.arange(J)
        print('iter %d: %d sites with a chain > 100K leapfrogs; their rank by last iteration\'s site max: %s'
              % (it, slow.size, np.sort(rank[slow])[:40]))
        print('   corr(log site max, prev) = %.2f' % np.corrcoef(np.log(site_max), np.log(prev))[0, 1])
        print('   slow chains per slow site:', np.bincount((lf[slow] > 100000).sum(axis=1), minlength=5))
    prev = site_max
