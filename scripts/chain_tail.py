"""Per-chain leapfrog totals of one sampler launch at the C2 shape: how far the slowest chain
(which sets the launch time, one chain per block) sits above the mean."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
J, D, n = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 16, 200)))
mod = models.MODELS['m4b'](J, D, n)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
           prec_estim='sample')
for it in range(4):
    M.run(1, verbose=False)
    cs = M.engine.get_chain_stats(4)
    lf = cs[:, :, 3].ravel()
    ms = M.engine.last_sample_ms if hasattr(M.engine, 'last_sample_ms') else float('nan')
    q = np.percentile(lf, [0, 25, 50, 75, 90, 99, 100])
    print('EP iter %d: leapfrogs/chain min %d q25 %d med %d q75 %d q90 %d q99 %d max %d  mean %.0f  max/mean %.2f'
          % ((it,) + tuple(q) + (lf.mean(), lf.max() / lf.mean())))
    print('   stats columns of the slowest chain:', cs.reshape(-1, cs.shape[2])[lf.argmax()])
    print('   pass_log', M.pass_log[-1] if M.pass_log else None)
