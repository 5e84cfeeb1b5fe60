"""K < J (several groups per site): one EP sampling launch per layout at the reference's default
experiment shape (fit.py:134-151: J=64, K=32, D=16, npg=20) and a larger one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
from epstan_amd.util import distribute_groups
import sys
CASES = ((64, 32, 16, 20), (1024, 256, 16, 50))
if len(sys.argv) > 4:
    CASES = (tuple(int(v) for v in sys.argv[1:5]),)
for J, K, D, npg in CASES:
    mod = models.m4b(J, D, npg)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    Nk, Nj_k, j_ind_k = distribute_groups(J, K, data.Nj)
    for layout in ((2, 4) if len(sys.argv) > 4 else (2, 3, 4)):
        M = Master('m4b', data.X, data.y, site_sizes=Nk, A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k + 1},
                   prior={'Q': Q0, 'r': r0}, chains=4, iter=100, layout=layout)
        opts = HipEngine.sampler_opts(chains=4, iter=100, init='random', layout=layout)
        best = 1e9
        for rep in range(2):
            stats, ms = M.engine.sample_batch(np.arange(K) + 1, opts)
            best = min(best, ms)
        ll = M.engine.last_layout()
        # layout 2: one workgroup per chain -> the slowest CHAIN sets the launch; 3 / 4: chains in lock step
        ticks = M.engine.get_chain_stats(4)[:, :, 3].max() if ll == 2 else M.engine.row_passes(4).max()
        print('J=%d K=%d D=%d npg=%d layout %d: %.1f ms, %.2f us per leapfrog of the slowest %s, P=%d'
              % (J, K, D, npg, ll, best, best * 1e3 / ticks, 'chain' if ll == 2 else 'site (4 chains in lock step)', M.engine.P))
