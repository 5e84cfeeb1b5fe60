"""Pieces per site at the C5 shard (or any shape), one process, same draws: sampling launch times of the first EP iterations.
   python3 scripts/ab_pieces.py [sites] [D] [n] [ep_iterations] [pieces,pieces,...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master

J = int(sys.argv[1]) if len(sys.argv) > 1 else 512
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
nit = int(sys.argv[4]) if len(sys.argv) > 4 else 2
pps = [int(x) for x in sys.argv[5].split(',')] if len(sys.argv) > 5 else [16, 32, 64, 8]
mod = models.m4b(J, D, n)
data = mod.simulate_data(rng=100)
_, _, Q0, r0 = mod.get_prior()
for pp in pps:
    Master.PIECES_PER_SITE = pp
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
               prec_estim='olse' if D > 64 else 'sample', df0=models.default_df0(J), sync_sites=False)
    M.run(nit, verbose=False, calc_moments=False, seed=1)
    st = M.last_site_stats
    print('pieces per site %3d: sampling launches (ms) %s, leapfrogs per transition of the last %.0f, layout %d, pieces %d' % (
        pp, ' '.join('%.0f' % x for x in M.sampling_ms), st[:, 2].sum() / (J * 4 * 200), M.engine.last_layout(), M.engine.last_segments()), flush=True)
    M.engine.close()
