"""Determinism stress of the piece queue on the streaming layout: 40 sites (D = 40, n = 260), one plain launch and eleven
queued ones (12 pieces per site) in one process; every launch must give the draws of the first bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
it, J = 36, 40
mod = models.m4b(J, 40, 260)
data = mod.simulate_data(rng=100)
_, _, Q0, r0 = mod.get_prior()
bad = 0
ref = None
for rep in range(12):
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    eng, seeds = M.engine, np.arange(J) + 5
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random')
    if rep > 0:
        eng.set_piece_queue(3, np.linspace(1.0, 3.0, J))
    eng.sample_batch(seeds, opts)
    dr = np.stack([eng.get_draws(k, all_params=True) for k in range(J)])
    if ref is None: ref = dr
    ok = np.array_equal(dr, ref)
    bad += (not ok)
    print(rep, eng.last_layout(), eng.last_segments(), 'OK' if ok else 'DIFFERENT')
    del M
print('bad', bad)
