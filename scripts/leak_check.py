"""Stress check of determinism: the same 3 EP iterations (300 sites, D = 20, n = 340, iter = 64) with and without the
piece queue, repeated; every run must give the same global parameters bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
Master.LEAD_FRACTION = 2.0
J = 300
mod = models.m4b(J, 20, 340)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()

def ep(queue, nit=3):
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=64,
               df0=models.default_df0(J), sync_sites=False)
    if not queue:
        M.engine.set_piece_queue = lambda *a, **k: None
    seen = []
    orig = M.engine.tilted_batch
    def wrapped(*a, **k):
        out = orig(*a, **k)
        seen.append(float(np.abs(M.engine.get_chain_stats(4)).sum()))
        return out
    M.engine.tilted_batch = wrapped
    info = M.run(nit, verbose=False, calc_moments=False, seed=5)
    return [round(x, 3) for x in seen], float(np.abs(M.Q).sum())

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ref = None
for i in range(n):
    for q in (True, False):
        r = ep(q)
        if ref is None:
            ref = r
        print(i, 'queue' if q else 'plain', 'OK' if r == ref else 'DIFFERENT', r if r != ref else '')
print('reference', ref)
