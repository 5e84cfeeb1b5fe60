"""Leapfrog counts of every chain for EP iterations 1..N of C3 (plain launches): how well does iteration t predict t+1?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
nit = int(sys.argv[1]) if len(sys.argv) > 1 else 10
J = 512
mod = models.m4b(J, 32, 500)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
           df0=models.default_df0(J), sync_sites=False)
M.engine.set_piece_queue = lambda *a, **k: None
hist = []
orig = M.engine.tilted_batch
def wrapped(*a, **k):
    out = orig(*a, **k)
    hist.append(M.engine.get_chain_stats(4)[:, :, 3].copy())
    return out
M.engine.tilted_batch = wrapped
M.run(nit, verbose=False, seed=1)
np.savez('gpurun_out/c3_history.npz', leapfrogs=np.stack(hist), ms=np.asarray(M.sampling_ms))
print('saved', np.stack(hist).shape, np.round(M.sampling_ms, 1))
