"""Dumps the leapfrog counts of every chain of a C3 sampling launch in the steady state (EP iteration N) together
with the launch time and the dispatch order: input of scripts/balance_model.py.
usage: python3 scripts/c3_chain_lengths.py [iterations] [out.npz]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
nit = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/c3_chains.npz'
J = 512
mod = models.m4b(J, 32, 500)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
           df0=models.default_df0(J), sync_sites=False)
M.run(nit, verbose=False, seed=1)
hist = [M.engine.get_chain_stats(4)[:, :, 3].copy()]
np.savez(out, leapfrogs=np.stack(hist), ms=np.asarray(M.sampling_ms), layout=M.engine.last_layout())
print('launch ms', np.round(M.sampling_ms, 1))
print('last: mean chain %.0f max %.0f ; per-site max/mean %.3f' % (hist[-1].mean(), hist[-1].max(), (hist[-1].max(axis=1) / hist[-1].mean(axis=1)).mean()))
