import sys; sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/golden']
import numpy as np
from test_gpu_parity import _site_problem, _engine_with_cavity
from epstan_amd.engine import HipEngine
from oracle import nuts_oracle as no
for tight in (100.0, 1000.0):
  for it in (44, 60):
    for model, D, n, layout in [('m4b_sg',16,200,2),('m4b_sg',16,200,1),('m4b_sg',32,120,1)]:
        X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 7 + D, K=3, tight=tight)
        eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
        seeds = np.array([101, 202, 303], dtype=np.int64)
        opts = HipEngine.sampler_opts(chains=4, iter=it, warmup=None, init='random', layout=layout)
        stats, ms = eng.sample_batch(seeds, opts)
        draws_o, last_o, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=4, iter=it)
        cs = eng.get_chain_stats(4)
        nk = it//2
        errs = []
        for k in range(3):
            dev = eng.get_draws(k, all_params=True).reshape(4, nk, P)
            errs.append(np.abs(dev - draws_o[k]).max())
        print(tight, it, model, D, layout, 'maxerr', ['%.1e' % e for e in errs], 'nleap equal', np.array_equal(cs[:,:,2], st_o[:,:,2]), 'mean nleap', st_o[:,:,2].mean())
