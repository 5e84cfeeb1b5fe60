import sys; sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/golden']
import numpy as np
from test_gpu_parity import _site_problem, _engine_with_cavity
from epstan_amd.engine import HipEngine
from oracle import nuts_oracle as no
for model, D, n, layout, it in [('m4b_sg',4,50,2,60),('m4b_sg',4,50,1,40),('m4b_sg',16,200,2,40),('m4b_sg',32,120,1,40),('m2b_sg',6,80,2,40),('m3b_sg',6,80,1,40),('m5b_sg',4,50,1,40)]:
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 7 + D, K=3)
    eng = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    Om_dev = np.stack([eng.get_cavity(k)[0] for k in range(3)])
    mu_dev = np.stack([eng.get_cavity(k)[1] for k in range(3)])
    seeds = np.array([101, 202, 303], dtype=np.int64)
    opts = HipEngine.sampler_opts(chains=4, iter=it, warmup=None, init='random', layout=layout)
    stats, ms = eng.sample_batch(seeds, opts)
    draws_o, last_o, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=4, iter=it)
    cs = eng.get_chain_stats(4)
    nk = it//2
    for k in range(3):
        dev = eng.get_draws(k, all_params=True).reshape(4, nk, P)
        err = np.abs(dev - draws_o[k]).max(axis=2)   # chains x nk
        print(model, D, layout, 'site', k, 'ms %.2f'%ms, 'err first', err[:,0].max(), 'err last', err[:,-1].max(), 'nleap', cs[k,:,2], st_o[k,:,2])
