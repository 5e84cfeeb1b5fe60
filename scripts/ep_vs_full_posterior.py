"""End-to-end check that needs no reference: the EP posterior of the device path against the FULL
posterior of the same hierarchical model, sampled with the oracle's NUTS on the joint model (one
"site" that holds all J groups with the prior as its cavity -- nuts_oracle.c, 4 chains on the host).
Shape: the reference's default experiment (fit.py:134-168: m4b, J = 64 groups, D = 16, 20 rows per
group) with K = 32 sites (two groups per site) or K = 64.  TEST / MEASUREMENT script (imports oracle/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import fit, models
from oracle import nuts_oracle as no

K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
niter = int(sys.argv[2]) if len(sys.argv) > 2 else 40
siter = int(sys.argv[3]) if len(sys.argv) > 3 else 200      # NUTS iterations per chain and site update (fit.py default 200)
J, D, npg = 64, 16, 20
mod = models.m4b(J, D, npg)
data = mod.simulate_data(Sigma_x='rand', rng=100)
S0, m0, Q0, r0 = mod.get_prior()
d = mod.dphi
t0 = time.time()
draws, _, st = no.nuts_sites('m4b', data.X, data.y, np.array([0, data.X.shape[0]]), m0[None], Q0[None], [11], chains=4,
                             iter=2000, g_cnt=np.array([J], dtype=np.int32), g_lim=data.j_lim.astype(np.int64), nthreads=4)
x = draws[0].reshape(-1, draws.shape[-1])[:, :d]
m_full, sd_full = x.mean(0), x.std(0)
rhat = max(no.split_rhat(draws[0, :, :, e]) for e in range(d))
print('full posterior (oracle NUTS on the joint model, P = %d): %.0f s, %d divergences, max split-Rhat of phi %.3f'
      % (draws.shape[-1], time.time() - t0, st[0, :, 4].sum(), rhat))
conf = fit.configurations(run_ep=True, iter=niter, save_res=False, K=K, siter=siter)
t0 = time.time()
res = fit.main('m4b', conf, verbose=False)
dt = time.time() - t0
m, S = res['m_s_ep'], res['S_s_ep']
print('EP on the device: K = %d sites, %d iterations of 4 x %d NUTS iterations per site in %.1f s' % (K, niter, siter, dt))
for it in sorted(set([0, 1, 2, 5, 10, 20, niter])):
    z = np.abs(m[it] - m_full) / sd_full
    r = np.sqrt(np.diagonal(S[it])) / sd_full
    print('  iteration %3d: |EP mean - full mean| / full sd: median %.2f max %.2f;  EP sd / full sd: median %.2f (min %.2f, max %.2f)'
          % (it, np.median(z), z.max(), np.median(r), r.min(), r.max()))
print('for scale: |full mean - phi_true| / full sd: median %.2f max %.2f (the data of one simulated set do not pin phi_true)'
      % (np.median(np.abs(m_full - data.phi_true) / sd_full), (np.abs(m_full - data.phi_true) / sd_full).max()))
