"""What the lock step of a site's four chains loses at the piece ends, from the device's own per-transition trace
(C3 site shape, late EP iterations): passes = sum over pieces of the LONGEST chain's leapfrogs in the piece.
Schemes: (a) pieces of iter/16 transitions (today), (a8) iter/8, (b) no pieces, (c) pieces by a pass budget: a stop
signal after B passes, every chain stops at its next transition boundary (per-chain transition index in the record)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master

J, D, n, it = 128, 32, 500, 200
mod = models.m4b(J, D, n)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it,
           df0=models.default_df0(J), sync_sites=True)
M.run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, verbose=False, seed=1)
M.engine.set_trace(J)
M.run(1, verbose=False, seed=2)
tr = M.engine.get_trace(4, it)
L = tr[:, :, :, 1]                      # (site, chain, transition) leapfrogs
print('sites %d, leapfrogs per transition: mean %.0f, median %.0f; layout %d' % (J, L.mean(), np.median(L), M.engine.last_layout()))
tot = L.sum()

def eff(passes): return tot / (4.0 * passes)

def by_transitions(nb):
    edges = np.round(np.linspace(0, it, nb + 1)).astype(int)
    p = 0.0
    for a, b in zip(edges[:-1], edges[1:]):
        p += L[:, :, a:b].sum(axis=2).max(axis=1).sum()
    return p

for nb in (1, 4, 8, 16, 32):
    print('pieces of iter/%-2d transitions: chains per pass %.3f' % (nb, 4 * eff(by_transitions(nb))))

def by_budget(B):
    passes = 0.0
    for s in range(J):
        t = np.zeros(4, dtype=int)                 # next transition of every chain
        while (t < it).any():
            # every live chain runs until the first transition boundary at or behind B passes
            used = np.zeros(4)
            for c in range(4):
                while t[c] < it and used[c] < B:
                    used[c] += L[s, c, t[c]]; t[c] += 1
            passes += used.max()
    return passes

mean_site = L.sum(axis=2).max(axis=1).mean()
for div in (8, 16, 32):
    B = mean_site / div
    print('pieces by a pass budget of 1/%-2d of a site: chains per pass %.3f' % (div, 4 * eff(by_budget(B))))
