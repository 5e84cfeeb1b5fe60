"""End-to-end EP at the C2 shape (J = 64 sites, D = 16, n_j = 200, m4b, 4 x 200 NUTS iterations, default_df0):
the device path against the CPU oracle path (C NUTS on the host threads + NumPy moment / cavity stages) on
the same inputs, tolerance relative to the CPU path's own seed-to-seed spread.  TEST script (imports oracle/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
from oracle.engine_oracle import OracleEngine
niter = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mod = models.m4b(64, 16, 200)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
res = {}
for tag, seed, kw in (('gpu', 1, {}), ('gpu', 2, {}), ('cpu', 1, {'_engine_factory': lambda m, X, y, kl: OracleEngine(m, X, y, kl)}),
                      ('cpu', 2, {'_engine_factory': lambda m, X, y, kl: OracleEngine(m, X, y, kl)})):
    t0 = time.time()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
               df0=models.default_df0(64), **kw)
    info, (m_s, S_s) = M.run(niter, verbose=False, seed=seed)
    res[(tag, seed)] = (m_s[-1], S_s[-1])
    print('%s seed %d: info %d, %d iterations in %.1f s' % (tag, seed, info, niter, time.time() - t0), flush=True)
sd = np.sqrt(np.diag(res[('cpu', 1)][1]))
def dist(a, b):
    return (np.abs(res[a][0] - res[b][0]) / sd).max(), np.abs(np.diag(res[a][1]) / np.diag(res[b][1]) - 1).max()
for a, b in ((('gpu', 1), ('cpu', 1)), (('gpu', 2), ('cpu', 2)), (('cpu', 1), ('cpu', 2)), (('gpu', 1), ('gpu', 2))):
    dm, dv = dist(a, b)
    print('%s vs %s: max |mean difference| / posterior sd %.3f, max relative variance difference %.3f' % (a, b, dm, dv))
