"""End-to-end EP: device path vs the CPU oracle path on the same inputs (C1 size)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
from oracle.engine_oracle import OracleEngine
for name in ('m4b', 'm1b'):
    mod = models.MODELS[name](4, 4, 50)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    res = {}
    for tag, kw in (('gpu', {}), ('cpu', {'_engine_factory': lambda m, X, y, kl: OracleEngine(m, X, y, kl)})):
        for seed in (1, 2):
            M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
                       chains=4, iter=400, df0=0.5, **kw)
            info, (m_s, S_s) = M.run(8, verbose=False, seed=seed)
            res[(tag, seed)] = (info, m_s[-1], S_s[-1])
    sd = np.sqrt(np.diag(res[('cpu', 1)][2]))
    def dist(a, b):
        return np.abs(res[a][1] - res[b][1]) / sd, np.abs(np.diag(res[a][2]) / np.diag(res[b][2]) - 1)
    for pair in ((('gpu', 1), ('cpu', 1)), (('gpu', 1), ('gpu', 2)), (('cpu', 1), ('cpu', 2)), (('gpu', 2), ('cpu', 2))):
        dm, dv = dist(*pair)
        print(name, pair, 'info', res[pair[0]][0], res[pair[1]][0], 'max |dm|/sd %.3f' % dm.max(), 'max rel dvar %.3f' % dv.max())
