"""Per-pass time of the streaming sampler on REAL C5 sites (m4b, D = 128, n = 2000) after `nit` EP iterations:
python3 scripts/c5_real_passes.py [sites] [ep_iters]   (EPX_LIB selects the build; sites <= CUs: one workgroup per site,
unpieced; more: the piece queue, as in the bench)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master

J = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mod = models.m4b(J, 128, 2000)
data = mod.simulate_data(rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
           df0=models.default_df0(J), sync_sites=False)
info = M.run(nit, verbose=False, seed=1)[0]
eng = M.engine
for i, (ms, p) in enumerate(zip(M.sampling_ms, M.pass_log)):
    p = np.asarray(p, dtype=float)
    print('%s J=%d iteration %d: launch %.0f ms, passes per site mean %.0f max %.0f, pieces %d -> %.1f us per pass and CU (all sites), '
          '%.1f us per pass of the slowest site' % (os.path.basename(os.environ.get('EPX_LIB', 'libepx.so')), J, i, ms, p.mean(), p.max(),
                                                   eng.last_segments(), ms * 1e3 * min(J, eng.cu_count()) / p.sum(), ms * 1e3 / p.max()))
