"""A/B in one process: C3 (512 sites), EP iterations 1..N with and without the piece queue (same draws)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
nit = int(sys.argv[1]) if len(sys.argv) > 1 else 8
J = int(sys.argv[2]) if len(sys.argv) > 2 else 512
shape = os.environ.get('AB_SHAPE', 'c3')
mod = models.m4b(J, 32, 500) if shape == 'c3' else models.m4b(J, 128, 2000)
data = mod.simulate_data(Sigma_x='rand', rng=100) if shape == 'c3' else mod.simulate_data(rng=100)
_, _, Q0, r0 = mod.get_prior()
res = {}
pp = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [8]
for tag in ['plain'] + ['segmented%d' % x for x in pp]:
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
               df0=models.default_df0(J), sync_sites=False)
    if tag.startswith('plain'):
        M.engine.set_piece_queue = lambda *a, **k: None      # never set: one workgroup per site, longest first
    if tag.startswith('segmented'):
        M.PIECES_PER_SITE = int(tag[9:])
    M.run(nit, verbose=False, seed=1)
    res[tag] = (np.asarray(M.sampling_ms), M.Q.copy(), M.engine.last_segments())
    print(tag, 'launch ms', np.round(M.sampling_ms, 1), 'pieces of the last launch', M.engine.last_segments())
    del M
for x in pp:
    r = res['segmented%d' % x]
    print('%d pieces per site: same global Q %s, time / plain over iterations 4..: %.3f' % (x, np.array_equal(res['plain'][1], r[1]), r[0][min(3, nit - 1):].sum() / res['plain'][0][min(3, nit - 1):].sum()))
