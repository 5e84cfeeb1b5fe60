"""Bring-up check of k_nuts_duo (layouts 5 / 6) against k_nuts (layout 1) and the other layouts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master


def setup(J, D, n, it, model='m4b'):
    mod = models.MODELS[model](J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    return M


def compare(J, D, n, it, model='m4b'):
    M = setup(J, D, n, it, model)
    eng = M.engine
    seeds = np.arange(J) + 11
    out = {}
    for lay in (1, 5, 6, 2):
        opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=lay)
        t0 = time.time()
        stats, ms = eng.sample_batch(seeds, opts)
        dr = np.stack([eng.get_draws(k, all_params=True) for k in range(J)])
        out[lay] = (dr, eng.get_chain_stats(4).copy(), ms, eng.last_layout())
    d1, c1 = out[1][0], out[1][1]
    for lay in (5, 6, 2):
        dl, cl, ms, ll = out[lay]
        same = np.array_equal(dl, d1)
        print('%s J=%d D=%d n=%d it=%d: layout %d (ran %d) %.1f ms vs layout 1 %.1f ms: draws identical %s, max|diff| %.3g, ngrad equal %s, fail %s'
              % (model, J, D, n, it, lay, ll, ms, out[1][2], same, np.abs(dl - d1).max(), np.array_equal(cl[:, :, 3], c1[:, :, 3]), cl[:, :, 7].sum()))
    # gradient through each layout
    th = 0.3 * np.random.RandomState(0).randn(eng.P)
    g = {lay: eng.logdensity_grad(1, th, layout=lay) for lay in (1, 2, 5, 6)}
    for lay in (2, 5, 6):
        print('   grad layout %d vs 1: lp diff %.3g, grad max diff %.3g' % (lay, abs(g[lay][0] - g[1][0]), np.abs(g[lay][1] - g[1][1]).max()))


def timing(J, nit, layout):
    mod = models.m4b(J, 32, 500)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=4, iter=200, df0=models.default_df0(J), layout=layout, sync_sites=False)
    info = M.run(nit, verbose=False, seed=1)[0]
    lf = M.engine.get_chain_stats(4)[:, :, 3]
    print('C3 site size J=%d layout %d (ran %d) info %d: launches ms %s ; last: gradients %.4g, us per leapfrog of the slowest workgroup chain %.2f, mean leapfrogs/transition %.0f'
          % (J, layout, M.engine.last_layout(), info, np.round(M.sampling_ms, 1), lf.sum(), M.sampling_ms[-1] * 1e3 / lf.max(), lf.mean() / 200))
    return M


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('all', 'cmp'):
        compare(6, 16, 200, 40)
        compare(6, 32, 500, 30)
        compare(5, 16, 120, 40, 'm1b')
        compare(5, 22, 300, 30)
    if what in ('all', 'time'):
        timing(256, 4, 1)
        timing(256, 4, 5)
        timing(512, 5, 5)
