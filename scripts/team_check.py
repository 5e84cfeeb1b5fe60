"""Layout 7 (row team on the matrix pipe, nuts_duo.hip) against the oracle and against layout 5, on the GPU box.
   python3 scripts/team_check.py [quick|full] ; exits non-zero on a failed check."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
from oracle import nuts_oracle as no
from test_gpu_parity import _engine_with_cavity, _site_problem

L = int(os.environ.get('TEAM_LAYOUT', '7'))
bad = 0


def check(ok, msg):
    global bad
    print(('ok   ' if ok else 'FAIL ') + msg, flush=True)
    if not ok:
        bad += 1


def gradients():
    for model, D, n in [('m4b_sg', 32, 500), ('m4b_sg', 16, 200), ('m1b_sg', 32, 300), ('m5b_sg', 21, 333), ('m3b_sg', 11, 64),
                        ('m2b_sg', 32, 100), ('m4b_sg', 32, 17), ('m4b_sg', 9, 77), ('m4b_sg', 32, 512)]:
        X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 100 + D)
        eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
        rng = np.random.RandomState(8)
        for k in range(2):
            theta = rng.randn(P) * 0.5
            lo, hi = k_lim[k], k_lim[k + 1]
            lp_o, g_o = no.logdensity_grad(model, X[lo:hi], y[lo:hi], mu_dev[k], Om_dev[k], theta)
            lp, g = eng.logdensity_grad(k, theta, layout=L)
            e_lp = abs(lp - lp_o) / max(1.0, abs(lp_o))
            e_g = np.abs(g - g_o).max() / max(1.0, np.abs(g_o).max())
            check(eng.last_layout() == L and e_lp < 1e-11 and e_g < 1e-10,
                  'gradient %s D=%d n=%d site %d: layout %d, lp err %.1e, grad err %.1e' % (model, D, n, k, eng.last_layout(), e_lp, e_g))


def whole_runs():
    for model, D, n, chains in [('m4b_sg', 32, 120, 4), ('m4b_sg', 16, 200, 4), ('m1b_sg', 16, 120, 3)]:
        X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 7 + D, K=3, tight=1000.)
        eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
        seeds = np.array([101, 202, 303], dtype=np.int64)
        it = 44
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=chains, iter=it, init='random', layout=L))
        draws_o, _, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=chains, iter=it)
        cs = eng.get_chain_stats(chains)
        n_full = 0; early = True
        for k in range(3):
            dev = eng.get_draws(k, all_params=True)
            ref = draws_o[k].reshape(-1, P)
            err = np.abs(dev - ref).reshape(chains, it // 2, P).max(axis=2) / max(1.0, np.abs(ref).max())
            for c in range(chains):
                early &= bool(np.all(err[c, :5] < 1e-3))
                if np.all(err[c] < 1e-4) and cs[k, c, 3] == st_o[k, c, 3]:
                    n_full += 1
        check(eng.last_layout() == L and early and n_full >= (3 * chains * 3) // 4,
              'whole run %s D=%d n=%d: first draws equal %s, chains equal to the oracle to the end %d of %d, failures %d'
              % (model, D, n, early, n_full, 3 * chains, int(cs[:, :, 7].sum())))


def teacher():
    for model, D, n in [('m4b_sg', 32, 500), ('m1b_sg', 32, 500)]:
        X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 3 + D, K=3, tight=1000.)
        eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
        seeds = np.array([11, 12, 13], dtype=np.int64)
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=40, init='random', layout=L))
        cs = eng.get_chain_stats(4)
        draws = np.stack([eng.get_draws(k, all_params=True).reshape(4, 20, P) for k in range(3)])
        q0 = draws[:, :, -1, :]
        eps = cs[:, :, 1]
        inv_e = np.repeat(draws.reshape(3, -1, P).var(axis=1)[:, None, :], 4, axis=1) + 1e-3
        out, st = eng.nuts_transitions(seeds, q0, eps, inv_e, nt=2, t_offset=5, layout=L)
        ref, st_o = no.nuts_transitions(model, X, y, k_lim, mu_dev, Om_dev, seeds, q0, eps, inv_e, nt=2, t_offset=5)
        errs = np.abs(out - ref).max(axis=(2, 3))
        nbad = int((errs > 1e-6).sum())
        check(eng.last_layout() == L and nbad <= 1, 'teacher-forced transitions %s (32, 500): %d of 12 chains differ, max err %.1e, leapfrogs %s'
              % (model, nbad, np.sort(errs.ravel())[-2], st[:, :, 2].sum()))


def timing(J=256, nit=3):
    mod = models.m4b(J, 32, 500)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    for layout in (5, L):
        M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
                   df0=models.default_df0(J), layout=layout, sync_sites=False)
        t0 = time.time()
        info = M.run(nit, verbose=False, seed=1)[0]
        ms = np.array(M.sampling_ms); ng = np.array(M.ngrad_log)
        print('layout %d (ran %d), J=%d: info %d, launch ms %s, gradients %s, ns per gradient x CU %s' %
              (layout, M.engine.last_layout(), J, info, np.round(ms, 1), ['%.4g' % g for g in ng],
               np.round(ms * 1e6 / ng * min(J, 256), 1)), flush=True)


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'quick'
    gradients()
    if bad == 0:
        whole_runs()
        teacher()
    if mode != 'quick' or bad == 0:
        timing(int(os.environ.get('TEAM_J', '256')), int(os.environ.get('TEAM_NIT', '3')))
    print('team_check: %d failed checks' % bad)
    sys.exit(1 if bad else 0)
