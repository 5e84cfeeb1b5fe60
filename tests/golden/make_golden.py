#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

What it does:
  * copies /root/reference/{epstan,setup.py} into a *temporary* directory
    outside this repository and builds the Cython helper there,
  * registers stub `pystan` / `pystan.constants` modules (PyStan is absent;
    MAX_UINT = 2**31 - 1 is the value PyStan 2.17 defines),
  * replaces `epstan.method._sample_stan` by the deterministic injected
    samplers of tests/golden/injectors.py (the forked child inherits them),
  * drives the unmodified reference functions and stores inputs + outputs as
    small .npz fixtures.

Only data is written into the repository: no reference source, no bytecode.
"""

import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.path.insert(0, HERE)
import injectors  # noqa: E402


def import_reference():
    tmp = tempfile.mkdtemp(prefix='epstan_ref_')
    shutil.copytree(os.path.join(REF, 'epstan'), os.path.join(tmp, 'epstan'))
    shutil.copy(os.path.join(REF, 'setup.py'), tmp)
    subprocess.check_call([sys.executable, 'setup.py', 'build_ext', '--inplace'],
                          cwd=tmp, stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    ps = types.ModuleType('pystan')
    ps.StanModel = object
    pc = types.ModuleType('pystan.constants')
    pc.MAX_UINT = 2**31 - 1
    ps.constants = pc
    sys.modules['pystan'] = ps
    sys.modules['pystan.constants'] = pc
    sys.path.insert(0, tmp)
    sys.path.insert(0, os.path.join(REF, 'experiment'))
    from epstan import util, method
    return util, method, tmp


def rand_spd(rng, d, scale=1.0):
    A = rng.randn(d, 2 * d + 3)
    S = A.dot(A.T) / (2 * d + 3) * scale
    return np.asfortranarray(0.5 * (S + S.T))


def patch_sampler(method, fn):
    def _inj(queue, path, data, stan_params, other_params=None):
        samp = fn(data, stan_params)
        queue.put([np.asfortranarray(samp),
                   [{}] * stan_params['chains'], 0.25, 0.125, 1.0625])
    method._sample_stan = _inj


# ----------------------------------------------------------------------------
def g1_invert(util, out):
    from scipy import linalg
    rng = np.random.RandomState(11)
    for d in (5, 17, 33, 66):
        S = rand_spd(rng, d)
        m = rng.randn(d)
        Q, r = util.invert_normal_params(S, m)
        out['g1_S_%d' % d] = S
        out['g1_m_%d' % d] = m
        out['g1_Q_%d' % d] = Q
        out['g1_r_%d' % d] = r
        U = np.asfortranarray(np.triu(linalg.cho_factor(S.copy(order='F'))[0]))
        Q2, r2 = util.invert_normal_params(U, m, cho_form=True)
        out['g1_U_%d' % d] = U
        out['g1_Qc_%d' % d] = Q2
        out['g1_rc_%d' % d] = r2
    # non positive-definite input -> LinAlgError
    bad = rand_spd(rng, 6)
    bad[3, 3] = -1.0
    try:
        util.invert_normal_params(bad, rng.randn(6))
        raised = 0
    except np.linalg.LinAlgError:
        raised = 1
    out['g1_bad'] = bad
    out['g1_bad_raises'] = np.array(raised)


def g2_olse(util, out):
    rng = np.random.RandomState(12)
    for d in (5, 17, 33):
        for n in (100, 400):
            Sig = rand_spd(rng, d)
            x = rng.randn(n, d).dot(np.linalg.cholesky(Sig).T)
            x -= x.mean(0)
            Shat = np.asfortranarray(x.T.dot(x) / n)
            P = rand_spd(rng, d, 2.0)
            key = '%d_%d' % (d, n)
            out['g2_S_' + key] = Shat
            out['g2_P_' + key] = P
            out['g2_none_' + key] = util.olse(Shat, n)
            out['g2_prior_' + key] = util.olse(Shat, n, P=P)


def g3_cavity(method, out):
    rng = np.random.RandomState(13)
    d = 10
    X = rng.randn(20, 4)
    y = (rng.rand(20) < 0.5).astype(int)
    w = method.Worker(0, 'none/m4b_sg', d, X, y)
    Q = rand_spd(rng, d, 4.0)
    r = rng.randn(d)
    Qi = rand_spd(rng, d, 0.2)
    ri = 0.1 * rng.randn(d)
    flag = w.cavity(Q, r, Qi, ri)
    out['g3_Q'] = Q
    out['g3_r'] = r
    out['g3_Qi'] = Qi
    out['g3_ri'] = ri
    out['g3_flag'] = np.array(flag)
    out['g3_Mat'] = w.Mat.copy()
    out['g3_vec'] = w.vec.copy()
    out['g3_phase'] = np.array(w.phase)
    # non-pd cavity: site precision larger than the global one
    Qi_bad = Q + rand_spd(rng, d, 0.5)
    flag = w.cavity(Q, r, Qi_bad, ri)
    out['g3_Qi_bad'] = Qi_bad
    out['g3_flag_bad'] = np.array(flag)
    out['g3_phase_bad'] = np.array(w.phase)


def g4_samples(seed, S, d):
    """Correlated, shifted draws; a pure function of (seed, S, d)."""
    rng = np.random.RandomState(seed)
    mix = np.eye(d) + 0.3 * rng.randn(d, d) / np.sqrt(d)
    shift = rng.randn(d)
    return np.asfortranarray(rng.randn(S, d).dot(mix) + shift)


def g4_tilted(method, out):
    rng = np.random.RandomState(14)
    X = rng.randn(20, 4)
    y = (rng.rand(20) < 0.5).astype(int)
    S = 400
    for d in (5, 10, 17, 34):
        for est in ('sample', 'olse'):
            seed = 1000 + d
            patch_sampler(method, lambda data, sp, _s=seed, _d=d:
                          g4_samples(_s, S, _d))
            w = method.Worker(0, 'none/m4b_sg', d, X, y, prec_estim=est,
                              chains=4, iter=200)
            Q = rand_spd(np.random.RandomState(50 + d), d, 3.0)
            r = np.random.RandomState(60 + d).randn(d)
            Qi = np.zeros((d, d), order='F')
            ri = np.zeros(d)
            assert w.cavity(Q, r, Qi, ri)
            dQi = np.zeros((d, d), order='F')
            dri = np.zeros(d)
            flag = w.tilted(dQi, dri, seed=7)
            key = '%s_%d' % (est, d)
            out['g4_Q_%d' % d] = Q
            out['g4_r_%d' % d] = r
            out['g4_seed_%d' % d] = np.array(seed)
            out['g4_dQi_' + key] = dQi
            out['g4_dri_' + key] = dri
            out['g4_vec_' + key] = w.vec.copy()
            out['g4_flag_' + key] = np.array(flag)
            out['g4_nsamp_' + key] = np.array(w.nsamp)
            out['g4_stanseed'] = np.array(w.stan_params['seed'])
            if est == 'olse':
                out['g4_scatter_%d' % d] = w.Mat.copy()   # unnormalised C'C
            else:
                R = np.triu(w.Mat)
                out['g4_scatter_qr_%d' % d] = R.T.dot(R)


def g5_master_init(method, out):
    rng = np.random.RandomState(15)
    N, D = 30, 3
    X = rng.randn(N, D)
    y = (rng.rand(N) < 0.5).astype(int)
    sizes = np.array([7, 11, 12])
    m = method.Master('none/m1b_sg', X, y, site_sizes=sizes, dphi=D + 1,
                      init_site=3.0)
    out['g5_X'] = X
    out['g5_y'] = y
    out['g5_sizes'] = sizes
    out['g5_Nk'] = np.asarray(m.Nk)
    out['g5_k_lim'] = m.k_lim
    out['g5_k_ind'] = m.k_ind
    out['g5_Q'] = m.Q.copy()
    out['g5_r'] = m.r.copy()
    out['g5_Qi'] = m.Qi.copy()
    S, mm = m.cur_approx()
    out['g5_S'] = S
    out['g5_m'] = mm
    ind_ord = np.repeat(np.arange(3), sizes)
    m2 = method.Master('none/m1b_sg', X, y, site_ind_ord=ind_ord, dphi=D + 1)
    out['g5_ord_Nk'] = np.asarray(m2.Nk)
    out['g5_ord_k_lim'] = m2.k_lim
    out['g5_w1_X'] = np.asarray(m2.workers[1].data['X'])
    out['g5_w1_Mat'] = m2.workers[1].Mat.copy()
    out['g5_w1_vec'] = m2.workers[1].vec.copy()


def g6_run(method, out, models):
    """Master.run trajectories at C1 size (J=K=4, D=4, n_j=50), m1b data."""
    mod = models['m1b'].model(4, 4, 50)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    K = 4
    out['g6_X'] = data.X
    out['g6_y'] = data.y
    out['g6_Nj'] = data.Nj
    out['g6_Q0'] = Q0
    out['g6_r0'] = r0

    def run(tag, scenario, niter, df0, prec_estim='sample', nsites=K, factor=60.0,
            prior_scale=1.0):
        inj = injectors.GaussianTilted(scenario, factor=factor)
        patch_sampler(method, inj)
        nrow = int(np.sum(data.Nj[:nsites]))
        M = method.Master(
            'none/m1b_sg', data.X[:nrow], data.y[:nrow],
            site_sizes=data.Nj[:nsites],
            A_k={'site_id': np.arange(nsites)},
            prior={'Q': Q0 * prior_scale, 'r': r0 * prior_scale},
            chains=4, iter=200, df0=df0, prec_estim=prec_estim)
        res = M.run(niter, verbose=False, return_analytics=True, seed=1)
        info = res[0]
        m_s, S_s = res[1]
        st, ms, rh, ot = res[2]
        out['g6_%s_info' % tag] = np.array(info)
        out['g6_%s_m' % tag] = m_s
        out['g6_%s_S' % tag] = S_s
        out['g6_%s_stimes' % tag] = st
        out['g6_%s_msteps' % tag] = ms
        out['g6_%s_mrhats' % tag] = rh
        out['g6_%s_Qi' % tag] = M.Qi.copy()
        out['g6_%s_ri' % tag] = M.ri.copy()
        out['g6_%s_Q' % tag] = M.Q.copy()
        out['g6_%s_r' % tag] = M.r.copy()
        out['g6_%s_iter' % tag] = np.array(M.iter)
        out['g6_%s_phase' % tag] = np.array([w.phase for w in M.workers])
        return M

    rng = np.random.RandomState(1)
    out['g6_seeds_12'] = rng.randint(0, 2**31 - 1, size=(12, K))
    run('smooth', 'smooth', 12, 0.5)
    run('smooth_olse', 'smooth', 6, 0.5, prec_estim='olse')
    run('decay', 'wide_first', 4, 1.0, nsites=3, factor=60.0)
    run('allfail', 'degenerate', 3, 0.5)
    run('badprior', 'wide_all', 3, 1.0, factor=400.0)
    # second run() call continues the same state (self.iter accumulates)
    inj = injectors.GaussianTilted('smooth')
    patch_sampler(method, inj)
    M = method.Master('none/m1b_sg', data.X, data.y, site_sizes=data.Nj,
                      A_k={'site_id': np.arange(K)},
                      prior={'Q': Q0, 'r': r0}, chains=4, iter=200, df0=0.5)
    M.run(2, verbose=False, seed=5)
    info, (m_s, S_s) = M.run(2, verbose=False, seed=6)
    out['g6_resume_m'] = m_s
    out['g6_resume_S'] = S_s
    out['g6_resume_iter'] = np.array(M.iter)


def g7_simulators(out, models):
    def summarize(tag, name, J, D, n, full):
        mod = models[name].model(J, D, n)
        data = mod.simulate_data(Sigma_x='rand', rng=100)
        S0, m0, Q0, r0 = mod.get_prior()
        out['g7_%s_Nj' % tag] = data.Nj
        out['g7_%s_Q0diag' % tag] = np.diag(Q0).copy()
        out['g7_%s_r0' % tag] = r0
        out['g7_%s_phi_true' % tag] = data.true_values['phi']
        out['g7_%s_mu_x' % tag] = data.X_param['mu_x']
        out['g7_%s_sigma_x' % tag] = data.X_param['sigma_x']
        if full:
            out['g7_%s_X' % tag] = data.X
            out['g7_%s_y' % tag] = data.y
            out['g7_%s_Sigma_x' % tag] = data.X_param['Sigma_x']
        else:
            out['g7_%s_X_head' % tag] = data.X[:8].copy()
            out['g7_%s_X_tail' % tag] = data.X[-8:].copy()
            out['g7_%s_X_colsum' % tag] = data.X.sum(0)
            out['g7_%s_X_sitesum' % tag] = np.add.reduceat(
                data.X.sum(1), data.j_lim[:-1])
            out['g7_%s_y_sitesum' % tag] = np.add.reduceat(
                data.y, data.j_lim[:-1])
            out['g7_%s_y_head' % tag] = data.y[:64].copy()
    summarize('m1b_c1', 'm1b', 4, 4, 50, True)
    summarize('m4b_c1', 'm4b', 4, 4, 50, True)
    summarize('m1b_c2', 'm1b', 64, 16, 200, False)
    summarize('m4b_c2', 'm4b', 64, 16, 200, False)
    summarize('m4b_c3', 'm4b', 512, 32, 500, False)


def g8_seeds(out):
    """Seed derivation of method.py:342-346 and :956-960 (RandomState only)."""
    seeds = np.random.RandomState(1).randint(0, 2**31 - 1, size=(3, 5))
    stan = np.array([[np.random.RandomState(s).randint(0, 2**31 - 1)
                      for s in row] for row in seeds])
    out['g8_seeds'] = seeds
    out['g8_stan_seeds'] = stan


def g9_damp_sweep(method, util, out, models):
    """The damping sweep of experiment/find_damp.py:137-173 at C1 size, driven through the
    reference's own Master / Worker.cavity / invert_normal_params / find_damp.kl_mvn."""
    import find_damp
    from scipy import linalg, stats
    from scipy.linalg import cho_factor
    mod = models['m1b'].model(4, 4, 50)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    K = 4
    patch_sampler(method, injectors.GaussianTilted('smooth'))
    master = method.Master('none/m1b_sg', data.X, data.y, site_sizes=data.Nj,
                           A_k={'site_id': np.arange(K)}, prior={'Q': Q0, 'r': r0},
                           chains=4, iter=200, df0=0.5)
    master.run(2, verbose=False, seed=3)
    d = master.dphi
    rng = np.random.RandomState(7)
    S_target = rand_spd(rng, d, 0.3)
    m_target = master.m + 0.3 * rng.randn(d)
    samp_target = rng.multivariate_normal(m_target, S_target, size=500)
    sum_log_diag_cho_S0 = np.sum(np.log(np.diag(cho_factor(S_target)[0])))
    S, m, Q, r = master.S, master.m, master.Q, master.r
    Qi, ri, Qi2, ri2, dQi, dri = master.Qi, master.ri, master.Qi2, master.ri2, master.dQi, master.dri
    posdefs = np.zeros(K, dtype=bool)
    for k, worker in enumerate(master.workers):
        posdefs[k] = worker.tilted(dQi[:, :, k], dri[:, k], seed=100 + k)    # find_damp.py:139 draws a random seed
    assert np.all(posdefs)
    out['g9_Q0'], out['g9_r0'] = master.Q0.copy(), master.r0.copy()
    out['g9_Qi'], out['g9_ri'] = Qi.copy(), ri.copy()
    out['g9_dQi'], out['g9_dri'] = dQi.copy(), dri.copy()
    out['g9_m_target'], out['g9_S_target'], out['g9_samp_target'] = m_target, S_target, samp_target
    # damps beyond 1 make the proposal leave the positive definite cone for this scenario
    damps = np.concatenate((np.linspace(0, 1, find_damp.N_DAMP + 2)[1:-1], [1.5, 2.5, 4.0, 5.5, 8.0, -1.0, -40.0]))
    n = damps.shape[0]
    mses, lls, kls = np.full(n, np.nan), np.full(n, np.nan), np.full(n, np.nan)
    gpd, cpd = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)
    for di, df in enumerate(damps):                                    # find_damp.py:146-173
        np.add(Qi, np.multiply(df, dQi, out=Qi2), out=Qi2)
        np.add(ri, np.multiply(df, dri, out=ri2), out=ri2)
        np.add(Qi2.sum(2, out=Q), master.Q0, out=Q)
        np.add(ri2.sum(1, out=r), master.r0, out=r)
        try:
            cho_Q = S
            np.copyto(cho_Q, Q)
            linalg.cho_factor(cho_Q, overwrite_a=True)
            util.invert_normal_params(cho_Q, r, out_A='in-place', out_b=m, cho_form=True)
            gpd[di] = True
            for k, worker in enumerate(master.workers):
                posdefs[k] = worker.cavity(Q, r, Qi2[:, :, k], ri2[:, k])
            if np.all(posdefs):
                cpd[di] = True
                mses[di] = np.mean((m - m_target)**2)
                lls[di] = np.sum(stats.multivariate_normal.logpdf(samp_target, mean=m, cov=S.T))
                kls[di] = find_damp.kl_mvn(m_target, S_target, m, S.T, sum_log_diag_cho_S0)
        except linalg.LinAlgError:
            pass
    out['g9_damps'], out['g9_mses'], out['g9_lls'], out['g9_kls'] = damps, mses, lls, kls
    out['g9_global_pd'], out['g9_cav_pd'] = gpd, cpd


def g10_distribute_groups(util, out):
    """util.distribute_groups, K < J branch (util.py:582-608)."""
    rng = np.random.RandomState(4)
    cases = [(8, 4, np.array([5, 3, 8, 2, 2, 9, 4, 4])), (64, 32, np.full(64, 20)),
             (12, 5, rng.randint(1, 30, size=12)), (7, 2, rng.randint(1, 9, size=7)), (6, 5, np.array([3, 3, 3, 3, 3, 3]))]
    for i, (J, K, Nj) in enumerate(cases):
        Nk, Nj_k, j_ind_k = util.distribute_groups(J, K, Nj)
        out['g10_%d_JK' % i] = np.array([J, K])
        out['g10_%d_Nj' % i] = Nj
        out['g10_%d_Nk' % i] = Nk
        out['g10_%d_Nj_k' % i] = Nj_k
        out['g10_%d_j_ind_k' % i] = j_ind_k
    out['g10_n'] = np.array(len(cases))


def g11_gauss_simulators(out):
    """Gaussian-likelihood simulators models/m1a.py, m4a.py (real responses; `rng=100` as fit.py:157)."""
    from models import m1a, m4a
    from models import m3a, m5a      # (m2a.py's simulator raises NameError in the reference: `rnd_data`)
    for tag, mod_ref, J, D, n, Sx in (('m1a_s', m1a, 5, 4, 20, 'rand'), ('m4a_s', m4a, 5, 4, 20, 'rand'),
                                      ('m1a_i', m1a, 3, 1, 10, None), ('m4a_i', m4a, 3, 6, 15, None),
                                      ('m3a_s', m3a, 5, 4, 20, 'rand'), ('m5a_s', m5a, 6, 5, (10, 30), None)):
        mod = mod_ref.model(J, D, n)
        data = mod.simulate_data(Sigma_x=Sx, rng=100)
        S0, m0, Q0, r0 = mod.get_prior()
        out['g11_%s_X' % tag] = data.X
        out['g11_%s_y' % tag] = data.y
        out['g11_%s_phi_true' % tag] = data.true_values['phi']
        out['g11_%s_sigma_x' % tag] = np.asarray(data.X_param['sigma_x'])
        out['g11_%s_Q0diag' % tag] = np.diag(Q0).copy()
        out['g11_%s_r0' % tag] = r0
        out['g11_%s_dphi' % tag] = np.array(mod.dphi)
    # the remaining logistic simulators models/m2b.py, m3b.py, m5b.py
    from models import m2b, m3b, m5b
    for tag, mod_ref, J, D, n, Sx in (('m2b_s', m2b, 5, 4, 20, 'rand'), ('m3b_s', m3b, 5, 4, 20, 'rand'),
                                      ('m5b_s', m5b, 5, 4, 20, 'rand'), ('m3b_r', m3b, 6, 5, (10, 30), None),
                                      ('m5b_r', m5b, 6, 5, (10, 30), None)):
        mod = mod_ref.model(J, D, n)
        data = mod.simulate_data(Sigma_x=Sx, rng=100)
        S0, m0, Q0, r0 = mod.get_prior()
        out['g12_%s_X' % tag] = data.X
        out['g12_%s_y' % tag] = data.y
        out['g12_%s_Nj' % tag] = data.Nj
        out['g12_%s_phi_true' % tag] = data.true_values['phi']
        out['g12_%s_Q0diag' % tag] = np.diag(Q0).copy()
        out['g12_%s_r0' % tag] = r0


def main():
    util, method, tmp = import_reference()
    from models import m1b, m4b
    models = {'m1b': m1b, 'm4b': m4b}
    if '--only-extra' in sys.argv or '--only-gauss' in sys.argv:
        try:
            sim = {}
            g11_gauss_simulators(sim)
            np.savez_compressed(os.path.join(HERE, 'simulators_extra.npz'), **sim)
            print('simulators_extra.npz', os.path.getsize(os.path.join(HERE, 'simulators_extra.npz')), 'bytes')
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    try:
        alg = {}
        g1_invert(util, alg)
        g2_olse(util, alg)
        g3_cavity(method, alg)
        g4_tilted(method, alg)
        g5_master_init(method, alg)
        g8_seeds(alg)
        np.savez_compressed(os.path.join(HERE, 'algebra.npz'), **alg)
        run = {}
        g6_run(method, run, models)
        np.savez_compressed(os.path.join(HERE, 'master_run.npz'), **run)
        swp = {}
        g10_distribute_groups(util, swp)
        g9_damp_sweep(method, util, swp, models)
        np.savez_compressed(os.path.join(HERE, 'damp_sweep.npz'), **swp)
        sim = {}
        g7_simulators(sim, models)
        np.savez_compressed(os.path.join(HERE, 'simulators.npz'), **sim)
        simg = {}
        g11_gauss_simulators(simg)
        np.savez_compressed(os.path.join(HERE, 'simulators_extra.npz'), **simg)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for f in ('algebra.npz', 'master_run.npz', 'simulators.npz', 'damp_sweep.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')


if __name__ == '__main__':
    main()
