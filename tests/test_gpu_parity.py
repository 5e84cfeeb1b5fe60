"""Parity tests proper: the HIP path (through the C ABI) against the oracle and
the golden vectors captured from the reference.  Need a real MI355X."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import epstan_amd                                    # noqa: E402
from epstan_amd import _lib, models, util            # noqa: E402
from epstan_amd.engine import HipEngine, QI, QI2, DQI   # noqa: E402
from epstan_amd.method import Master, Worker         # noqa: E402
from oracle import ep_oracle as eo                   # noqa: E402
from oracle import nuts_oracle as no                 # noqa: E402
import injectors                                     # noqa: E402
from conftest import record_slack                    # noqa: E402

# deterministic stages: SURVEY.md §8c tolerance (QR vs Cholesky-of-scatter and
# reduction order differ by cond * eps)
RTOL, ATOL = 1e-9, 1e-10


@pytest.fixture(scope='module')
def alg(golden_dir):
    return np.load(os.path.join(golden_dir, 'algebra.npz'))


@pytest.fixture(scope='module')
def runs(golden_dir):
    return np.load(os.path.join(golden_dir, 'master_run.npz'))


def test_native_library_is_loaded():
    lib = _lib.load()
    assert _lib.device_count() >= 1
    with open('/proc/self/maps') as f:
        assert 'libepx.so' in f.read()


# ------------------------------------------------------------------ RNG
def test_rng_stream_matches_oracle():
    import ctypes
    lib = _lib.load()
    out = np.zeros(4)
    for (seed, chain, t, kind, a, b) in [(1, 0, 0, 0, 0, 0), (327741615, 3, 17, 4, (5 << 16) | 13, 2),
                                         (2**31 - 2, 1, 200, 1, 25, 0), (12345678901, 2, 7, 5, 3, 9)]:
        _lib.check(lib.epx_rng_probe(0, seed, chain, t, kind, a, b, _lib.dptr(out)))
        ref = no.rng_probe(seed, chain, t, kind, a, b)
        assert out[0] == ref[0] and out[1] == ref[1]          # Philox + u01: bit exact
        np.testing.assert_allclose(out[2:], ref[2:], rtol=1e-13, atol=1e-15)   # libm vs ocml


# ------------------------------------------------------------------ util
@pytest.mark.parametrize('d', [5, 17, 33, 66])
def test_invert_normal_params_golden(alg, d):
    Q, r = util.invert_normal_params(alg['g1_S_%d' % d], alg['g1_m_%d' % d])
    assert Q.flags['F_CONTIGUOUS']
    np.testing.assert_allclose(Q, alg['g1_Q_%d' % d], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(r, alg['g1_r_%d' % d], rtol=RTOL, atol=ATOL)
    np.testing.assert_array_equal(Q, Q.T)                     # copy_triu_to_tril
    Q, r = util.invert_normal_params(alg['g1_U_%d' % d], alg['g1_m_%d' % d], cho_form=True)
    np.testing.assert_allclose(Q, alg['g1_Qc_%d' % d], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(r, alg['g1_rc_%d' % d], rtol=RTOL, atol=ATOL)
    # in-place / out_ forms (util.py:88-108)
    A = alg['g1_S_%d' % d].copy(order='F'); b = alg['g1_m_%d' % d].copy()
    oA, ob = util.invert_normal_params(A, b, out_A='in-place', out_b='in-place')
    assert oA is A and ob is b
    np.testing.assert_allclose(A, alg['g1_Q_%d' % d], rtol=RTOL, atol=ATOL)
    C = np.ascontiguousarray(alg['g1_S_%d' % d])              # C-order input is transposed
    np.testing.assert_allclose(util.invert_normal_params(C)[0], alg['g1_Q_%d' % d], rtol=RTOL, atol=ATOL)


def test_invert_not_posdef_raises(alg):
    with pytest.raises(np.linalg.LinAlgError):
        util.invert_normal_params(alg['g1_bad'], np.zeros(6))
    with pytest.raises(np.linalg.LinAlgError):
        util.invert_normal_params(np.zeros((3, 3), order='F'), cho_form=True)
    big = np.asfortranarray(np.eye(140) * 2.0 + 0.01)         # d = 140: global-workspace path
    Q, _ = util.invert_normal_params(big)
    np.testing.assert_allclose(Q, np.linalg.inv(big), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('d', [5, 17, 33])
@pytest.mark.parametrize('n', [100, 400])
def test_olse_golden(alg, d, n):
    key = '%d_%d' % (d, n)
    np.testing.assert_allclose(util.olse(alg['g2_S_' + key], n), alg['g2_none_' + key], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(util.olse(alg['g2_S_' + key], n, P=alg['g2_P_' + key]),
                               alg['g2_prior_' + key], rtol=RTOL, atol=ATOL)


# ------------------------------------------------------------------ Worker
def test_worker_cavity_golden(alg):
    rng = np.random.RandomState(13)
    X = rng.randn(20, 4); y = (rng.rand(20) < 0.5).astype(int)
    w = Worker(0, 'none/m4b_sg', 10, X, y)
    assert w.cavity(alg['g3_Q'], alg['g3_r'], alg['g3_Qi'], alg['g3_ri']) == bool(alg['g3_flag'])
    assert w.phase == int(alg['g3_phase'])
    np.testing.assert_allclose(w.Mat, alg['g3_Mat'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(w.vec, alg['g3_vec'], rtol=RTOL, atol=ATOL)
    assert not w.cavity(alg['g3_Q'], alg['g3_r'], alg['g3_Qi_bad'], alg['g3_ri'])
    assert w.phase == int(alg['g3_phase_bad']) == 0
    with pytest.raises(RuntimeError):
        w.tilted(np.zeros((10, 10), order='F'), np.zeros(10))


def g4_samples(seed, S, d):
    rng = np.random.RandomState(seed)
    mix = np.eye(d) + 0.3 * rng.randn(d, d) / np.sqrt(d)
    shift = rng.randn(d)
    return np.asfortranarray(rng.randn(S, d).dot(mix) + shift)


@pytest.mark.parametrize('d,model,D', [(5, 'm1b_sg', 4), (10, 'm4b_sg', 4), (17, 'm1b_sg', 16), (34, 'm4b_sg', 16)])
@pytest.mark.parametrize('est', ['sample', 'olse'])
def test_tilted_moment_stage_golden(alg, d, model, D, est):
    """Worker.tilted's moment stage with the reference's injected draws (G4)."""
    rng = np.random.RandomState(14)
    X = rng.randn(40, D); y = (rng.rand(40) < 0.5).astype(int)
    eng = HipEngine(model, X, y, np.array([0, 20, 40]))
    assert eng.d == d
    eng.set_global(alg['g4_Q_%d' % d], alg['g4_r_%d' % d])
    samp = g4_samples(int(alg['g4_seed_%d' % d]), 400, d)
    both = np.asfortranarray(np.stack([samp, samp[::-1]], axis=2))
    flags = eng.moments_batch(both, est)
    key = '%s_%d' % (est, d)
    assert flags.tolist() == [bool(alg['g4_flag_' + key])] * 2
    for k in range(2):      # reversing the draw order must not matter beyond rounding
        dQ, dr = eng.get_site(DQI, k)
        np.testing.assert_allclose(dQ, alg['g4_dQi_' + key], rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(dr, alg['g4_dri_' + key], rtol=RTOL, atol=1e-9)
        Mat, vec, nsamp = eng.get_tilted(k)
        np.testing.assert_allclose(vec, alg['g4_vec_' + key], rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(Mat, alg['g4_scatter_%d' % d], rtol=RTOL, atol=1e-9)
        assert nsamp == int(alg['g4_nsamp_' + key])


def test_tilted_singular_scatter_fails_softly():
    rng = np.random.RandomState(3)
    X = rng.randn(20, 4); y = (rng.rand(20) < 0.5).astype(int)
    eng = HipEngine('m1b_sg', X, y, np.array([0, 10, 20]))
    eng.set_global(np.eye(5), np.zeros(5))
    samp = rng.randn(400, 5, 2)
    samp[:, 0, 1] = 1.25                                    # constant column -> singular (method.py:460-465)
    flags = eng.moments_batch(np.asfortranarray(samp), 'sample')
    assert flags.tolist() == [True, False]
    dQ, dr = eng.get_site(DQI, 1)
    assert not dQ.any() and not dr.any()


# ------------------------------------------------------------------ Master.run trajectories (G6)
def _master(runs, scenario, df0, nsites=4, prec_estim='sample', factor=60.0, **kw):
    Nj = runs['g6_Nj'][:nsites]
    nrow = int(Nj.sum())
    M = Master('some/dir/m1b_sg', runs['g6_X'][:nrow], runs['g6_y'][:nrow], site_sizes=Nj,
               prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']},
               A_k={'site_id': np.arange(nsites)}, chains=4, iter=200, df0=df0,
               prec_estim=prec_estim, **kw)
    M._sample_injector = injectors.GaussianTilted(scenario, factor=factor)
    return M


@pytest.mark.parametrize('tag,scenario,niter,df0,nsites,est,factor', [
    ('smooth', 'smooth', 12, 0.5, 4, 'sample', 60.0),
    ('smooth_olse', 'smooth', 6, 0.5, 4, 'olse', 60.0),
    ('decay', 'wide_first', 4, 1.0, 3, 'sample', 60.0),
    ('allfail', 'degenerate', 3, 0.5, 4, 'sample', 60.0),
    ('badprior', 'wide_all', 3, 1.0, 4, 'sample', 400.0),
])
def test_master_run_reference_trajectory_on_gpu(runs, tag, scenario, niter, df0, nsites, est, factor):
    M = _master(runs, scenario, df0, nsites, est, factor)
    assert isinstance(M.engine, HipEngine)
    info, (m_s, S_s), (st, ms, rh, ot) = M.run(niter, verbose=False, return_analytics=True, seed=1)
    assert info == int(runs['g6_%s_info' % tag])
    assert M.iter == int(runs['g6_%s_iter' % tag])
    np.testing.assert_allclose(m_s, runs['g6_%s_m' % tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(S_s, runs['g6_%s_S' % tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(M.Qi, runs['g6_%s_Qi' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(M.ri, runs['g6_%s_ri' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(M.Q, runs['g6_%s_Q' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(M.r, runs['g6_%s_r' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_array_equal([w.phase for w in M.workers], runs['g6_%s_phase' % tag])


def test_master_resume_and_force_pd_on_gpu(runs):
    M = _master(runs, 'smooth', 0.5)
    M.run(2, verbose=False, seed=5)
    info, (m_s, S_s) = M.run(2, verbose=False, seed=6)
    np.testing.assert_allclose(m_s, runs['g6_resume_m'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(S_s, runs['g6_resume_S'], rtol=1e-8, atol=1e-10)
    S, m = M.cur_approx()
    np.testing.assert_allclose(S, S_s[-1].T, rtol=1e-9)
    # force-pd branch vs the oracle restatement
    Mf = _master(runs, 'wide_first', 1.0, nsites=3, df_treshold=0.9)
    info = Mf.run(2, verbose=False, calc_moments=False, seed=1)
    inj = injectors.GaussianTilted('wide_first')
    Nj = runs['g6_Nj'][:3]; nrow = int(Nj.sum())
    O = eo.OracleMaster(runs['g6_X'][:nrow], runs['g6_y'][:nrow], Nj,
                        lambda data, sp: (inj(data, sp), [{}] * 4, 0.25, 0.125, 1.0625),
                        prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']},
                        A_k={'site_id': np.arange(3)}, chains=4, iter=200, df0=1.0, df_treshold=0.9)
    assert O.run(2, seed=1)[0] == info == 0
    np.testing.assert_allclose(Mf.Qi, O.Qi, rtol=1e-8, atol=1e-9)


def test_find_damp_access_pattern_on_gpu(runs):
    M = _master(runs, 'smooth', 0.5)
    posdefs = [w.tilted(M.dQi[:, :, k], M.dri[:, k], seed=11 + k) for k, w in enumerate(M.workers)]
    assert all(posdefs)
    df = 0.3
    np.add(M.Qi, np.multiply(df, M.dQi, out=M.Qi2), out=M.Qi2)
    np.add(M.ri, np.multiply(df, M.dri, out=M.ri2), out=M.ri2)
    np.add(M.Qi2.sum(2, out=M.Q), M.Q0, out=M.Q)
    np.add(M.ri2.sum(1, out=M.r), M.r0, out=M.r)
    for k, w in enumerate(M.workers):
        assert w.cavity(M.Q, M.r, M.Qi2[:, :, k], M.ri2[:, k])
        Mat, vec, ok = eo.cavity(M.Q, M.r, M.Qi2[:, :, k], M.ri2[:, k])
        np.testing.assert_allclose(w.Mat, Mat, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(w.vec, vec, rtol=1e-9, atol=1e-12)


# ------------------------------------------------------------------ site log-density
def _site_problem(model, D, n, seed, K=2, tight=1.0):
    """K random sites; `tight` scales the cavity precision: a dominant Gaussian
    cavity gives short, non-chaotic trajectories (used where whole runs are
    compared draw by draw)."""
    rng = np.random.RandomState(seed)
    X = rng.randn(K * n, D) * 1.5
    y = (rng.rand(K * n) < 0.6).astype(int)
    if no.is_gauss(model):                      # m*a_sg: real responses
        y = 0.4 + X.dot(rng.randn(D) * 0.5) + 0.8 * rng.randn(K * n)
    d, P = no.dims(model, D)
    Oms, mus = [], []
    for k in range(K):
        A = rng.randn(d, d + 3)
        Oms.append((A.dot(A.T) / (d + 3) + 0.5 * np.eye(d)) * tight)
        mus.append(0.5 * rng.randn(d))
    return X, y, np.arange(K + 1) * n, np.array(Oms), np.array(mus), d, P


def _engine_with_cavity(model, X, y, k_lim, Oms, mus):
    """Engine whose device cavities are (Oms[k], mus[k]) up to rounding; returns
    the engine and the exact device cavities for the oracle."""
    eng = HipEngine(model, X, y, k_lim)
    d = eng.d
    for k in range(eng.K):
        # cavity = Q - Qi with Q = Om + I, Qi = I ; mean = Om^-1 (r - ri)
        Q = Oms[k] + np.eye(d)
        r = Oms[k].dot(mus[k])
        assert eng.cavity_site(k, Q, r, np.eye(d), np.zeros(d))
    Om_dev = np.stack([eng.get_cavity(k)[0] for k in range(eng.K)])
    mu_dev = np.stack([eng.get_cavity(k)[1] for k in range(eng.K)])
    return eng, Om_dev, mu_dev


@pytest.mark.parametrize('model', ['m1b_sg', 'm2b_sg', 'm3b_sg', 'm4b_sg', 'm5b_sg',
                                   'm1a_sg', 'm2a_sg', 'm3a_sg', 'm4a_sg', 'm5a_sg'])
@pytest.mark.parametrize('D,n', [(3, 7), (4, 50), (16, 200), (21, 333), (32, 500)])
def test_logdensity_gradient_matches_oracle(model, D, n):
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 100 + D)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    rng = np.random.RandomState(5)
    layouts = (2, 1)
    if no.is_gauss(model):
        # the Gaussian-likelihood family is built for sites that are fully LDS resident
        dp = [c for c in (4, 8, 16, 32) if c >= D][0]
        lds1 = n * dp * 8 + n * 8 + d * d * 8                       # rows, responses, cavity precision
        nv = (P + 63) // 64
        lds2 = lds1 + 2 * 4 * (64 * (1 + nv) + 2) * 8 + 10 * (4 * nv * 64 + 2) * 8 + 2 * (7 * nv * 64 + 72) * 8
        if lds1 > 160 * 1024:
            layouts = (0,)                      # rows beyond the LDS: streamed (layout 3) since round 2
        elif lds2 > 160 * 1024:
            layouts = (1,)                      # one workgroup per chain needs exchange, tree stack and mailbox in LDS too
    for k in range(2):
        for trial in range(3):
            theta = rng.randn(P) * (0.2 + 0.4 * trial)
            lo, hi = k_lim[k], k_lim[k + 1]
            lp_o, g_o = no.logdensity_grad(model, X[lo:hi], y[lo:hi], mu_dev[k], Om_dev[k], theta)
            for layout in layouts:
                lp, g = eng.logdensity_grad(k, theta, layout=layout)
                assert layout != 0 or eng.last_layout() == 3
                assert abs(lp - lp_o) <= 1e-11 * max(1.0, abs(lp_o))
                np.testing.assert_allclose(g, g_o, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(g_o).max()))


# ------------------------------------------------------------------ sampler vs oracle, draw by draw
@pytest.mark.parametrize('model,D,n,layout,it,tight', [
    ('m1b_sg', 4, 50, 2, 60, 100.), ('m4b_sg', 4, 50, 2, 60, 100.), ('m4b_sg', 4, 50, 1, 60, 100.),
    ('m5b_sg', 4, 50, 1, 60, 100.), ('m2b_sg', 6, 80, 2, 60, 100.), ('m3b_sg', 6, 80, 1, 60, 100.),
    ('m4b_sg', 16, 200, 2, 44, 1000.), ('m4b_sg', 16, 200, 1, 44, 1000.), ('m4b_sg', 32, 120, 1, 44, 1000.),
    ('m1b_sg', 32, 300, 2, 60, 100.), ('m3b_sg', 11, 64, 2, 60, 100.), ('m2b_sg', 32, 100, 1, 44, 1000.),
    # layouts 5 and 7: state wave + row wave per chain; state waves + a row team on the matrix pipe (the default at C3 / C4)
    ('m4b_sg', 32, 120, 5, 44, 1000.), ('m4b_sg', 16, 200, 5, 44, 1000.),
    ('m4b_sg', 32, 120, 7, 44, 1000.), ('m4b_sg', 16, 200, 7, 44, 1000.), ('m4b_sg', 32, 500, 7, 44, 1000.),
    ('m1b_sg', 16, 120, 7, 60, 100.), ('m5b_sg', 21, 333, 7, 44, 1000.), ('m3b_sg', 32, 300, 7, 44, 1000.), ('m2b_sg', 9, 77, 7, 60, 100.),
    # layout 4: chains in lock step, rows resident, MFMA products
    ('m4b_sg', 4, 50, 4, 60, 100.), ('m1b_sg', 16, 200, 4, 60, 100.), ('m4b_sg', 32, 120, 4, 44, 1000.),
    ('m5b_sg', 9, 77, 4, 60, 100.), ('m3b_sg', 32, 300, 4, 44, 1000.), ('m2b_sg', 21, 333, 4, 44, 1000.),
    # Gaussian-likelihood family (real responses, phi[0] = log sigma)
    ('m1a_sg', 4, 50, 2, 60, 100.), ('m4a_sg', 4, 50, 1, 60, 100.), ('m4a_sg', 16, 200, 2, 44, 1000.),
    ('m5a_sg', 7, 33, 1, 60, 100.), ('m2a_sg', 6, 80, 2, 60, 100.), ('m3a_sg', 16, 120, 1, 44, 1000.),
])
def test_nuts_full_run_matches_oracle(model, D, n, layout, it, tight):
    """Whole site updates (random init, step-size search, dual averaging, metric
    window, sampling): same Philox stream, same decisions, so the device draws
    follow the C restatement.  HMC trajectories amplify the ~1e-16 differences of
    reduction order and libm/ocml; a dominant cavity keeps that growth small
    enough to compare every draw of the run at 2e-5 (a single differing decision
    would give O(1) differences; generic cavities: next test).  iter >= 44 so that
    the warm-up contains a metric window (update + second step-size search)."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 7 + D, K=3, tight=tight)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.array([101, 202, 303], dtype=np.int64)
    opts = HipEngine.sampler_opts(chains=4, iter=it, warmup=None, init='random', layout=layout)
    stats, ms = eng.sample_batch(seeds, opts)
    draws_o, last_o, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=4, iter=it)
    cs = eng.get_chain_stats(4)
    nk = it // 2
    n_full = 0
    for k in range(3):
        dev = eng.get_draws(k, all_params=True)
        ref = draws_o[k].reshape(-1, P)
        phi = eng.get_draws(k)
        np.testing.assert_array_equal(phi, dev[:, :d])                     # (S, dphi) F-order view
        assert phi.flags['F_CONTIGUOUS'] and phi.shape == (4 * nk, d)
        err = np.abs(dev - ref).reshape(4, nk, P).max(axis=2) / max(1.0, np.abs(ref).max())
        for c in range(4):
            # warm-up (search, dual averaging, metric window) and the first kept draws must agree;
            # later a chain may part ways once rounding differences have been amplified past a
            # decision threshold (chaotic trajectories) -- most chains never do in these runs
            assert np.all(err[c, :5] < 1e-3), (k, c, err[c, :5])      # a wrong decision gives O(0.1-1)
            if np.all(err[c] < 1e-4):
                n_full += 1
                assert cs[k, c, 2] == st_o[k, c, 2]                              # same leapfrog count
                assert cs[k, c, 3] == st_o[k, c, 3]                              # same gradient count
                np.testing.assert_allclose(cs[k, c, 0], st_o[k, c, 0], rtol=1e-5)    # step-size path
                np.testing.assert_allclose(cs[k, c, 5], st_o[k, c, 5], rtol=1e-4)    # accept_stat
    # (layouts 5 and 7, the kernels the bench times: 12 of 12 in every case through round 5.  Round 6 shortened the logistic
    # arithmetic -- a degree-11 near-minimax exp, a third-order reciprocal step, magic-number rounding: the same accuracy,
    # other last bits -- and ONE chain of the 108 of these nine cases (m2b, D = 9) now parts from the oracle before the
    # end, as one in a few hundred does under every layout: 11 of 12 per case is the bound, the slack table shows the count)
    need = 11 if layout in (5, 7) else 9
    record_slack('full run vs oracle %s D=%d n=%d layout %d: chains equal to the end' % (model, D, n, layout), n_full, '>= %d' % need, 12)
    assert n_full >= need, n_full                                          # of 12 (site, chain) runs
    if n_full == 12:
        np.testing.assert_allclose(stats[:, 2], st_o[:, :, 2].sum(1))
        rh = [max(no.split_rhat(draws_o[k, :, :, e]) for e in range(P)) for k in range(3)]
        rh = np.array(rh)
        sane = np.isfinite(rh) & (rh < 1e3)        # a coordinate that never moved gives 0/0-like ratios
        np.testing.assert_allclose(stats[sane, 1], rh[sane], rtol=1e-4)
        assert np.all(stats[~sane, 1] > 1e3)
        np.testing.assert_allclose(stats[:, 0], st_o[:, :, 0].mean(1), rtol=1e-5)


@pytest.mark.parametrize('model,D,n,layout', [
    ('m4b_sg', 4, 50, 1), ('m4b_sg', 4, 50, 2), ('m5b_sg', 8, 64, 2), ('m3b_sg', 16, 100, 1),
    ('m4b_sg', 16, 200, 2), ('m4b_sg', 16, 200, 1), ('m4b_sg', 32, 500, 1), ('m4b_sg', 32, 500, 2),
    ('m1b_sg', 32, 500, 1), ('m4b_sg', 32, 500, 4), ('m4b_sg', 16, 200, 4), ('m1b_sg', 32, 500, 4), ('m3b_sg', 7, 45, 4),
    ('m4b_sg', 32, 500, 5), ('m1b_sg', 32, 500, 5),           # the kernel round 2's bench timed, at its own site shape
    ('m4b_sg', 32, 500, 7), ('m1b_sg', 32, 500, 7), ('m4b_sg', 16, 200, 7), ('m5b_sg', 32, 300, 7), ('m3b_sg', 21, 100, 7),   # ... and round 3's
    ('m4a_sg', 16, 200, 2), ('m4a_sg', 16, 200, 1), ('m1a_sg', 32, 300, 1), ('m3a_sg', 8, 64, 2),
])
def test_nuts_transitions_match_oracle_teacher_forced(model, D, n, layout):
    """Generic (wide, funnel-shaped) tilted distributions up to BASELINE config
    C3's site size (D=32, n_j=500): trees of hundreds of leapfrogs are chaotic, so
    each transition is compared on its own -- device and oracle start from the
    same typical-set point with the same step size and metric and must build the
    same tree (same leapfrog count, same multinomial picks) and return the same
    draw to 1e-6."""
    K = 2
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 31 + D, K=K)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.array([11, 12], dtype=np.int64)
    opts = HipEngine.sampler_opts(chains=4, iter=100, init='random', layout=layout)
    eng.sample_batch(seeds, opts)                       # adapted state + typical-set points
    cs = eng.get_chain_stats(4)
    draws = np.stack([eng.get_draws(k, True).reshape(4, 50, P) for k in range(K)])
    eps = cs[:, :, 1]
    inv_e = np.repeat(draws.reshape(K, -1, P).var(axis=1)[:, None, :], 4, axis=1) + 1e-3
    nbad = 0
    for start, t_off in [(10, 0), (25, 7), (49, 123)]:
        q0 = draws[:, :, start, :]
        out, st = eng.nuts_transitions(seeds, q0, eps, inv_e, nt=1, t_offset=t_off, layout=layout)
        ref, st_o = no.nuts_transitions(model, X, y, k_lim, mu_dev, Om_dev, seeds, q0, eps, inv_e,
                                        nt=1, t_offset=t_off)
        err = np.abs(out - ref).max(axis=(2, 3)) / np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
        same_tree = st[:, :, 2] == st_o[:, :, 2]
        # a decision landing within rounding of its threshold may legitimately flip: allow one
        nbad += int(np.sum(~same_tree | (err > 1e-6)))
        assert np.all(err[same_tree] < 1e-6), err
        assert st_o[:, :, 2].min() >= 1
    allow = 0 if layout in (5, 7) else 1        # (the bench's kernels: observed 0 of 24 in every case since they exist)
    record_slack('teacher-forced transitions %s D=%d n=%d layout %d: transitions that differ' % (model, D, n, layout), nbad, '<= %d' % allow, 24)
    assert nbad <= allow, nbad


@pytest.mark.parametrize('model,D,n', [('m4b_sg', 16, 200), ('m1b_sg', 4, 50), ('m4b_sg', 32, 150), ('m5b_sg', 7, 33),
                                       ('m4b_sg', 32, 500)])       # last: Omega in L2 and the tree stack in HBM
def test_speculative_bookkeeping_wave_gives_identical_draws(model, D, n):
    """Layout 2 with the tree bookkeeping on a fifth wave (gradient waves integrate ahead and drop
    what they integrated past a change of state) against the sequential kernel: bit-identical
    draws, statistics and gradient counts."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 17 + D, K=3, tight=3.0)
    eng, _, _ = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.array([41, 42, 43], dtype=np.int64)
    res = []
    for flags in (1, 0):
        opts = HipEngine.sampler_opts(chains=4, iter=50, init='random', layout=2, flags=flags)
        stats, ms = eng.sample_batch(seeds, opts)
        res.append((np.stack([eng.get_draws(k, True) for k in range(3)]), eng.get_chain_stats(4), stats))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    assert res[0][1][:, :, 2].max() >= 7           # trees of several doublings were built


def test_nuts_layouts_agree_and_are_deterministic():
    X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 8, 90, 3, K=2, tight=100.0)
    eng, _, _ = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
    seeds = np.array([5, 6], dtype=np.int64)
    out = {}
    for layout in (1, 2, 1):
        opts = HipEngine.sampler_opts(chains=4, iter=44, init='random', layout=layout)
        eng.sample_batch(seeds, opts)
        out.setdefault(layout, []).append(np.stack([eng.get_draws(k, True) for k in range(2)]))
    np.testing.assert_array_equal(out[1][0], out[1][1])                    # bitwise reproducible
    # 1 wave per chain vs 4 waves per chain: same algorithm, different summation order
    err = np.abs(out[1][0] - out[2][0]).reshape(2, 4, 22, P).max(axis=3)
    assert np.all(err[:, :, :5] < 1e-3)
    assert np.sum(np.all(err < 1e-4, axis=2)) >= 6


@pytest.mark.gpu
@pytest.mark.parametrize('model,D,n,layout', [('m4b_sg', 8, 90, 1), ('m4b_sg', 8, 90, 2), ('m4b_sg', 40, 70, 3), ('m4b_sg', 8, 90, 4),
                                              ('m4b_sg', 12, 90, 5), ('m4b_sg', 12, 90, 6), ('m4b_sg', 27, 150, 5)])
def test_site_order_hint_does_not_change_results(model, D, n, layout):
    """epx_set_site_order only permutes which workgroup takes which site."""
    K = 5
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 11, K=K, tight=30.0)
    eng, _, _ = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.arange(K, dtype=np.int64) + 3
    opts = HipEngine.sampler_opts(chains=4, iter=30, init='random', layout=layout)
    eng.sample_batch(seeds, opts)
    ref = np.stack([eng.get_draws(k, True) for k in range(K)])
    ref_stats = eng.get_chain_stats(4)
    eng.set_site_order([3, 0, 4, 2, 1])
    eng.sample_batch(seeds, opts)
    np.testing.assert_array_equal(np.stack([eng.get_draws(k, True) for k in range(K)]), ref)
    np.testing.assert_array_equal(eng.get_chain_stats(4), ref_stats)
    with pytest.raises(Exception):
        eng.set_site_order([0, 0, 1, 2, 3])
    eng.set_site_order(None)
    eng.sample_batch(seeds, opts)
    np.testing.assert_array_equal(np.stack([eng.get_draws(k, True) for k in range(K)]), ref)
    assert eng.last_layout() == layout


@pytest.mark.gpu
@pytest.mark.parametrize('model,D,n,K', [('m1b_sg', 4, 30, 330), ('m4b_sg', 32, 300, 200)])
def test_split_launch_runs_lead_sites_one_workgroup_per_chain(model, D, n, K):
    """epx_set_site_split: the leading sites of the order give exactly the draws of layout 2, the
    others those of the layout the library picks for them (7, or 1 for D <= 8), whatever the split (the two launches share
    every buffer).  K: enough sites for the library to pick one workgroup per site by itself (320
    when two layout-2 workgroups fit a CU): layout 5 where the shape is instantiated, else layout 1."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 5, K=K, tight=30.0)
    eng, _, _ = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.arange(K, dtype=np.int64) + 11
    ref = {}
    auto = 7 if D >= 9 else 1            # what the library picks for the rest: the row-team kernel where it is instantiated
    for layout in (1, 2, auto):
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=12, init='random', layout=layout, max_depth=6))
        assert eng.last_layout() == layout and eng.last_split() == 0
        ref[layout] = (np.stack([eng.get_draws(k, True) for k in range(K)]), eng.get_chain_stats(4))
    order = np.random.RandomState(1).permutation(K)
    eng.set_site_order(order)
    opts = HipEngine.sampler_opts(chains=4, iter=12, init='random', max_depth=6)
    for n_lead in (7, 1, 150):
        eng.set_site_split(n_lead)
        stats, ms = eng.sample_batch(seeds, opts)
        m = eng.last_split()
        assert eng.last_layout() == auto and 1 <= m <= n_lead
        assert m == n_lead or n_lead == 150          # clamped to half of the CUs
        dr = np.stack([eng.get_draws(k, True) for k in range(K)])
        cs = eng.get_chain_stats(4)
        lead, rest = order[:m], order[m:]
        np.testing.assert_array_equal(dr[lead], ref[2][0][lead])
        np.testing.assert_array_equal(cs[lead], ref[2][1][lead])
        np.testing.assert_array_equal(dr[rest], ref[auto][0][rest])
        np.testing.assert_array_equal(cs[rest], ref[auto][1][rest])
    eng.set_site_split(0)
    eng.sample_batch(seeds, opts)
    assert eng.last_split() == 0
    np.testing.assert_array_equal(np.stack([eng.get_draws(k, True) for k in range(K)]), ref[auto][0])
    # an explicit layout is never split
    eng.set_site_split(5)
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=12, init='random', layout=1, max_depth=6))
    assert eng.last_split() == 0


@pytest.mark.gpu
def test_nuts_warm_start_and_thin():
    """init_prev (method.py:404-406): the next call starts at the last draws."""
    X, y, k_lim, Oms, mus, d, P = _site_problem('m1b_sg', 4, 60, 9, K=2, tight=40.0)
    eng, Om_dev, mu_dev = _engine_with_cavity('m1b_sg', X, y, k_lim, Oms, mus)
    seeds = np.array([1, 2], dtype=np.int64)
    with pytest.raises(_lib.EpxError):
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=2, iter=40, init='prev'))
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=2, iter=40, init='0'))
    d0, last0, _ = no.nuts_sites('m1b_sg', X, y, k_lim, mu_dev, Om_dev, seeds, chains=2, iter=40,
                                 init=np.zeros((2, 2, P)))
    assert np.abs(eng.get_draws(0, True) - d0[0].reshape(-1, P)).max() < 1e-6
    seeds2 = np.array([8, 9], dtype=np.int64)
    eng.sample_batch(seeds2, HipEngine.sampler_opts(chains=2, iter=40, thin=1, init='prev'))
    d1, _, _ = no.nuts_sites('m1b_sg', X, y, k_lim, mu_dev, Om_dev, seeds2, chains=2, iter=40, init=last0)
    assert np.abs(eng.get_draws(1, True) - d1[1].reshape(-1, P)).max() < 1e-5


def test_nuts_long_run_moments_match_oracle_statistically():
    """Independent seeds: GPU and oracle estimate the same tilted moments within
    Monte-Carlo error (4 sigma of the MCSE from split chains)."""
    X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 4, 50, 21, K=2)
    eng, Om_dev, mu_dev = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
    opts = HipEngine.sampler_opts(chains=4, iter=3000, warmup=500, init='random')
    stats, _ = eng.sample_batch(np.array([1, 2], dtype=np.int64), opts)
    draws_o, _, _ = no.nuts_sites('m4b_sg', X, y, k_lim, mu_dev, Om_dev, np.array([77, 78]),
                                  chains=4, iter=3000, warmup=500)
    assert stats[:, 1].max() < 1.05
    for k in range(2):
        g = eng.get_draws(k, True)
        o = draws_o[k].reshape(-1, P)
        sd = o.std(0)
        ess = 2000.0                                        # conservative for 10000 NUTS draws
        assert np.all(np.abs(g.mean(0) - o.mean(0)) < 4 * sd * np.sqrt(2 / ess))
        assert np.all(np.abs(g.std(0) / sd - 1) < 4 * np.sqrt(1.0 / ess))


# ------------------------------------------------------------------ full pipeline at BASELINE sizes
@pytest.mark.parametrize('name,J,D,n', [('m4b', 64, 16, 200)])
def test_ep_iterations_at_c2_properties(name, J, D, n):
    """Size-independent invariants of the hot path at config C2 (too large for a
    draw-by-draw oracle run): symmetry, Q = Q0 + sum_k Qi (checksum of the
    reduction), damping linearity of the packed sums, pos.def. moments, R-hat."""
    mod = models.MODELS[name](J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=4, iter=200, df0=models.default_df0(J))
    info, (m_s, S_s), (st, ms, rh, ot) = M.run(3, verbose=False, return_analytics=True, seed=1)
    assert info == 0
    np.testing.assert_allclose(M.Qi, np.swapaxes(M.Qi, 0, 1), rtol=0, atol=1e-9)
    np.testing.assert_allclose(M.Q, M.Qi.sum(2) + M.Q0, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(M.r, M.ri.sum(1) + M.r0, rtol=1e-10, atol=1e-9)
    packed = M.engine.site_sums()
    d = M.dphi
    np.testing.assert_allclose(packed[:d * d].reshape(d, d, order='F'), M.Qi.sum(2), rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(packed[d * d + d:2 * d * d + d].reshape(d, d, order='F'), M.dQi.sum(2),
                               rtol=1e-11, atol=1e-9)
    for i in range(3):
        assert np.all(np.linalg.eigvalsh(S_s[i]) > 0)
    assert np.all(st > 0) and np.all(np.isfinite(rh)) and np.all(rh > 0.9) and np.all(ms > 0)
    # one site of the same iteration against the oracle moment stage
    k = 17
    samp = M.engine.get_draws(k)
    w = M.workers[k]
    # tilted was computed against the PREVIOUS global (Q,r); recompute the moment stage alone
    dQ, dr, mt, scatter, ok = eo.tilted_moments(samp, np.zeros((d, d)), np.zeros(d), 'sample')
    Mat, vec, nsamp = M.engine.get_tilted(k)
    np.testing.assert_allclose(vec, mt, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(Mat, scatter, rtol=1e-9, atol=1e-9)
    assert nsamp == 400


# ------------------------------------------------------------------ shapes and edge cases
@pytest.mark.parametrize('chains,it,thin,init', [(1, 30, 1, 'random'), (2, 40, 3, '0'), (8, 24, 1, 'random'),
                                                 (3, 30, 2, 'random')])
def test_nuts_chain_counts_thin_init(chains, it, thin, init):
    """chains != 4 (find_damp.py uses 8), thinning, init='0' (method.py:154-160, 579-583)."""
    X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 4, 40, 77, K=3, tight=100.0)
    eng, Om_dev, mu_dev = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
    seeds = np.array([3, 4, 5], dtype=np.int64)
    for layout in (1, 2):
        opts = HipEngine.sampler_opts(chains=chains, iter=it, thin=thin, init=init, layout=layout)
        stats, ms = eng.sample_batch(seeds, opts)
        ini = None if init == 'random' else np.zeros((3, chains, P))
        draws_o, _, st_o = no.nuts_sites('m4b_sg', X, y, k_lim, mu_dev, Om_dev, seeds, chains=chains, iter=it,
                                         thin=thin, init=ini)
        nk = draws_o.shape[2]
        assert eng.num_draws() == chains * nk
        for k in range(3):
            dev = eng.get_draws(k, True).reshape(chains, nk, P)
            err = np.abs(dev - draws_o[k]).max(axis=(1, 2))
            assert np.sum(err < 1e-4) >= chains - 1, err


def test_ragged_sites_and_extreme_shapes():
    """Sites of very different sizes (n_j = 1 ... 1500), D = 1, and rows that need several passes
    per lane: the gradient at every site equals the oracle's."""
    rng = np.random.RandomState(5)
    for model, D, sizes in [('m1b_sg', 1, [1, 2, 63, 64, 65, 700]), ('m4b_sg', 5, [1, 257, 1500, 3]),
                            ('m3b_sg', 32, [300, 1, 129]), ('m2b_sg', 7, [10, 512])]:
        N = int(np.sum(sizes))
        X = rng.randn(N, D)
        y = (rng.rand(N) < 0.4).astype(int)
        k_lim = np.concatenate(([0], np.cumsum(sizes)))
        d, P = no.dims(model, D)
        K = len(sizes)
        Oms = np.stack([np.eye(d) * (1.0 + k) for k in range(K)])
        mus = rng.randn(K, d) * 0.3
        eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
        for k in range(K):
            th = rng.randn(P) * 0.5
            lp, g = eng.logdensity_grad(k, th)
            lp_o, g_o = no.logdensity_grad(model, X[k_lim[k]:k_lim[k + 1]], y[k_lim[k]:k_lim[k + 1]],
                                           mu_dev[k], Om_dev[k], th)
            assert abs(lp - lp_o) <= 1e-10 * max(1.0, abs(lp_o))
            np.testing.assert_allclose(g, g_o, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(g_o).max()))
        # and a short sampling run over the ragged batch in both layouts terminates with finite draws
        for layout in (1, 2):
            stats, ms = eng.sample_batch(np.arange(K) + 1, HipEngine.sampler_opts(chains=2, iter=20, layout=layout))
            assert np.all(np.isfinite(stats)) and np.all(stats[:, 7] == 0)


def test_many_sites_batch_and_errors():
    """K = 600 small sites in one launch (more sites than CUs, layout 1 by default), reproducible
    and equal to sampling the same sites one at a time; unsupported shapes fail loudly."""
    rng = np.random.RandomState(8)
    K, n, D = 600, 12, 3
    X = rng.randn(K * n, D); y = (rng.rand(K * n) < 0.5).astype(int)
    eng = HipEngine('m1b_sg', X, y, np.arange(K + 1) * n)
    eng.set_prior(np.eye(4), np.zeros(4))
    eng.set_global(np.eye(4) * 2.0, np.zeros(4))
    assert np.all(eng.cavity_batch(QI))
    seeds = np.arange(K) + 100
    opts = HipEngine.sampler_opts(chains=4, iter=20)
    eng.sample_batch(seeds, opts)
    assert eng.last_layout() == 1
    all_draws = np.stack([eng.get_draws(k, True) for k in (0, 299, 599)])
    for j, k in enumerate((0, 299, 599)):
        eng.sample_batch(seeds[k:k + 1], HipEngine.sampler_opts(chains=4, iter=20, layout=1), k0=k, count=1)
        np.testing.assert_array_equal(eng.get_draws(k, True), all_draws[j])
    # the lock-step layout runs the same algorithm with another summation order
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=20, layout=4))
    assert eng.last_layout() == 4
    for j, k in enumerate((0, 299, 599)):
        np.testing.assert_allclose(eng.get_draws(k, True)[:3], all_draws[j][:3], rtol=1e-6, atol=1e-8)
    with pytest.raises(_lib.EpxError):                      # D > 128: beyond the streaming variant
        e2 = HipEngine('m1b_sg', rng.randn(40, 200), np.zeros(40, dtype=int), np.array([0, 20, 40]))
        e2.set_global(np.eye(201), np.zeros(201)); e2.cavity_batch(QI)
        e2.sample_batch(np.array([1, 2]), opts)
    with pytest.raises(_lib.EpxError):
        HipEngine('m1b_sg', X, np.full(K * n, 2), np.arange(K + 1) * n)      # y must be 0/1
    with pytest.raises(_lib.EpxError):
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=20, warmup=20))


def test_fit_main_and_kl_on_gpu(tmp_path, monkeypatch):
    """fit.py's EP branch end to end on the device, scored with kl_mvn (plot_res.py:41-60)."""
    from epstan_amd import fit
    monkeypatch.setattr(fit, 'RES_PATH', str(tmp_path))
    conf = fit.configurations(J=4, D=4, K=4, npg=50, iter=3, siter=100, run_ep=True, save_res=False)
    res = fit.main('m4b', conf, verbose=False)
    assert res['m_s_ep'].shape == (4, 10)
    S, m = res['S_s_ep'][-1], res['m_s_ep'][-1]
    assert abs(fit.kl_mvn(m, S, m, S)) < 1e-9
    S1 = S * 1.3
    ref = 0.5 * (np.trace(np.linalg.solve(S1, S)) - 10 + np.linalg.slogdet(S1)[1] - np.linalg.slogdet(S)[1])
    assert abs(fit.kl_mvn(m, S, m, S1) - ref) < 1e-9
    # three EP iterations move the approximation away from the prior towards the data
    assert fit.kl_mvn(res['m_s_ep'][-1], res['S_s_ep'][-1], res['m_s_ep'][0], res['S_s_ep'][0]) > 1.0


@pytest.mark.parametrize('name', ['m4b', 'm4a', 'm1a'])
def test_ep_posterior_matches_cpu_path_within_monte_carlo_error(name):
    """north_star's end-to-end bar: the EP posterior mean/covariance from the device path equals
    the CPU (oracle) path's on the same inputs up to Monte-Carlo error.  The tolerance is stated
    relative to the run-to-run spread of the CPU path itself (two seeds): 3x that spread, with
    floors of 0.15 posterior sd for means and 25 % for variances (S = 800 draws per site update,
    8 damped iterations; SURVEY.md 8c: tighter claims are not meaningful)."""
    from oracle.engine_oracle import OracleEngine
    mod = models.MODELS[name](4, 4, 50)         # m4b: the paper's model; m*a: Gaussian-likelihood family
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()

    def run(seed, **kw):
        M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
                   chains=4, iter=400, df0=0.5, **kw)
        info, (m_s, S_s) = M.run(8, verbose=False, seed=seed)
        assert info == 0
        return m_s[-1], S_s[-1]

    cpu = lambda m, X, y, kl: OracleEngine(m, X, y, kl)
    m_g, S_g = run(1)
    m_c1, S_c1 = run(1, _engine_factory=cpu)
    m_c2, S_c2 = run(2, _engine_factory=cpu)
    sd = np.sqrt(np.diag(S_c1))
    spread_m = np.abs(m_c1 - m_c2) / sd
    spread_v = np.abs(np.diag(S_c1) / np.diag(S_c2) - 1)
    tol_m = max(0.15, 3 * spread_m.max())
    tol_v = max(0.25, 3 * spread_v.max())
    assert np.all(np.abs(m_g - m_c1) / sd < tol_m), (np.abs(m_g - m_c1) / sd, tol_m)
    assert np.all(np.abs(np.diag(S_g) / np.diag(S_c1) - 1) < tol_v), tol_v
    # and EP actually learned something: the posterior is much tighter than the prior
    assert np.all(np.diag(S_g) < 0.8 * np.diag(np.linalg.inv(Q0)))


# ------------------------------------------------------------------ streaming sampler (layout 3)
@pytest.mark.parametrize('model,D,n', [('m4b_sg', 40, 150), ('m1b_sg', 64, 100), ('m3b_sg', 100, 257),
                                       ('m2b_sg', 128, 64), ('m5b_sg', 33, 70), ('m4b_sg', 128, 2000)])
def test_streaming_gradient_matches_oracle(model, D, n):
    """Sites beyond the LDS-resident kernel (D > 32, up to config C5's D = 128, n_j = 2000):
    the streaming kernel's log density and gradient equal the oracle's."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 300 + D, K=2)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    rng = np.random.RandomState(6)
    for k in range(2):
        theta = rng.randn(P) * 0.3
        lp, g = eng.logdensity_grad(k, theta)
        lo, hi = k_lim[k], k_lim[k + 1]
        lp_o, g_o = no.logdensity_grad(model, X[lo:hi], y[lo:hi], mu_dev[k], Om_dev[k], theta)
        assert abs(lp - lp_o) <= 1e-10 * max(1.0, abs(lp_o))
        np.testing.assert_allclose(g, g_o, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(g_o).max()))


@pytest.mark.parametrize('model,D,n,chains', [('m4b_sg', 8, 90, 4), ('m1b_sg', 16, 120, 3), ('m4b_sg', 40, 100, 4),
                                              ('m3b_sg', 64, 80, 2), ('m4b_sg', 20, 300, 6)])
def test_streaming_sampler_equals_resident_and_oracle(model, D, n, chains):
    """Layout 3 (lock-step chains, tiled X) on shapes the resident layouts also handle, and on
    D > 32: whole short site updates against the C oracle, chain by chain."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 40 + D, K=2, tight=300.0)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.array([21, 22], dtype=np.int64)
    it = 44
    opts = HipEngine.sampler_opts(chains=chains, iter=it, init='random', layout=3)
    stats, ms = eng.sample_batch(seeds, opts)
    draws_o, _, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=chains, iter=it)
    cs = eng.get_chain_stats(chains)
    nk = it // 2
    n_full = 0
    for k in range(2):
        dev = eng.get_draws(k, True).reshape(chains, nk, P)
        err = np.abs(dev - draws_o[k]).max(axis=2) / max(1.0, np.abs(draws_o[k]).max())
        for c in range(chains):
            assert np.all(err[c, :3] < 1e-3), (k, c, err[c, :3])
            if np.all(err[c] < 1e-4):
                n_full += 1
                assert cs[k, c, 2] == st_o[k, c, 2] and cs[k, c, 3] == st_o[k, c, 3]
    record_slack('streaming run vs oracle %s D=%d n=%d: chains equal to the end' % (model, D, n), n_full, '>= %d' % ((3 * 2 * chains) // 4), 2 * chains)
    assert n_full >= (3 * 2 * chains) // 4, n_full


def test_streaming_transitions_at_c5_site_size():
    """One site of BASELINE config C5 (D = 128, n_j = 2000, m4b: d = 258, P = 387): single
    transitions from typical-set points against the oracle (teacher forced), plus the moment and
    cavity kernels at d = 258 (global-workspace path)."""
    model, D, n = 'm4b_sg', 128, 2000
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 5, K=1, tight=20.0)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    assert (d, P) == (258, 387)
    seeds = np.array([9], dtype=np.int64)
    stats, ms = eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=30, init='random'))
    cs = eng.get_chain_stats(4)
    draws = eng.get_draws(0, True).reshape(1, 4, 15, P)
    q0 = draws[:, :, -1, :]
    eps = cs[:, :, 1]
    inv_e = np.repeat(draws.reshape(1, -1, P).var(axis=1)[:, None, :], 4, axis=1) + 1e-3
    out, st = eng.nuts_transitions(seeds, q0, eps, inv_e, nt=1, t_offset=3)
    ref, st_o = no.nuts_transitions(model, X, y, k_lim, mu_dev, Om_dev, seeds, q0, eps, inv_e, nt=1, t_offset=3)
    err = np.abs(out - ref).max(axis=(2, 3)) / np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
    same = st[:, :, 2] == st_o[:, :, 2]
    assert np.sum(~same | (err > 1e-6)) <= 1, (err, st[:, :, 2], st_o[:, :, 2])
    # moment stage + cavity at d = 258 against the NumPy oracle
    eng.set_global(np.eye(d) * 3.0, np.zeros(d))
    rng = np.random.RandomState(2)
    samp = rng.randn(600, d, 1) * 0.5 + 0.1
    flags = eng.moments_batch(np.asfortranarray(samp), 'sample')
    dQ, dr = eng.get_site(DQI, 0)
    dQ_o, dr_o, mt, scatter, ok = eo.tilted_moments(samp[:, :, 0], np.eye(d) * 3.0, np.zeros(d), 'sample')
    assert flags[0] and ok
    np.testing.assert_allclose(dQ, dQ_o, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(dr, dr_o, rtol=1e-8, atol=1e-8)


# ---------------------------------------------------------------- damping sweep (find_damp.py:146-173)
def test_damp_sweep_golden_on_gpu(golden_dir):
    z = np.load(os.path.join(golden_dir, 'damp_sweep.npz'))
    runs = np.load(os.path.join(golden_dir, 'master_run.npz'))
    M = _master(runs, 'smooth', 0.5)
    M.run(1, verbose=False, seed=3)
    M.Qi[...] = z['g9_Qi']; M.ri[...] = z['g9_ri']
    M._upload_sites()
    M.engine.set_sites(2, z['g9_dQi'], z['g9_dri'])
    for samp in (z['g9_samp_target'], None):
        res = M.damp_sweep(z['g9_damps'], z['g9_m_target'], z['g9_S_target'], samp)
        np.testing.assert_array_equal(res['global_pd'], z['g9_global_pd'])
        np.testing.assert_array_equal(res['cav_pd'], z['g9_cav_pd'])
        np.testing.assert_allclose(res['mses'], z['g9_mses'], rtol=1e-9, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(res['kls'], z['g9_kls'], rtol=1e-9, atol=1e-9, equal_nan=True)
        if samp is not None:
            np.testing.assert_allclose(res['lls'], z['g9_lls'], rtol=1e-9, atol=1e-7, equal_nan=True)
        else:
            assert np.all(np.isnan(res['lls']))
    # the sweep leaves the site parameters alone
    Qi, ri = M.engine.get_sites(0)
    np.testing.assert_array_equal(Qi, z['g9_Qi'])


@pytest.mark.parametrize('d,K', [(17, 9), (66, 24), (130, 6)])
def test_damp_sweep_matches_oracle_at_larger_sizes(d, K):
    rng = np.random.RandomState(d)
    def spd(scale):
        A = rng.randn(d, d + 5)
        return np.asfortranarray(A.dot(A.T) / (d + 5) * scale)
    Q0 = spd(1.0) + np.eye(d)
    r0 = rng.randn(d)
    Qi = np.zeros((d, d, K), order='F'); ri = np.zeros((d, K), order='F')
    dQi = np.zeros((d, d, K), order='F'); dri = np.zeros((d, K), order='F')
    for k in range(K):
        Qi[:, :, k] = spd(0.5); ri[:, k] = rng.randn(d)
        B = rng.randn(d, d) * 0.15
        dQi[:, :, k] = spd(0.4) - 0.25 * Qi[:, :, k] + 0.5 * (B + B.T); dri[:, k] = rng.randn(d)
    D = d - 1                                   # m1b_sg: dphi = D + 1
    X = rng.randn(K * 3, D); y = (rng.rand(K * 3) < 0.5).astype(int)
    eng = HipEngine('m1b_sg', X, y, np.arange(K + 1) * 3)
    assert eng.d == d
    eng.set_prior(Q0, r0); eng.set_sites(0, Qi, ri); eng.set_sites(2, dQi, dri)
    damps = np.concatenate((np.linspace(0, 1, 33)[1:-1], [2.0, 6.0, 20.0, -3.0]))
    m_t = rng.randn(d) * 0.1; S_t = spd(0.2) + 0.05 * np.eye(d)
    samp = rng.multivariate_normal(m_t, S_t, size=64)
    ref = eo.damp_sweep(Q0, r0, Qi, ri, dQi, dri, damps, m_t, S_t, samp)
    out = eng.damp_sweep(damps, eng.site_sums(), m_t, S_t, samp)
    np.testing.assert_array_equal(out[:, :2], ref[:, :2])
    assert 0 < ref[:, 1].sum() < len(damps)
    np.testing.assert_allclose(out[:, 2:], ref[:, 2:], rtol=1e-8, atol=1e-8, equal_nan=True)


def test_find_damp_driver_on_gpu(tmp_path, monkeypatch):
    from epstan_amd import fit, find_damp
    monkeypatch.setattr(fit, 'RES_PATH', str(tmp_path))
    conf = fit.configurations(J=6, D=4, K=6, npg=30, siter=60, chains=4)
    rng = np.random.RandomState(0)
    tgt = dict(m_target=np.zeros(5), S_target=np.eye(5) * 0.5, samp_target=rng.randn(80, 5) * 0.7)
    out = find_damp.main('m1b', iters=3, target=tgt, conf=conf, seed=2, verbose=False)
    assert out['kls'].shape == (3, 31)
    assert np.isfinite(out['kls']).sum() >= 31            # small damping factors are always admissible
    assert np.all(np.isfinite(out['kls_selected'])) and np.all(out['damps_selected'] > 0)
    # the selected factor's criteria agree with the sweep entry at the same damping factor when it is on the grid
    for it in range(3):
        j = np.argmin(np.abs(out['damps'] - out['damps_selected'][it]))
        if abs(out['damps'][j] - out['damps_selected'][it]) < 1e-12:
            assert abs(out['kls'][it, j] - out['kls_selected'][it + 1]) < 1e-7 * max(1.0, abs(out['kls'][it, j]))


# ---------------------------------------------------------------- multi-group sites (K < J)
def _group_problem(model, D, groups, seed, tight=1.0):
    """Sites with several groups each: `groups` = list (per site) of lists of group sizes."""
    rng = np.random.RandomState(seed)
    sizes = [int(np.sum(g)) for g in groups]
    N = int(np.sum(sizes))
    X = rng.randn(N, D) * 1.2
    y = (rng.rand(N) < 0.55).astype(int)
    if no.is_gauss(model):
        y = 0.2 + X.dot(rng.randn(D) * 0.4) + 0.9 * rng.randn(N)
    k_lim = np.concatenate(([0], np.cumsum(sizes)))
    g_cnt = np.array([len(g) for g in groups], dtype=np.int32)
    g_lim = np.concatenate(([0], np.cumsum([n for g in groups for n in g])))
    d = no.dims(model, D)[0]
    Oms, mus = [], []
    for k in range(len(groups)):
        A = rng.randn(d, d + 3)
        Oms.append((A.dot(A.T) / (d + 3) + 0.5 * np.eye(d)) * tight)
        mus.append(0.4 * rng.randn(d))
    return X, y, k_lim, g_cnt, g_lim, np.array(Oms), np.array(mus), d


def _group_engine(model, X, y, k_lim, g_cnt, g_lim, Oms, mus):
    eng = HipEngine(model, X, y, k_lim, g_cnt=g_cnt, g_lim=g_lim)
    d = eng.d
    for k in range(eng.K):
        assert eng.cavity_site(k, Oms[k] + np.eye(d), Oms[k].dot(mus[k]), np.eye(d), np.zeros(d))
    Om_dev = np.stack([eng.get_cavity(k)[0] for k in range(eng.K)])
    mu_dev = np.stack([eng.get_cavity(k)[1] for k in range(eng.K)])
    return eng, Om_dev, mu_dev


@pytest.mark.parametrize('model', ['m1b', 'm2b', 'm3b', 'm4b', 'm5b'])
@pytest.mark.parametrize('D,groups', [(3, [[5, 1, 9], [4, 4]]), (16, [[20, 20], [13, 30, 7], [40]]),
                                      (40, [[30, 25], [17]]), (70, [[33, 16, 16]])])
def test_multigroup_gradient_matches_oracle(model, D, groups):
    X, y, k_lim, g_cnt, g_lim, Oms, mus, d = _group_problem(model, D, groups, 11 + D)
    eng, Om_dev, mu_dev = _group_engine(model, X, y, k_lim, g_cnt, g_lim, Oms, mus)
    rng = np.random.RandomState(3)
    off = np.concatenate(([0], np.cumsum(g_cnt)))
    for k in range(len(groups)):
        Pk = no.dims(model, D, g_cnt[k])[1]
        assert eng.P >= Pk and eng.site_P[k] == Pk
        theta = np.zeros(eng.P)
        theta[:Pk] = rng.randn(Pk) * 0.3
        lo, hi = k_lim[k], k_lim[k + 1]
        gl = g_lim[off[k]:off[k + 1] + 1] - lo
        lp_o, g_o = no.logdensity_grad(model, X[lo:hi], y[lo:hi], mu_dev[k], Om_dev[k], theta[:Pk], gl=gl)
        # layout 2: one workgroup per chain, the gradient waves share the groups (D <= 32, P <= 128; larger
        # sites fall through to the lock-step layouts); 3 streaming; 4 lock step with resident rows (D <= 32)
        for layout in (2, 3, 4):
            lp, g = eng.logdensity_grad(k, theta, layout=layout)
            assert abs(lp - lp_o) <= 1e-10 * max(1.0, abs(lp_o)), (k, layout, lp, lp_o)
            np.testing.assert_allclose(g[:Pk], g_o, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(g_o).max()))
            assert np.all(g[Pk:] == 0.0)
            if layout == 2 and D <= 32 and eng.P <= 128:
                assert eng.last_layout() == 2


@pytest.mark.parametrize('layout', [2, 3, 4])
@pytest.mark.parametrize('model,D,groups,chains', [('m4b', 4, [[20, 14, 9], [25, 25]], 4), ('m1b', 16, [[30, 30], [18, 18, 18]], 3),
                                                   ('m3b', 8, [[40], [16, 24]], 4), ('m4b', 32, [[17, 40, 5]], 4),
                                                   ('m4b', 6, [[9, 70, 3, 11, 20], [150]], 4), ('m2b', 16, [[64, 65], [10, 12]], 2)])
def test_multigroup_site_updates_match_oracle(model, D, groups, chains, layout):
    """Whole short site updates of multi-group sites against the C oracle, chain by chain (same
    random stream; chains are compared until rounding differences make them part)."""
    X, y, k_lim, g_cnt, g_lim, Oms, mus, d = _group_problem(model, D, groups, 70 + D, tight=300.0)
    eng, Om_dev, mu_dev = _group_engine(model, X, y, k_lim, g_cnt, g_lim, Oms, mus)
    if layout == 2 and eng.P > 128:
        pytest.skip('one workgroup per chain holds at most 128 coordinates in registers')
    K = len(groups)
    seeds = np.arange(K, dtype=np.int64) + 31
    it = 44
    stats, ms = eng.sample_batch(seeds, HipEngine.sampler_opts(chains=chains, iter=it, init='random', layout=layout))
    assert eng.last_layout() == layout
    draws_o, _, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=chains, iter=it,
                                     g_cnt=g_cnt, g_lim=g_lim)
    cs = eng.get_chain_stats(chains)
    assert np.all(cs[:, :, 7] == 0) and np.all(np.isfinite(stats))
    nk = it // 2
    P = eng.P
    n_full = 0
    for k in range(K):
        dev = eng.get_draws(k, True).reshape(chains, nk, P)
        assert np.all(dev[:, :, eng.site_P[k]:] == 0.0)
        err = np.abs(dev - draws_o[k]).max(axis=2) / max(1.0, np.abs(draws_o[k]).max())
        for c in range(chains):
            assert np.all(err[c, :3] < 1e-3), (k, c, err[c, :3])
            if np.all(err[c] < 1e-4):
                n_full += 1
                assert cs[k, c, 2] == st_o[k, c, 2] and cs[k, c, 3] == st_o[k, c, 3]
    record_slack('multi-group run vs oracle: chains equal to the end', n_full, '>= %d' % ((3 * K * chains) // 4), 3 * K * chains)
    assert n_full >= (3 * K * chains) // 4, n_full
    # one group per site through the groups entry point = the `_sg` model, bit for bit
    ones = np.ones(K, dtype=np.int32)
    e1 = HipEngine(model, X, y, k_lim, g_cnt=ones, g_lim=k_lim)
    e2 = HipEngine(model + '_sg', X, y, k_lim)
    for e in (e1, e2):
        for k in range(K):
            assert e.cavity_site(k, Oms[k] + np.eye(d), Oms[k].dot(mus[k]), np.eye(d), np.zeros(d))
        e.sample_batch(seeds, HipEngine.sampler_opts(chains=chains, iter=20, init='random', layout=layout))
    for k in range(K):
        if layout == 2:
            # grouped kernel: a group's rows on ONE wave; `_sg` kernel: rows over four waves -> other summation order
            np.testing.assert_allclose(e1.get_draws(k, True)[:3], e2.get_draws(k, True)[:3], rtol=1e-6, atol=1e-8)
        else:
            np.testing.assert_array_equal(e1.get_draws(k, True), e2.get_draws(k, True))


@pytest.mark.parametrize('model,D,groups', [('m1a', 3, [[5, 1, 9], [4, 4]]), ('m2a', 16, [[20, 20], [13, 30, 7], [40]]),
                                            ('m3a', 8, [[9, 70, 3, 11, 20], [150]]), ('m4a', 16, [[20, 20], [64, 65]]),
                                            ('m5a', 6, [[10, 10, 10, 10], [33]])])
def test_gaussian_family_multigroup_matches_oracle(model, D, groups):
    """Gaussian likelihood with several groups per site (experiment/models/m1a.stan etc.): served by the
    one-workgroup-per-chain layout; gradients to 1e-9 and a short site update chain by chain against the oracle."""
    X, y, k_lim, g_cnt, g_lim, Oms, mus, d = _group_problem(model, D, groups, 5 + D, tight=300.0)
    eng, Om_dev, mu_dev = _group_engine(model, X, y, k_lim, g_cnt, g_lim, Oms, mus)
    rng = np.random.RandomState(4)
    off = np.concatenate(([0], np.cumsum(g_cnt)))
    K = len(groups)
    for k in range(K):
        Pk = no.dims(model, D, g_cnt[k])[1]
        assert eng.site_P[k] == Pk
        theta = np.zeros(eng.P); theta[:Pk] = rng.randn(Pk) * 0.3
        lo, hi = k_lim[k], k_lim[k + 1]
        gl = g_lim[off[k]:off[k + 1] + 1] - lo
        lp_o, g_o = no.logdensity_grad(model, X[lo:hi], y[lo:hi], mu_dev[k], Om_dev[k], theta[:Pk], gl=gl)
        lp, g = eng.logdensity_grad(k, theta)
        assert eng.last_layout() == 2
        assert abs(lp - lp_o) <= 1e-10 * max(1.0, abs(lp_o))
        np.testing.assert_allclose(g[:Pk], g_o, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(g_o).max()))
        assert np.all(g[Pk:] == 0.0)
    if model == 'm5a':
        return                                  # Laplace kinks: whole runs part ways with the oracle early (chaos)
    seeds = np.arange(K, dtype=np.int64) + 7
    it = 44
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=2, iter=it, init='random'))
    assert eng.last_layout() == 2
    draws_o, _, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=2, iter=it, g_cnt=g_cnt, g_lim=g_lim)
    cs = eng.get_chain_stats(2)
    n_full = 0
    for k in range(K):
        dev = eng.get_draws(k, True).reshape(2, it // 2, eng.P)
        err = np.abs(dev - draws_o[k]).max(axis=2) / max(1.0, np.abs(draws_o[k]).max())
        for c in range(2):
            assert np.all(err[c, :3] < 1e-3), (k, c, err[c, :3])
            if np.all(err[c] < 1e-4):
                n_full += 1
                assert cs[k, c, 2] == st_o[k, c, 2]
    assert n_full >= K
    # the lock-step resident layout (4) is not built for the Gaussian family: the request is served by streaming
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=2, iter=it, init='random', layout=4))
    assert eng.last_layout() == 3


def test_multigroup_ep_posterior_matches_cpu_path_within_monte_carlo_error():
    """K < J end to end (the reference's default experiment shape, scaled down: J = 8 groups on
    K = 4 sites): the EP posterior of the device path against the CPU (oracle) path, tolerance
    relative to the CPU path's own seed-to-seed spread as in the single-group test above."""
    from oracle.engine_oracle import OracleEngine
    from epstan_amd.util import distribute_groups
    mod = models.m4b(8, 3, 40)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    Nk, Nj_k, j_ind_k = distribute_groups(8, 4, data.Nj)

    def run(seed, **kw):
        M = Master('m4b', data.X, data.y, site_sizes=Nk, A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k + 1},
                   prior={'Q': Q0, 'r': r0}, chains=4, iter=400, df0=0.5, **kw)
        info, (m_s, S_s) = M.run(6, verbose=False, seed=seed)
        assert info == 0
        return m_s[-1], S_s[-1], M

    cpu = lambda m, X, y, kl, **g: OracleEngine(m, X, y, kl, **g)
    m_g, S_g, Mg = run(1)
    assert Mg.engine.last_layout() == 2 and Mg.engine.P == 8 + 2 * 4      # few sites: one workgroup per chain
    m_c1, S_c1, _ = run(1, _engine_factory=cpu)
    m_c2, S_c2, _ = run(2, _engine_factory=cpu)
    sd = np.sqrt(np.diag(S_c1))
    tol_m = max(0.15, 3 * (np.abs(m_c1 - m_c2) / sd).max())
    tol_v = max(0.25, 3 * np.abs(np.diag(S_c1) / np.diag(S_c2) - 1).max())
    assert np.all(np.abs(m_g - m_c1) / sd < tol_m), (np.abs(m_g - m_c1) / sd, tol_m)
    assert np.all(np.abs(np.diag(S_g) / np.diag(S_c1) - 1) < tol_v), tol_v
    assert np.all(np.diag(S_g) < 0.8 * np.diag(np.linalg.inv(Q0)))


def test_fit_main_with_fewer_sites_than_groups_on_gpu(tmp_path, monkeypatch):
    from epstan_amd import fit
    monkeypatch.setattr(fit, 'RES_PATH', str(tmp_path))
    conf = fit.configurations(J=12, D=4, K=5, npg=15, iter=3, siter=60, run_ep=True, id='kj')
    res = fit.main('m4b', conf, verbose=False)
    assert res['m_s_ep'].shape == (4, 10) and np.all(np.isfinite(res['S_s_ep']))
    assert np.all(np.linalg.eigvalsh(res['S_s_ep'][-1]) > 0)
    assert os.path.exists(os.path.join(str(tmp_path), 'res_d_m4b_kj.npz'))


def test_mix_phi_on_gpu_equals_pooled_sample_moments():
    """Master.mix_phi (method.py:1250-1296) from the device's per-site tilted moments against the
    pooled mean / covariance of the actual draws."""
    mod = models.m1b(6, 3, 30)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=80, df0=0.5)
    assert M.run(2, verbose=False, seed=4)[0] == 0
    S, m = M.mix_phi()
    draws = [M.engine.get_draws(k) for k in range(M.K)]
    means = np.stack([dk.mean(0) for dk in draws])
    mref = means.mean(0)
    allc = np.concatenate([dk - mk for dk, mk in zip(draws, means)])
    n = draws[0].shape[0]
    Sref = (allc.T.dot(allc) + n * sum(np.outer(mk - mref, mk - mref) for mk in means)) / (n * M.K - 1)
    np.testing.assert_allclose(m, mref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(S, Sref, rtol=1e-8, atol=1e-12)


@pytest.mark.gpu
def test_layout_policy_for_the_baseline_shapes():
    """Which thread layout the library picks by itself (epx_sampler_opts.layout = 0) for the shapes
    of BASELINE.json and of the reference's default experiment; tiny iteration counts, the point
    is the decision (HISTORY.md 3.1-3.1d)."""
    rng = np.random.RandomState(0)

    def picked(model, K, D, n, g=None, chains=4):
        X = rng.randn(K * n, D)
        y = (rng.rand(K * n) < 0.5).astype(int)
        kw = {}
        if g is not None:                                       # g groups of n // g rows in every site
            kw = dict(g_cnt=np.full(K, g, dtype=np.int32), g_lim=np.arange(K * g + 1) * (n // g))
        eng = HipEngine(model, X, y, np.arange(K + 1) * n, **kw)
        d = eng.d
        eng.set_prior(np.eye(d), np.zeros(d))
        eng.set_global(np.eye(d) * 2.0, np.zeros(d))
        assert np.all(eng.cavity_batch(QI))
        eng.sample_batch(np.arange(K) + 1, HipEngine.sampler_opts(chains=chains, iter=4, init='random', max_depth=3))
        lay = eng.last_layout()
        eng.close()
        return lay

    assert picked('m4b_sg', 64, 16, 200) == 6          # C2: every chain gets a CU -> one workgroup per chain, roles on waves
    assert picked('m3b_sg', 64, 16, 200) == 2          # ... models without per-coefficient scales: gradient waves + bookkeeping wave
    assert picked('m4b_sg', 256, 16, 200) == 2         # two such workgroups share a CU: still ahead (measured)
    assert picked('m4b_sg', 400, 16, 200) == 7         # many small sites -> one workgroup per site (state waves + row team)
    assert picked('m4b_sg', 512, 32, 500) == 7         # C3 / C4 per GPU
    assert picked('m1b_sg', 400, 4, 50) == 1           # D <= 8: the one-wave-per-chain kernel
    assert picked('m4b_sg', 100, 32, 500) == 2
    assert picked('m4b_sg', 4, 128, 2000) == 3         # C5 site size: rows beyond the LDS -> streaming
    assert picked('m1b_sg', 6, 64, 100) == 3           # D > 32
    assert picked('m4b', 32, 16, 40, g=2) == 2         # the reference's default experiment: 2 groups per site, few sites
    assert picked('m4b', 300, 16, 40, g=2) == 2        # ... and many of them (ahead of lock step at every count measured)
    assert picked('m4b', 12, 16, 160, g=8) == 4        # P = 170 > 128 coordinates: chains in lock step, rows resident
    assert picked('m4b', 8, 64, 60, g=3) == 3          # multi-group, D > 32 -> streaming


@pytest.mark.gpu
@pytest.mark.parametrize('name,K', [('m4b', 8), ('m4b', 16), ('m4a', 8), ('m1b', 16), ('m1a', 16)])
def test_ep_converges_to_the_full_posterior_of_the_joint_model(name, K):
    """A target that involves neither the reference nor the EP code: the posterior of the whole hierarchical
    model, sampled by the oracle's NUTS as ONE site that holds all J groups with the prior as its cavity.
    Device EP (K = J: one group per site; K < J: two groups per site; logistic and Gaussian likelihood), with
    enough draws per site update for the Monte-Carlo error to be small, has to land on it: means within 0.35
    posterior sd in every coordinate (measured over three seeds per case: 0.08-0.21), marginal sd within a
    factor 0.55-1.25 (measured 0.65-1.07; the low end is the skewed log-scale coordinate of the logistic models).  scripts/ep_vs_full_posterior.py is the full-size version."""
    from epstan_amd.util import distribute_groups
    J, D, npg = 16, 4, 30
    mod = models.MODELS[name](J, D, npg)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    S0, m0, Q0, r0 = mod.get_prior()
    d = mod.dphi
    draws, _, st = no.nuts_sites(name, data.X, data.y, np.array([0, data.X.shape[0]]), m0[None], Q0[None], [11],
                                 chains=4, iter=3000, g_cnt=np.array([J], dtype=np.int32),
                                 g_lim=data.j_lim.astype(np.int64), nthreads=4)
    x = draws[0].reshape(-1, draws.shape[-1])[:, :d]
    m_full, sd_full = x.mean(0), x.std(0)
    assert max(no.split_rhat(draws[0, :, :, e]) for e in range(d)) < 1.1
    if K < J:
        Nk, Nj_k, j_ind_k = distribute_groups(J, K, data.Nj)
        M = Master(name, data.X, data.y, site_sizes=Nk, A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k + 1},
                   prior={'Q': Q0, 'r': r0}, chains=4, iter=1500, df0=0.4)
    else:
        M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=1500, df0=0.3)
    info, (m_s, S_s) = M.run(14 if K < J else 22, verbose=False, seed=3)
    assert info == 0 and M.engine.last_layout() == 2
    z = np.abs(m_s[-1] - m_full) / sd_full
    ratio = np.sqrt(np.diag(S_s[-1])) / sd_full
    z0 = np.abs(m0 - m_full) / sd_full
    assert z.max() < 0.35, z
    assert ratio.min() > 0.55 and ratio.max() < 1.25, ratio
    assert z0.max() > 2.0                       # the prior is far from it: EP did the work
