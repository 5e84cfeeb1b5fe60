"""The NumPy oracle (oracle/ep_oracle.py) against the golden vectors captured
from the imported reference (tests/golden/make_golden.py). CPU only."""

import os

import numpy as np
import pytest

from oracle import ep_oracle as eo
import injectors

RTOL, ATOL = 1e-9, 1e-12     # SURVEY.md §8c stated tolerance for deterministic stages


@pytest.fixture(scope='module')
def alg(golden_dir):
    return np.load(os.path.join(golden_dir, 'algebra.npz'))


@pytest.fixture(scope='module')
def runs(golden_dir):
    return np.load(os.path.join(golden_dir, 'master_run.npz'))


@pytest.mark.parametrize('d', [5, 17, 33, 66])
def test_invert_normal_params(alg, d):
    Q, r = eo.invert_normal_params(alg['g1_S_%d' % d], alg['g1_m_%d' % d])
    np.testing.assert_allclose(Q, alg['g1_Q_%d' % d], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(r, alg['g1_r_%d' % d], rtol=RTOL, atol=ATOL)
    Q, r = eo.invert_normal_params(alg['g1_U_%d' % d], alg['g1_m_%d' % d], cho_form=True)
    np.testing.assert_allclose(Q, alg['g1_Qc_%d' % d], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(r, alg['g1_rc_%d' % d], rtol=RTOL, atol=ATOL)


def test_invert_not_posdef(alg):
    assert int(alg['g1_bad_raises']) == 1
    with pytest.raises(eo.NotPosDef):
        eo.invert_normal_params(alg['g1_bad'], np.zeros(6))


@pytest.mark.parametrize('d', [5, 17, 33])
@pytest.mark.parametrize('n', [100, 400])
def test_olse(alg, d, n):
    key = '%d_%d' % (d, n)
    np.testing.assert_allclose(eo.olse(alg['g2_S_' + key], n),
                               alg['g2_none_' + key], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(eo.olse(alg['g2_S_' + key], n, P=alg['g2_P_' + key]),
                               alg['g2_prior_' + key], rtol=RTOL, atol=ATOL)


def test_cavity(alg):
    Mat, vec, ok = eo.cavity(alg['g3_Q'], alg['g3_r'], alg['g3_Qi'], alg['g3_ri'])
    assert ok == bool(alg['g3_flag']) and ok
    np.testing.assert_allclose(Mat, alg['g3_Mat'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(vec, alg['g3_vec'], rtol=RTOL, atol=ATOL)
    _, _, ok = eo.cavity(alg['g3_Q'], alg['g3_r'], alg['g3_Qi_bad'], alg['g3_ri'])
    assert ok == bool(alg['g3_flag_bad']) and not ok


def g4_samples(seed, S, d):
    rng = np.random.RandomState(seed)
    mix = np.eye(d) + 0.3 * rng.randn(d, d) / np.sqrt(d)
    shift = rng.randn(d)
    return np.asfortranarray(rng.randn(S, d).dot(mix) + shift)


@pytest.mark.parametrize('d', [5, 10, 17, 34])
@pytest.mark.parametrize('est', ['sample', 'olse'])
def test_tilted_moments(alg, d, est):
    samp = g4_samples(int(alg['g4_seed_%d' % d]), 400, d)
    dQi, dri, mt, scatter, ok = eo.tilted_moments(
        samp, alg['g4_Q_%d' % d], alg['g4_r_%d' % d], est)
    key = '%s_%d' % (est, d)
    assert ok == bool(alg['g4_flag_' + key])
    np.testing.assert_allclose(dQi, alg['g4_dQi_' + key], rtol=RTOL, atol=1e-10)
    np.testing.assert_allclose(dri, alg['g4_dri_' + key], rtol=RTOL, atol=1e-10)
    np.testing.assert_allclose(mt, alg['g4_vec_' + key], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(scatter, alg['g4_scatter_%d' % d], rtol=RTOL, atol=1e-10)


def test_seed_derivation(alg):
    seeds = eo.run_seeds(1, 3, 5)
    np.testing.assert_array_equal(seeds, alg['g8_seeds'])
    stan = np.array([[eo.stan_seed(s) for s in row] for row in seeds])
    np.testing.assert_array_equal(stan, alg['g8_stan_seeds'])
    assert eo.stan_seed(7) == int(alg['g4_stanseed'])


def _sampler(scenario, factor=60.0):
    inj = injectors.GaussianTilted(scenario, factor=factor)

    def f(data, stan_params):
        return inj(data, stan_params), [{}] * stan_params['chains'], 0.25, 0.125, 1.0625
    return f


def _master(runs, scenario, df0, nsites=4, prec_estim='sample', factor=60.0):
    Nj = runs['g6_Nj'][:nsites]
    nrow = int(Nj.sum())
    return eo.OracleMaster(
        runs['g6_X'][:nrow], runs['g6_y'][:nrow], Nj, _sampler(scenario, factor),
        prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']},
        A_k={'site_id': np.arange(nsites)}, chains=4, iter=200, df0=df0,
        prec_estim=prec_estim)


@pytest.mark.parametrize('tag,scenario,niter,df0,nsites,est,factor', [
    ('smooth', 'smooth', 12, 0.5, 4, 'sample', 60.0),
    ('smooth_olse', 'smooth', 6, 0.5, 4, 'olse', 60.0),
    ('decay', 'wide_first', 4, 1.0, 3, 'sample', 60.0),
    ('allfail', 'degenerate', 3, 0.5, 4, 'sample', 60.0),
    ('badprior', 'wide_all', 3, 1.0, 4, 'sample', 400.0),
])
def test_master_run_trajectories(runs, tag, scenario, niter, df0, nsites, est, factor):
    M = _master(runs, scenario, df0, nsites, est, factor)
    info, (m_s, S_s), (st, ms, rh) = M.run(niter, seed=1)
    assert info == int(runs['g6_%s_info' % tag])
    assert M.iter == int(runs['g6_%s_iter' % tag])
    np.testing.assert_allclose(m_s, runs['g6_%s_m' % tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(S_s, runs['g6_%s_S' % tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(M.Qi, runs['g6_%s_Qi' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(M.ri, runs['g6_%s_ri' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(st, runs['g6_%s_stimes' % tag])
    np.testing.assert_allclose(ms, runs['g6_%s_msteps' % tag])
    np.testing.assert_array_equal([w.phase for w in M.workers], runs['g6_%s_phase' % tag])
    if tag == 'decay':
        # the scenario really exercises the damping back-off (method.py:1160-1176)
        assert M.df_log[0][1] < 1.0


def test_master_run_resume(runs):
    M = _master(runs, 'smooth', 0.5)
    M.run(2, seed=5)
    info, (m_s, S_s), _ = M.run(2, seed=6)
    assert M.iter == int(runs['g6_resume_iter'])
    np.testing.assert_allclose(m_s, runs['g6_resume_m'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(S_s, runs['g6_resume_S'], rtol=1e-8, atol=1e-10)


# ---------------------------------------------------------------- damping sweep (find_damp.py:146-173)
def test_damp_sweep_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'damp_sweep.npz'))
    out = eo.damp_sweep(z['g9_Q0'], z['g9_r0'], z['g9_Qi'], z['g9_ri'], z['g9_dQi'], z['g9_dri'], z['g9_damps'],
                        z['g9_m_target'], z['g9_S_target'], z['g9_samp_target'])
    np.testing.assert_array_equal(out[:, 0] > 0, z['g9_global_pd'])
    np.testing.assert_array_equal(out[:, 1] > 0, z['g9_cav_pd'])
    assert z['g9_cav_pd'].sum() >= 31 and (~z['g9_cav_pd']).sum() >= 3 and (~z['g9_global_pd']).sum() >= 1
    np.testing.assert_allclose(out[:, 2], z['g9_mses'], rtol=RTOL, atol=ATOL, equal_nan=True)
    np.testing.assert_allclose(out[:, 3], z['g9_kls'], rtol=RTOL, atol=1e-10, equal_nan=True)
    np.testing.assert_allclose(out[:, 4], z['g9_lls'], rtol=RTOL, atol=1e-8, equal_nan=True)
