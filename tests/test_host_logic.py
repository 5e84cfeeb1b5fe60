"""Host-side logic on CPU: simulators, seed derivation, the C-ABI library's
symbols, and `Master`'s driver logic (partition, damping state machine,
return conventions) replayed against the reference's golden trajectories with
the oracle standing in for the device engine.  No GPU needed."""

import ctypes
import os
import re
import sys

import numpy as np
import pytest

import epstan_amd
from epstan_amd import _lib, dist, models, seeds
from epstan_amd.method import Master, Worker
from oracle.engine_oracle import OracleEngine
import injectors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def factory(model, X, y, k_lim, **groups):
    return OracleEngine(model, X, y, k_lim, **groups)


@pytest.fixture(scope='module')
def runs(golden_dir):
    return np.load(os.path.join(golden_dir, 'master_run.npz'))


@pytest.fixture(scope='module')
def alg(golden_dir):
    return np.load(os.path.join(golden_dir, 'algebra.npz'))


@pytest.fixture(scope='module')
def sim(golden_dir):
    return np.load(os.path.join(golden_dir, 'simulators.npz'))


# ---------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    """libepx.so loads and exports each function include/epx.h declares."""
    hdr = open(os.path.join(ROOT, 'include', 'epx.h')).read()
    declared = set(re.findall(r'\b(epx_[a-z_0-9]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    _lib.load()


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product refuses to run (no CPU fallback)."""
    if _lib.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_lib.EpxError):
        epstan_amd.util.invert_normal_params(np.eye(3, order='F'), np.zeros(3))
    X = np.zeros((4, 2)); y = np.zeros(4, dtype=int)
    with pytest.raises(_lib.EpxError):
        Master('m1b_sg', X, y, site_sizes=np.array([2, 2]), dphi=3)


# ---------------------------------------------------------------- seeds
def test_seed_derivation_matches_reference(alg):
    s = seeds.run_seeds(1, 3, 5)
    np.testing.assert_array_equal(s, alg['g8_seeds'])
    np.testing.assert_array_equal(seeds.stan_seeds(s), alg['g8_stan_seeds'])
    assert seeds.stan_seed(7) == int(alg['g4_stanseed'])
    big = seeds.run_seeds(123, 4, 700)
    ref = np.array([[np.random.RandomState(v).randint(0, 2**31 - 1) for v in row] for row in big])
    np.testing.assert_array_equal(seeds.stan_seeds(big), ref)


# ---------------------------------------------------------------- simulators
@pytest.mark.parametrize('name,tag,J,D,n', [('m1b', 'm1b_c1', 4, 4, 50), ('m4b', 'm4b_c1', 4, 4, 50)])
def test_simulator_small_exact(sim, name, tag, J, D, n):
    mod = models.MODELS[name](J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    np.testing.assert_allclose(data.X, sim['g7_%s_X' % tag], rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(data.y, sim['g7_%s_y' % tag])
    np.testing.assert_allclose(data.X_param['Sigma_x'], sim['g7_%s_Sigma_x' % tag], rtol=1e-13)
    np.testing.assert_allclose(data.phi_true, sim['g7_%s_phi_true' % tag], rtol=1e-14)
    _, _, Q0, r0 = mod.get_prior()
    np.testing.assert_allclose(np.diag(Q0), sim['g7_%s_Q0diag' % tag])
    np.testing.assert_allclose(r0, sim['g7_%s_r0' % tag])


@pytest.mark.parametrize('name,tag,J,D,n', [('m1b', 'm1b_c2', 64, 16, 200), ('m4b', 'm4b_c2', 64, 16, 200),
                                           ('m4b', 'm4b_c3', 512, 32, 500)])
def test_simulator_bench_sizes(sim, name, tag, J, D, n):
    data = models.MODELS[name](J, D, n).simulate_data(Sigma_x='rand', rng=100)
    np.testing.assert_allclose(data.X[:8], sim['g7_%s_X_head' % tag], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(data.X[-8:], sim['g7_%s_X_tail' % tag], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(data.X.sum(0), sim['g7_%s_X_colsum' % tag], rtol=1e-10)
    np.testing.assert_array_equal(np.add.reduceat(data.y, data.j_lim[:-1]), sim['g7_%s_y_sitesum' % tag])
    np.testing.assert_array_equal(data.y[:64], sim['g7_%s_y_head' % tag])
    np.testing.assert_allclose(data.X_param['sigma_x'], sim['g7_%s_sigma_x' % tag], rtol=1e-12)


def test_default_df0():
    f = models.default_df0(64)
    assert abs(f(1) - 0.5) < 1e-15
    assert abs(f(64) - (0.5 - 1/64) * 0.1 - 1/64) < 1e-12


# ---------------------------------------------------------------- Master host logic
def _master(runs, scenario, df0, nsites=4, prec_estim='sample', factor=60.0, **kw):
    Nj = runs['g6_Nj'][:nsites]
    nrow = int(Nj.sum())
    M = Master('some/dir/m1b_sg', runs['g6_X'][:nrow], runs['g6_y'][:nrow], site_sizes=Nj,
               prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']},
               A_k={'site_id': np.arange(nsites)}, chains=4, iter=200, df0=df0,
               prec_estim=prec_estim, _engine_factory=factory, **kw)
    M._sample_injector = injectors.GaussianTilted(scenario, factor=factor)
    return M


@pytest.mark.parametrize('tag,scenario,niter,df0,nsites,est,factor', [
    ('smooth', 'smooth', 12, 0.5, 4, 'sample', 60.0),
    ('smooth_olse', 'smooth', 6, 0.5, 4, 'olse', 60.0),
    ('decay', 'wide_first', 4, 1.0, 3, 'sample', 60.0),
    ('allfail', 'degenerate', 3, 0.5, 4, 'sample', 60.0),
    ('badprior', 'wide_all', 3, 1.0, 4, 'sample', 400.0),
])
def test_master_run_matches_reference_trajectory(runs, tag, scenario, niter, df0, nsites, est, factor):
    M = _master(runs, scenario, df0, nsites, est, factor)
    res = M.run(niter, verbose=False, return_analytics=True, seed=1)
    info, (m_s, S_s), (st, ms, rh, ot) = res
    assert info == int(runs['g6_%s_info' % tag])
    # list on early exits, tuple on the normal path (method.py:1040 vs :1247)
    assert isinstance(res, tuple) == (info == 0)
    assert M.iter == int(runs['g6_%s_iter' % tag])
    np.testing.assert_allclose(m_s, runs['g6_%s_m' % tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(S_s, runs['g6_%s_S' % tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(M.Qi, runs['g6_%s_Qi' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(M.ri, runs['g6_%s_ri' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(M.Q, runs['g6_%s_Q' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(st, runs['g6_%s_stimes' % tag])
    np.testing.assert_allclose(ms, runs['g6_%s_msteps' % tag])
    np.testing.assert_allclose(rh, runs['g6_%s_mrhats' % tag])
    np.testing.assert_array_equal([w.phase for w in M.workers], runs['g6_%s_phase' % tag])
    assert M.Qi.flags['F_CONTIGUOUS'] and M.Qi.shape == (5, 5, nsites)


def test_master_run_resume_and_return_shapes(runs):
    M = _master(runs, 'smooth', 0.5)
    assert M.run(0, verbose=False) == [0, (None, None)]
    assert M.run(0, verbose=False, calc_moments=False) == 0
    assert M.run(0, verbose=False, return_analytics=True) == [0, (None, None), (None, None, None)]
    M.run(2, verbose=False, seed=5)
    info, (m_s, S_s) = M.run(2, verbose=False, seed=6)
    assert info == 0 and M.iter == int(runs['g6_resume_iter'])
    np.testing.assert_allclose(m_s, runs['g6_resume_m'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(S_s, runs['g6_resume_S'], rtol=1e-8, atol=1e-10)
    assert M.run(1, verbose=False, calc_moments=False, seed=1) == 0


def test_master_init_partition_and_errors(alg):
    X, y, sizes = alg['g5_X'], alg['g5_y'], alg['g5_sizes']
    M = Master('m1b_sg', X, y, site_sizes=sizes, dphi=4, init_site=3.0, _engine_factory=factory)
    np.testing.assert_array_equal(M.k_lim, alg['g5_k_lim'])
    np.testing.assert_array_equal(M.k_ind, alg['g5_k_ind'])
    np.testing.assert_allclose(M.Q, alg['g5_Q'], rtol=1e-12)
    np.testing.assert_allclose(M.Qi, alg['g5_Qi'], rtol=1e-12)
    ind_ord = np.repeat(np.arange(3), sizes)
    M2 = Master('m1b_sg', X, y, site_ind_ord=ind_ord, dphi=4, _engine_factory=factory)
    np.testing.assert_array_equal(M2.k_lim, alg['g5_ord_k_lim'])
    np.testing.assert_array_equal(np.asarray(M2.Nk), alg['g5_ord_Nk'])
    np.testing.assert_allclose(M2.workers[1].data['X'], alg['g5_w1_X'])
    np.testing.assert_allclose(M2.workers[1].Mat, alg['g5_w1_Mat'], rtol=1e-12)
    np.testing.assert_allclose(M2.workers[1].vec, alg['g5_w1_vec'], atol=1e-14)
    perm = np.random.RandomState(0).permutation(30)
    M3 = Master('m1b_sg', X[perm], y[perm], site_ind=ind_ord[perm], dphi=4, _engine_factory=factory)
    np.testing.assert_array_equal(M3.k_lim, alg['g5_k_lim'])
    # error conventions of method.py:656-730, 774-797, 808-810, 874, 882
    with pytest.raises(TypeError):
        Master('m1b_sg', X, y, site_sizes=sizes, dphi=4, bogus=1, _engine_factory=factory)
    with pytest.raises(NotImplementedError):
        Master('m1b_sg', X, y, dphi=4, _engine_factory=factory)
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=np.array([7, 11, 11]), dphi=4, _engine_factory=factory)
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=np.array([7, 0, 23]), dphi=4, _engine_factory=factory)
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=np.array([30]), dphi=4, _engine_factory=factory)
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=sizes, _engine_factory=factory)          # no prior, no dphi
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=sizes, dphi=4, df0=1.5, _engine_factory=factory)
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y[:-1], site_sizes=sizes, dphi=4, _engine_factory=factory)
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=sizes, prior={'Q': -np.eye(4), 'r': np.zeros(4)},
               _engine_factory=factory)                                             # non-pd initial
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=sizes, dphi=4, prec_estim='glassocv', _engine_factory=factory)
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=sizes, dphi=4, A={'X': 1}, _engine_factory=factory)
    with pytest.raises(TypeError):
        Master(object(), X, y, site_sizes=sizes, dphi=4, _engine_factory=factory)
    assert Master.DEFAULT_KWARGS['df_decay'] == 0.8 and Master.DEFAULT_KWARGS['df_treshold'] == 1e-6


def test_worker_direct_calls_find_damp_pattern(runs):
    """find_damp.py:136-160 drives workers directly with NumPy views."""
    M = _master(runs, 'smooth', 0.5)
    posdefs = [w.tilted(M.dQi[:, :, k], M.dri[:, k], seed=11 + k) for k, w in enumerate(M.workers)]
    assert all(posdefs) and all(w.phase == 2 for w in M.workers)
    df = 0.3
    np.add(M.Qi, np.multiply(df, M.dQi, out=M.Qi2), out=M.Qi2)
    np.add(M.ri, np.multiply(df, M.dri, out=M.ri2), out=M.ri2)
    np.add(M.Qi2.sum(2, out=M.Q), M.Q0, out=M.Q)
    np.add(M.ri2.sum(1, out=M.r), M.r0, out=M.r)
    for k, w in enumerate(M.workers):
        assert w.cavity(M.Q, M.r, M.Qi2[:, :, k], M.ri2[:, k])
        np.testing.assert_allclose(w.Mat, M.Q - M.Qi2[:, :, k], rtol=1e-12, atol=1e-12)
    with pytest.raises(RuntimeError):
        M.workers[0].phase = 0
        M.workers[0].tilted(M.dQi[:, :, 0], M.dri[:, 0])


def test_force_pd_branch(runs):
    """Damping collapses below df_treshold -> one force-pd pass (method.py:1177-1207:
    min-eig shift on Qi, df reset to df0) after which the update is accepted.
    The reference's own eigvalsh(eigvals=(0,0)) call no longer runs on this
    SciPy, so the oracle restatement is the comparison."""
    from oracle import ep_oracle as eo
    M = _master(runs, 'wide_first', 1.0, nsites=3, df_treshold=0.9)
    info = M.run(2, verbose=False, calc_moments=False, seed=1)
    inj = injectors.GaussianTilted('wide_first')
    Nj = runs['g6_Nj'][:3]
    nrow = int(Nj.sum())
    O = eo.OracleMaster(runs['g6_X'][:nrow], runs['g6_y'][:nrow], Nj,
                        lambda data, sp: (inj(data, sp), [{}] * 4, 0.25, 0.125, 1.0625),
                        prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']},
                        A_k={'site_id': np.arange(3)}, chains=4, iter=200, df0=1.0, df_treshold=0.9)
    assert O.run(2, seed=1)[0] == info == 0
    np.testing.assert_allclose(M.Qi, O.Qi, rtol=1e-9, atol=1e-10)    # includes the diagonal shift
    # without the threshold the same scenario only decays df: different site parameters
    M0 = _master(runs, 'wide_first', 1.0, nsites=3)
    M0.run(2, verbose=False, calc_moments=False, seed=1)
    assert np.abs(M0.Qi - M.Qi).max() > 1e-3


def test_site_range():
    assert [dist.site_range(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert dist.site_range(4096, 7, 8) == (3584, 4096)


# ---------------------------------------------------------------- fit.py plumbing
def test_fit_configurations_and_ep_branch_schema(tmp_path, monkeypatch):
    from epstan_amd import fit
    conf = fit.configurations()
    assert (conf.J, conf.D, conf.K, conf.npg, conf.siter, conf.chains) == (64, 16, 32, 20, 200, 4)   # fit.py:134-151
    assert (conf.seed_data, conf.seed_ep, conf.prec_estim) == (100, 1, 'sample')
    assert fit.EP_DEFAULT_ITERS_TO_RUN(64) == 256 and fit.EP_DEFAULT_ITERS_TO_RUN(3) == 20
    with pytest.raises(ValueError):
        fit.configurations(bogus=1)
    assert 'J = 64' in str(conf)
    monkeypatch.setattr(fit, 'RES_PATH', str(tmp_path))
    conf = fit.configurations(J=4, D=3, K=4, npg=15, iter=2, siter=40, run_ep=True, id='t')
    res = fit.main('m1b', conf, verbose=False, _engine_factory=factory)
    d = 4
    assert res['m_s_ep'].shape == (3, d) and res['S_s_ep'].shape == (3, d, d)      # initial approx prepended
    assert res['time_s_ep'][0] == 0.0 and np.all(np.diff(res['time_s_ep']) >= 0)
    assert np.isnan(res['mstepsize_s_ep'][0]) and np.isnan(res['mrhat_s_ep'][0]) and res['othertimes'].shape == (2,)
    saved = np.load(os.path.join(str(tmp_path), 'res_d_m1b_t.npz'), allow_pickle=True)
    assert set(saved.files) == {'conf', 'm_s_ep', 'S_s_ep', 'time_s_ep', 'mstepsize_s_ep', 'mrhat_s_ep', 'othertimes'}
    np.testing.assert_allclose(res['S_s_ep'][0], np.eye(d) * 1.5**2, rtol=1e-12)        # the prior (m1b.py:46-50)
    with pytest.raises(NotImplementedError):
        fit.main('m1b', fit.configurations(run_full=True), _engine_factory=factory)
    # conf.mix (fit.py:408-411, 440-441): the final approximation mixed from the last samples of all the sites
    conf = fit.configurations(J=4, D=3, K=4, npg=15, iter=2, siter=40, run_ep=True, id='mx', mix=True)
    res = fit.main('m1b', conf, verbose=False, _engine_factory=factory)
    saved = np.load(os.path.join(str(tmp_path), 'res_d_m1b_mx.npz'), allow_pickle=True)
    assert {'m_phi_ep', 'S_phi_ep'} <= set(saved.files)
    assert res['m_phi_ep'].shape == (d,) and res['S_phi_ep'].shape == (d, d)
    assert np.linalg.eigvalsh(res['S_phi_ep'])[0] > 0
    np.testing.assert_allclose(saved['S_phi_ep'], res['S_phi_ep'])
    M = fit.main('m4b', fit.configurations(J=4, D=2, K=4, npg=10), ret_master=True, _engine_factory=factory)
    assert isinstance(M, Master) and M.dphi == 6 and abs(M.df0(1) - 0.5) < 1e-15


# ---------------------------------------------------------------- find_damp
def test_master_damp_sweep_and_find_damp_driver(golden_dir, tmp_path, monkeypatch):
    from epstan_amd import fit, find_damp
    from oracle import ep_oracle as eo
    z = np.load(os.path.join(golden_dir, 'damp_sweep.npz'))
    runs = np.load(os.path.join(golden_dir, 'master_run.npz'))
    M = _master(runs, 'smooth', 0.5)
    M.run(2, verbose=False, seed=3)
    # put the golden state into the host mirrors: the sweep reads the device copies
    M.Qi[...] = z['g9_Qi']; M.ri[...] = z['g9_ri']
    M._upload_sites()
    M.engine.set_sites(2, z['g9_dQi'], z['g9_dri'])                 # EPX_DQI
    res = M.damp_sweep(z['g9_damps'], z['g9_m_target'], z['g9_S_target'], z['g9_samp_target'])
    np.testing.assert_array_equal(res['global_pd'], z['g9_global_pd'])
    np.testing.assert_array_equal(res['cav_pd'], z['g9_cav_pd'])
    np.testing.assert_allclose(res['mses'], z['g9_mses'], rtol=1e-9, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(res['kls'], z['g9_kls'], rtol=1e-9, atol=1e-10, equal_nan=True)
    np.testing.assert_allclose(res['lls'], z['g9_lls'], rtol=1e-9, atol=1e-8, equal_nan=True)
    # run(..., sweep=...) logs one sweep per iteration and the accepted damping factor
    M2 = _master(runs, 'smooth', 0.5)
    sw = dict(damps=find_damp.default_damps(), m_target=z['g9_m_target'], S_target=z['g9_S_target'])
    assert M2.run(3, verbose=False, seed=3, sweep=sw)[0] == 0
    assert len(M2.sweep_log) == 3 and len(M2.df_log) == 3 and M2.df_log[0] == 0.5
    assert M2.sweep_log[0]['kls'].shape == (31,) and np.all(np.isnan(M2.sweep_log[0]['lls']))
    # the trajectory is the same as without the sweep
    M3 = _master(runs, 'smooth', 0.5)
    M3.run(3, verbose=False, seed=3)
    np.testing.assert_array_equal(M2.Qi, M3.Qi)
    assert M2.df_log == M3.df_log
    # the driver: schema of find_damp_K<K>.npz (find_damp.py:246-256)
    monkeypatch.setattr(fit, 'RES_PATH', str(tmp_path))
    conf = fit.configurations(J=4, D=3, K=4, npg=15, siter=40, chains=2)
    tgt = dict(m_target=np.zeros(4), S_target=np.eye(4), samp_target=np.random.RandomState(0).randn(50, 4))
    out = find_damp.main('m1b', iters=2, target=tgt, conf=conf, seed=1, verbose=False, _engine_factory=factory)
    assert out['kls'].shape == (2, 31) and out['kls_selected'].shape == (3,) and out['damps'].shape == (31,)
    assert np.all(np.isfinite(out['kls_selected'])) and np.all(np.isfinite(out['lls_selected']))
    assert np.all(out['damps_selected'] > 0)
    saved = np.load(os.path.join(str(tmp_path), 'find_damp_K4.npz'))
    assert set(saved.files) == {'damps', 'mses', 'lls', 'kls', 'damps_selected', 'mses_selected', 'lls_selected',
                                'kls_selected'}


# ---------------------------------------------------------------- K < J: several groups per site
def test_distribute_groups_matches_reference(golden_dir):
    from epstan_amd.util import distribute_groups
    z = np.load(os.path.join(golden_dir, 'damp_sweep.npz'))
    for i in range(int(z['g10_n'])):
        J, K = z['g10_%d_JK' % i]
        Nk, Nj_k, j_ind_k = distribute_groups(int(J), int(K), z['g10_%d_Nj' % i])
        np.testing.assert_array_equal(Nk, z['g10_%d_Nk' % i])
        np.testing.assert_array_equal(Nj_k, z['g10_%d_Nj_k' % i])
        np.testing.assert_array_equal(j_ind_k, z['g10_%d_j_ind_k' % i])
    assert distribute_groups(3, 3, np.array([2, 2, 2]))[1] is None
    with pytest.raises(ValueError):
        distribute_groups(3, 1, np.array([2, 2, 2]))
    with pytest.raises(ValueError):
        distribute_groups(3, 2, np.array([2, 0, 2]))
    with pytest.raises(NotImplementedError):
        distribute_groups(3, 4, np.array([2, 2, 2]))


def test_multigroup_master_plumbing(tmp_path, monkeypatch):
    from epstan_amd import fit
    from epstan_amd.util import distribute_groups
    monkeypatch.setattr(fit, 'RES_PATH', str(tmp_path))
    conf = fit.configurations(J=7, D=2, K=3, npg=9, iter=2, siter=30, chains=2, run_ep=True, save_res=False)
    M = fit.main('m4b', conf, ret_master=True, _engine_factory=factory)
    Nk, Nj_k, j_ind_k = distribute_groups(7, 3, np.full(7, 9))
    assert M.K == 3 and M.model_name == 'm4b' and M.dphi == 6
    np.testing.assert_array_equal(M.engine.g_cnt, Nj_k)
    np.testing.assert_array_equal(np.diff(M.engine.g_lim), np.full(7, 9))
    assert M.engine.P == 6 + int(Nj_k.max()) * 3
    np.testing.assert_array_equal(M.workers[1].data['j_ind'] if hasattr(M.workers[1], 'data') else
                                  M.A_n['j_ind'][M.k_lim[1]:M.k_lim[2]], j_ind_k[M.k_lim[1]:M.k_lim[2]] + 1)
    res = fit.main('m4b', conf, verbose=False, _engine_factory=factory)
    assert res['m_s_ep'].shape == (3, 6) and np.all(np.isfinite(res['m_s_ep']))
    # the group structure is mandatory for the multi-group programs, and has to be contiguous
    data = M.X, M.y
    with pytest.raises(ValueError):
        Master('m4b', M.X, M.y, site_sizes=Nk, dphi=6, _engine_factory=factory)
    bad = j_ind_k + 1
    bad[[0, 1]] = bad[[1, 0]] if bad[0] != bad[1] else bad[[0, 1]]
    bad[3] = 2 if Nj_k[0] > 1 else bad[3]
    bad[4] = 1
    with pytest.raises(ValueError):
        Master('m4b', M.X, M.y, site_sizes=Nk, dphi=6, A_k={'J': Nj_k}, A_n={'j_ind': bad}, _engine_factory=factory)


def test_mix_phi_pools_the_tilted_samples(runs):
    """Master.mix_phi (method.py:1250-1296) against the pooled moments of the samples themselves."""
    M = _master(runs, 'smooth', 0.5)
    with pytest.raises(RuntimeError):
        M.mix_phi()
    assert M.run(2, verbose=False, seed=3)[0] == 0
    S, m = M.mix_phi()
    # from the per-site tilted moments (what the reference computes from saved_samp)
    means = np.stack([M.engine.get_tilted(k)[1] for k in range(M.K)])
    scat = sum(M.engine.get_tilted(k)[0] for k in range(M.K))
    nk = M.engine.get_tilted(0)[2]
    mref = means.mean(0)
    Sref = (scat + nk * sum(np.outer(mk - mref, mk - mref) for mk in means)) / (nk * M.K - 1)
    np.testing.assert_allclose(m, mref, rtol=1e-12)
    np.testing.assert_allclose(S, Sref, rtol=1e-10, atol=1e-14)


def test_site_schedule_puts_the_slow_sites_first():
    """Master._site_schedule: sites whose slowest chain is within LEAD_FRACTION of the slowest one
    lead (by slowest chain); the others follow by total work; no lead set without a tail."""
    from epstan_amd.method import Master
    lf = np.full((40, 4), 1000.0)
    lf[7] = [900, 50000, 800, 700]          # one slow chain
    lf[3] = [30000, 30000, 30000, 30000]    # all slow
    lf[20] = [5000, 5000, 5000, 5000]       # heavy, but not near the slowest
    order, n_lead = Master._site_schedule(lf.sum(axis=1), lf)
    assert n_lead == 2 and list(order[:3]) == [7, 3, 20]
    assert sorted(order) == list(range(40))
    # every site about equally slow (first iteration): nothing to single out
    order, n_lead = Master._site_schedule(np.full(40, 4e5), np.full((40, 4), 1e5))
    assert n_lead == 0 and sorted(order) == list(range(40))


def test_master_hands_the_schedule_of_the_next_launch_to_the_engine():
    """After every sampling launch `run` passes a dispatch order (a permutation of the local sites)
    and a lead-site count to the engine (engine.set_site_order / set_site_split)."""
    mod = models.MODELS['m1b'](4, 2, 12)
    data = mod.simulate_data(Sigma_x='rand', rng=3)
    _, _, Q0, r0 = mod.get_prior()
    calls = []

    def spying(model, X, y, k_lim, **groups):
        eng = OracleEngine(model, X, y, k_lim, **groups)
        eng.set_site_order = lambda order=None: calls.append(('order', None if order is None else list(order)))
        eng.set_site_split = lambda n: calls.append(('split', n))
        return eng

    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=2, iter=30, _engine_factory=spying)
    assert M.run(2, verbose=False, seed=5)[0] == 0
    orders = [c[1] for c in calls if c[0] == 'order']
    splits = [c[1] for c in calls if c[0] == 'split']
    assert len(orders) == 2 and len(splits) == 2
    assert all(sorted(o) == [0, 1, 2, 3] for o in orders)
    assert all(isinstance(n, int) and 0 <= n <= 1 for n in splits)
    M2 = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
                chains=2, iter=30, balance_sites=False, _engine_factory=spying)
    n0 = len(calls)
    assert M2.run(1, verbose=False, seed=5)[0] == 0
    assert len(calls) == n0


# ---------------------------------------------------------------- Gaussian-likelihood family
@pytest.mark.parametrize('name,tag,J,D,n,Sx', [('m1a', 'm1a_s', 5, 4, 20, 'rand'), ('m4a', 'm4a_s', 5, 4, 20, 'rand'),
                                              ('m1a', 'm1a_i', 3, 1, 10, None), ('m4a', 'm4a_i', 3, 6, 15, None),
                                              ('m3a', 'm3a_s', 5, 4, 20, 'rand'), ('m5a', 'm5a_s', 6, 5, (10, 30), None)])
def test_gaussian_family_simulators_match_the_reference(golden_dir, name, tag, J, D, n, Sx):
    """models.m1a / m4a against vectors of the imported reference (models/m1a.py, m4a.py;
    tests/golden/make_golden.py g11): data, true parameters, input scale, prior."""
    g = np.load(os.path.join(golden_dir, 'simulators_extra.npz'))
    mod = models.MODELS[name](J, D, n)
    data = mod.simulate_data(Sigma_x=Sx, rng=100)
    assert mod.dphi == int(g['g11_%s_dphi' % tag]) and mod.site_model == name + '_sg'
    np.testing.assert_allclose(data.X, g['g11_%s_X' % tag], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(data.y, g['g11_%s_y' % tag], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(data.phi_true, g['g11_%s_phi_true' % tag], rtol=1e-14)
    np.testing.assert_allclose(data.X_param['sigma_x'], g['g11_%s_sigma_x' % tag], rtol=1e-13)
    _, _, Q0, r0 = mod.get_prior()
    np.testing.assert_allclose(np.diag(Q0), g['g11_%s_Q0diag' % tag])
    np.testing.assert_allclose(r0, g['g11_%s_r0' % tag])
    assert Q0.flags['F_CONTIGUOUS']


@pytest.mark.parametrize('name,tag,J,D,n,Sx', [('m2b', 'm2b_s', 5, 4, 20, 'rand'), ('m3b', 'm3b_s', 5, 4, 20, 'rand'),
                                              ('m5b', 'm5b_s', 5, 4, 20, 'rand'), ('m3b', 'm3b_r', 6, 5, (10, 30), None),
                                              ('m5b', 'm5b_r', 6, 5, (10, 30), None)])
def test_remaining_logistic_simulators_match_the_reference(golden_dir, name, tag, J, D, n, Sx):
    """models.m2b / m3b / m5b against vectors of the imported reference (make_golden.py g12)."""
    g = np.load(os.path.join(golden_dir, 'simulators_extra.npz'))
    mod = models.MODELS[name](J, D, n)
    data = mod.simulate_data(Sigma_x=Sx, rng=100)
    np.testing.assert_array_equal(data.Nj, g['g12_%s_Nj' % tag])
    np.testing.assert_allclose(data.X, g['g12_%s_X' % tag], rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(data.y, g['g12_%s_y' % tag])
    np.testing.assert_allclose(data.phi_true, g['g12_%s_phi_true' % tag], rtol=1e-14)
    _, _, Q0, r0 = mod.get_prior()
    np.testing.assert_allclose(np.diag(Q0), g['g12_%s_Q0diag' % tag])
    np.testing.assert_allclose(r0, g['g12_%s_r0' % tag])
    assert mod.site_model == name + '_sg' and Q0.shape == (mod.dphi, mod.dphi)


def test_m2a_simulator_runs_where_the_reference_raises():
    """models/m2a.py names its generator `rnd_data` but receives `rng` (NameError): the mirror
    follows the evident intent; shape, prior and determinism only."""
    mod = models.MODELS['m2a'](5, 4, 20)
    a, b = mod.simulate_data(rng=100), mod.simulate_data(rng=100)
    np.testing.assert_array_equal(a.y, b.y)
    assert a.X.shape == (100, 4) and a.y.dtype == np.float64 and mod.dphi == 3
    np.testing.assert_allclose(np.diag(mod.get_prior()[2]), 1 / 1.5**2)


def test_master_runs_a_gaussian_family_model_on_the_oracle_engine():
    """EP with the m1a site model end to end (host logic + oracle sampler): the real-valued
    responses reach the engine unchanged and the posterior of beta moves to the simulated truth."""
    mod = models.MODELS['m1a'](4, 2, 30)
    data = mod.simulate_data(rng=11)
    _, _, Q0, r0 = mod.get_prior()
    seen = {}

    def spying(model, X, y, k_lim, **groups):
        seen['model'], seen['dtype'] = model, np.asarray(y).dtype
        return OracleEngine(model, X, y, k_lim, **groups)

    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=2, iter=120, _engine_factory=spying)
    info, (m_s, S_s) = M.run(3, verbose=False, seed=2)
    assert info == 0 and seen['model'] == 'm1a_sg' and seen['dtype'] == np.float64
    m, S = m_s[-1], S_s[-1]
    assert m.shape == (4,) and np.all(np.linalg.eigvalsh(S) > 0)
    # beta (elements 2, 3) within a few posterior standard deviations of the truth
    assert np.all(np.abs(m[2:] - data.phi_true[2:]) < 5 * np.sqrt(np.diag(S)[2:]) + 0.2)


# ------------------------------------------------------------------ round 2: launch + communicator plumbing
def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` outside torchrun spawns two fresh ranks (here without GPUs:
    --dry-run stops after the rendezvous) and rank 0 reports n_gpus = 2."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2 and rec['dry_run'] is True


def test_epxcomm_hands_the_rccl_id_from_rank_0_to_every_rank():
    """The 128-byte id exchange of dist.EpxComm (plain TCP, late and early joiners alike)."""
    import socket
    import threading
    import time as _time
    from epstan_amd import _lib, dist
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    uid = bytes(range(128))
    got = {}

    def rank(r, delay):
        _time.sleep(delay)
        c = dist.EpxComm(rank=r, world=3, addr='127.0.0.1', port=port)
        got[r] = c._exchange_id(uid if r == 0 else b'')

    th = [threading.Thread(target=rank, args=(r, d)) for r, d in ((1, 0.0), (0, 0.3), (2, 0.6))]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert got == {0: uid, 1: uid, 2: uid} and len(uid) == _lib.COMM_ID_BYTES
    with pytest.raises(ValueError):
        dist.EpxComm(rank=3, world=3)


def test_named_draws_restates_the_transformed_parameters():
    """site_params.named_draws against the transformed-parameters blocks of
    experiment/models/m4b_sg.stan:24-37, m1b_sg.stan, m4a_sg.stan and the multi-group m4b.stan:29-44."""
    from epstan_amd import site_params as sp
    rng = np.random.RandomState(0)
    D, S = 3, 7
    th = rng.randn(S, 3 * D + 3)                            # m4b_sg: phi (2D+2), eta, etb (D)
    out = sp.named_draws(3, D, 1, False, True, th, ['alpha', 'beta', 'sigma_a', 'mu_b', 'phi', 'eta', 'etb'])
    phi, eta, etb = th[:, :2 * D + 2], th[:, 2 * D + 2], th[:, 2 * D + 3:]
    np.testing.assert_allclose(out['alpha'], phi[:, 0] + eta * np.exp(phi[:, 1]))
    np.testing.assert_allclose(out['beta'], phi[:, 2:2 + D] + etb * np.exp(phi[:, 2 + D:]))
    np.testing.assert_allclose(out['sigma_a'], np.exp(phi[:, 1]))
    assert out['alpha'].shape == (S,) and out['beta'].shape == (S, D) and out['phi'].shape == (S, 2 * D + 2)
    th1 = rng.randn(S, D + 2)                               # m1b_sg: phi = [log sigma_a, beta], eta
    o1 = sp.named_draws(0, D, 1, False, True, th1, ['alpha', 'beta'])
    np.testing.assert_allclose(o1['alpha'], th1[:, D + 1] * np.exp(th1[:, 0]))
    np.testing.assert_allclose(o1['beta'], th1[:, 1:D + 1])
    tha = rng.randn(S, 3 * D + 4)                           # m4a_sg: log sigma in front
    oa = sp.named_draws(3, D, 1, True, True, tha, ['sigma', 'alpha'])
    np.testing.assert_allclose(oa['sigma'], np.exp(tha[:, 0]))
    np.testing.assert_allclose(oa['alpha'], tha[:, 1] + tha[:, 2 * D + 3] * np.exp(tha[:, 2]))
    ng = 2                                                  # m4b (two groups): eta (2), etb (2 x D)
    thg = rng.randn(S, 2 * D + 2 + ng + ng * D)
    og = sp.named_draws(3, D, ng, False, False, thg, ['alpha', 'beta'])
    assert og['alpha'].shape == (S, ng) and og['beta'].shape == (S, ng, D)
    np.testing.assert_allclose(og['beta'][:, 1, :], thg[:, 2:2 + D] + thg[:, 2 * D + 2 + ng + D:] * np.exp(thg[:, 2 + D:2 + 2 * D]))
    with pytest.raises(ValueError):
        sp.named_draws(0, D, 1, False, True, th1, ['etb'])


def test_inline_assembly_behind_a_matrix_result_keeps_its_wait_states(tmp_path):
    """gfx950 leaves the MFMA -> VALU read distance to software and LLVM's hazard recogniser does not look into inline
    assembly: the row team's logistic clamp (`logistic_pair_lean`, epx_device.h) may be handed
    the result of v_mfma_f64_4x4x4 directly and has to bring its own 6 wait states.  Compile exactly that and count."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = tmp_path / 'haz.hip'
    src.write_text('#include <hip/hip_runtime.h>\n#include "epx_device.h"\n'
                   '__global__ void k(double *x, double a, double b) {\n'
                   '    double f0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x[threadIdx.x], 0, 0, 0);\n'
                   '    double f1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, x[threadIdx.x + 64], 0, 0, 0);\n'
                   '    double l0, l1, w0, w1, g0, g1;\n'
                   '    epx::logistic_pair_lean(f0, f1, 1.0, 0.0, l0, l1, w0, w1, g0, g1);\n'
                   '    x[threadIdx.x] = l0 + l1 + w0 + w1 + g0 + g1;\n}\n')
    out = tmp_path / 'haz.s'
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only', '-S',
                    '-I', os.path.join(ROOT, 'ep-stan_amd', 'csrc'), str(src), '-o', str(out)],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    ins = [l.strip() for l in out.read_text().splitlines() if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    last_mfma = max(i for i, l in enumerate(ins) if l.startswith('v_mfma_f64_4x4x4'))
    first_max = min(i for i, l in enumerate(ins) if l.startswith('v_max_f64') and i > last_mfma)
    waits = 0
    for l in ins[last_mfma + 1:first_max]:
        m = re.match(r's_nop (\d+)', l)
        waits += int(m.group(1)) + 1 if m else 1
    assert waits >= 6, ins[last_mfma:first_max + 1]


def test_row_dma_reads_its_scalar_base_five_wait_states_behind_readfirstlane(tmp_path):
    """The second software-managed distance of the hand-written assembly: a VMEM instruction that
    reads an SGPR which a VALU instruction (v_readfirstlane) wrote needs 5 wait states.  The LDS-DMA pieces of the
    streaming sampler's ring take their scalar base that way (`ring_issue`, epx_stream_tile.h): compile it and count."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = tmp_path / 'dma.hip'
    src.write_text('#include <hip/hip_runtime.h>\n#include "epx_stream_tile.h"\n'
                   '__global__ void k(const double *X, const int *y, int D, int tt, int slot) {\n'
                   '    extern __shared__ double smem[];\n'
                   '    epx::PassArgs<128> s;\n'
                   '    s.Xg = X; s.yg = y; s.gauss = 0; s.n = 2000; s.D = D; s.ntile = 125; s.ngmax = 1; s.ntmax = 125;\n'
                   '    s.lds0 = (unsigned)(size_t)smem; s.slot_f = 0; s.slot_i = 0; s.t_i = 0; s.wave = 4; s.lane = threadIdx.x;\n'
                   '    epx::loader_init<128>(s, threadIdx.x);\n'
                   '    const epx::StreamMap M = epx::stream_map<128>(1, 125, 0);\n'
                   '    epx::ring_issue<128>(s, M, tt, slot, threadIdx.x);\n}\n')
    out = tmp_path / 'dma.s'
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only', '-S',
                    '-I', os.path.join(ROOT, 'ep-stan_amd', 'csrc'), str(src), '-o', str(out)],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    # the statement has to be self-sufficient: wherever the compiler puts the v_readfirstlane, the wait states INSIDE the
    # assembly block in front of its first scalar-base DMA are at least 5
    blocks = out.read_text().split(';;#ASMSTART')[1:]
    checked = 0
    for b in blocks:
        body = [l.strip() for l in b.split(';;#ASMEND')[0].splitlines() if l.strip()]
        loads = [i for i, l in enumerate(body) if re.match(r'global_load_lds_dwordx4 v\d+, s\[', l)]
        if not loads:
            continue
        waits = 0
        for l in body[:loads[0]]:
            n = re.match(r's_nop (\d+)', l)
            waits += int(n.group(1)) + 1 if n else 1
        assert waits >= 5, body[:loads[0] + 1]
        assert len(loads) == 16                        # the 16 pieces of a full tile, every later one behind an M0 update + s_nop
        checked += 1
    assert checked >= 1


def test_pooled_z_statistics_see_a_shared_bias_that_per_coordinate_bounds_miss():
    """bench.pooled_z (the statistical leg of the parity record): z-scores that all stay within 4
    but share a bias of 0.3 standard errors give a large t-statistic of the per-site means; unbiased ones do not."""
    sys.path.insert(0, ROOT)
    import bench
    rng = np.random.RandomState(3)
    z0 = rng.randn(32, 66)
    z1 = z0 + 0.3
    a, b = bench.pooled_z(z0, 'unbiased'), bench.pooled_z(z1, 'biased')
    assert np.abs(z1).max() < 4.5                                   # "share_within_4" would pass both
    assert abs(a['mean_of_site_means_t_statistic']) < 3.5 and abs(b['mean_of_site_means_t_statistic']) > 8.0
    assert 0.85 < a['variance'] < 1.15 and 0.01 < a['chi2_p_value_nominal'] < 0.99
    assert a['n'] == 32 * 66 and a['sites'] == 32
    # a scatter twice as wide as expected is far outside the chi-square's noise
    assert bench.pooled_z(2.0 * z0, 'wide')['chi2_p_value_nominal'] < 1e-6
