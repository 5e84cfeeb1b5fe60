"""BASELINE config C5 at its OWN size, once (named so that pytest collects it LAST: ~5-6 min).

J = 4096 sites, D = 128, n_j = 2000 (d = 258, P = 387), prec_estim = 'olse', the damped path of
/root/reference/experiment/find_damp.py:105-236 driven as /root/reference/experiment/fit.py:326-335 drives it: ONE rank,
device 0, one EP iteration with the real sampler (the streaming kernel, layout 3, from the piece queue: 4096 site updates
of 2 MB of rows each, 8.4 GB of rows resident in HBM, three 2.18 GB site arrays) and, inside that iteration, the damping
sweep over the reference's 31 factors scored against a target.  On the 8-GPU node every rank holds an eighth of this; the
per-rank code is what tests/test_gpu_multirank.py::test_two_c5_shards_two_ranks_on_one_device runs."""

import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c5_at_its_own_size_one_rank_one_ep_iteration_with_the_damping_sweep():
    from epstan_amd import find_damp, models
    from epstan_amd.method import Master
    J, D, n = 4096, 128, 2000
    t0 = time.time()
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(rng=100)                   # fit.py's cor_input=False branch (the vine matrix is not positive definite from D ~ 120 on)
    _, _, Q0, r0 = mod.get_prior()
    t_data = time.time() - t0
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
               prec_estim='olse', df0=models.default_df0(J), sync_sites=True)
    d = M.dphi
    assert (M.K, d, M.engine.P) == (J, 2 * D + 2, 3 * D + 3)
    # a target for the sweep's criteria (find_damp.py:121-173 scores against a full-posterior fit; here: the prior's
    # moments shrunk -- any positive definite target exercises the 31 x (global Cholesky + 4096 cavities) of the sweep)
    S_t = np.linalg.inv(Q0) * 0.25
    m_t = np.zeros(d)
    damps = find_damp.default_damps()
    assert damps.shape == (31,)
    t0 = time.time()
    info, (m_s, S_s), an = M.run(1, verbose=False, return_analytics=True, seed=1,
                                 sweep=dict(damps=damps, m_target=m_t, S_target=S_t, samp_target=None))
    t_iter = time.time() - t0
    eng = M.engine
    stats = M.last_site_stats
    assert info == 0
    assert eng.last_layout() == 3 and eng.last_segments() < 0          # the streaming kernel, from the piece queue
    assert stats.shape[0] == J and stats[:, 7].sum() == 0 and stats[:, 2].min() > 0       # no site failed, every site sampled
    assert np.all(np.isfinite(stats[:, 0])) and np.all(stats[:, 0] > 0)                      # mean step sizes
    # the damping sweep of this iteration: 31 factors, criteria finite wherever the proposal is positive definite, and the
    # factor the iteration then took is one the sweep found admissible
    sw = M.sweep_log[-1]
    assert sw['mses'].shape == (31,) and sw['kls'].shape == (31,)
    ok = np.isfinite(sw['kls'])
    assert ok.any() and np.all(np.isfinite(sw['mses'][ok])) and np.all(sw['kls'][ok] > -1e-9)
    df = M.df_log[-1]
    assert 0 < df <= models.default_df0(J)(1) + 1e-15
    # global approximation: symmetric positive definite, and Q = Q0 + the sum of ALL 4096 accepted site precisions
    # (method.py:1073-1074), r likewise
    S, m = M.cur_approx()
    assert np.all(np.isfinite(m_s)) and np.all(np.isfinite(S_s))
    np.testing.assert_allclose(S, S.T, rtol=1e-9, atol=1e-13)
    assert np.linalg.eigvalsh(S)[0] > 0
    np.testing.assert_allclose(M.Q, M.Q0 + M.Qi.sum(axis=2), rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(M.r, M.r0 + M.ri.sum(axis=1), rtol=1e-9, atol=1e-6)
    # a site's tilted moments are those of its 400 draws (the olse precision estimate is built on them)
    for k in (0, 2049, 4095):
        samp = eng.get_draws(k)
        Mat, vec, nsamp = eng.get_tilted(k)
        c = samp - samp.mean(axis=0)
        assert nsamp == 400 and np.abs(vec - samp.mean(axis=0)).max() < 1e-10
        assert np.abs(Mat - c.T.dot(c)).max() / np.abs(Mat).max() < 1e-8
    passes = float(np.sum(M.pass_log[-1]))
    ms = float(M.sampling_ms[-1])
    print('C5 at J = 4096 on one device: data %.0f s, EP iteration %.0f s (sampler launch %.1f s, %.3g row passes = %.2f TB/s '
          'of rows + responses + cavity precision), df = %.4g, %d of 31 damping factors admissible'
          % (t_data, t_iter, ms * 1e-3, passes, passes * (n * D * 8 + n * 4 + d * d * 8) / (ms * 1e-3) / 1e12, df, int(ok.sum())))
