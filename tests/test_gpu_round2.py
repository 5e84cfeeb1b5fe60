"""Round-2 device tests: the row-wave / state-wave sampler (layouts 5, 6), carried adaptation,
the BASELINE configs at their own size, the fused update phase, named parameter draws.

Everything goes through the C ABI (ctypes).  References are the oracle (oracle/), golden vectors
produced by the imported reference (tests/golden/), and bit-equality between layouts that run the
same arithmetic."""

import os

import numpy as np
import pytest

from epstan_amd import _lib, models
from epstan_amd.engine import DQI, QI, HipEngine
from epstan_amd.method import Master, Worker
from oracle import ep_oracle as eo
from oracle import nuts_oracle as no
from test_gpu_parity import _engine_with_cavity, _site_problem
from conftest import record_slack

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _all_draws(eng, K):
    return np.stack([eng.get_draws(k, all_params=True) for k in range(K)])


# ------------------------------------------------------------------ layouts 5 / 6 (nuts_duo.hip)
@pytest.mark.parametrize('model,D,n,chains,it', [
    ('m4b_sg', 16, 200, 4, 44),      # C2 site: one register per vector, everything in LDS
    ('m4b_sg', 32, 500, 4, 30),      # C3 site: d = 66 (two tail rows of the cavity precision), stack + cold store in HBM
    ('m4b_sg', 22, 300, 3, 40),      # two registers per vector, d = 46 < 64, odd chain count
    ('m1b_sg', 16, 120, 4, 44), ('m2b_sg', 21, 333, 2, 44), ('m3b_sg', 32, 300, 4, 44), ('m5b_sg', 32, 150, 8, 30),
    ('m4b_sg', 9, 77, 1, 44),        # odd D: padded columns
])
def test_row_and_state_waves_give_the_draws_of_one_wave_per_chain(model, D, n, chains, it):
    """Layout 5 splits a chain over a row wave and a state wave that runs the tree bookkeeping one
    leapfrog behind; every accepted state is the sequential algorithm's, in the same arithmetic
    order: draws, last states and every statistic equal layout 1's bit for bit."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 31 + D, K=3)
    eng, _, _ = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.array([5, 6, 7], dtype=np.int64)
    out = {}
    for layout in (1, 5):
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=chains, iter=it, init='random', layout=layout))
        assert eng.last_layout() == layout
        out[layout] = (_all_draws(eng, 3), eng.get_chain_stats(chains))
    np.testing.assert_array_equal(out[5][0], out[1][0])
    np.testing.assert_array_equal(out[5][1], out[1][1])
    assert out[1][1][:, :, 7].sum() == 0
    # warm start from the previous call's last draws: still identical
    for layout in (1, 5):
        eng.sample_batch(seeds + 9, HipEngine.sampler_opts(chains=chains, iter=20, init='random', layout=layout))
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=chains, iter=20, init='prev', layout=layout))
        out[layout] = _all_draws(eng, 3)
    np.testing.assert_array_equal(out[5], out[1])


@pytest.mark.parametrize('model,D,n', [('m4b_sg', 16, 200), ('m4b_sg', 32, 500), ('m1b_sg', 32, 300), ('m5b_sg', 21, 333),
                                       ('m3b_sg', 11, 64), ('m2b_sg', 32, 100)])
def test_duo_gradients_match_oracle(model, D, n):
    """Every layout evaluates the density with its own gradient code: 5 (one row wave) and 6 (two row
    waves, partial sums added by the state wave; the bookkeeping on a wave of its own) against the C
    restatement.  Layout 6 keeps tree stack and mailbox in LDS: at the C3 site size the rows leave no room
    and the request falls back to layout 2."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 100 + D)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    rng = np.random.RandomState(8)
    for k in range(2):
        theta = rng.randn(P) * 0.5
        lo, hi = k_lim[k], k_lim[k + 1]
        lp_o, g_o = no.logdensity_grad(model, X[lo:hi], y[lo:hi], mu_dev[k], Om_dev[k], theta)
        for layout in (5, 6):
            lp, g = eng.logdensity_grad(k, theta, layout=layout)
            assert eng.last_layout() == (2 if layout == 6 and (D, n) == (32, 500) else layout)
            assert abs(lp - lp_o) <= 1e-11 * max(1.0, abs(lp_o))
            np.testing.assert_allclose(g, g_o, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(g_o).max()))


@pytest.mark.parametrize('chains,it,thin,init', [(2, 40, 3, '0'), (8, 24, 1, 'random'), (3, 30, 2, 'random')])
def test_duo_chain_counts_thinning_partial_batches(chains, it, thin, init):
    """chains != 4, thinning, init='0' and a sub-range of the sites (k0 > 0) through layout 5: the draws of
    layout 1, which the oracle-based tests pin (method.py:154-160, 579-583; find_damp.py uses 8 chains)."""
    X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 12, 70, 77, K=5, tight=100.0)
    eng, _, _ = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
    seeds = np.arange(5, dtype=np.int64) + 3
    out = {}
    for layout in (1, 5):
        opts = HipEngine.sampler_opts(chains=chains, iter=it, thin=thin, init=init, layout=layout)
        eng.sample_batch(seeds, opts)
        full = _all_draws(eng, 5)
        eng.sample_batch(seeds[2:4], opts, k0=2, count=2)                  # sites 2, 3 again, alone
        assert eng.last_layout() == layout
        part = _all_draws(eng, 5)
        np.testing.assert_array_equal(part, full)
        out[layout] = full
    np.testing.assert_array_equal(out[5], out[1])
    assert eng.num_draws() == chains * ((it - it // 2 + thin - 1) // thin)


@pytest.mark.parametrize('model,D,n', [('m4b_sg', 16, 200), ('m4b_sg', 32, 120)])
def test_layout_6_follows_the_oracle_run(model, D, n):
    """One workgroup per chain -- state wave, two row waves, bookkeeping wave: whole site updates against the C restatement,
    chain by chain until rounding differences are amplified past a decision (as for layouts 1, 2)."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 7 + D, K=3, tight=1000.)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.array([101, 202, 303], dtype=np.int64)
    it = 44
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=6))
    assert eng.last_layout() == 6
    draws_o, _, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=4, iter=it)
    cs = eng.get_chain_stats(4)
    n_full = 0
    for k in range(3):
        dev = eng.get_draws(k, all_params=True)
        ref = draws_o[k].reshape(-1, P)
        err = np.abs(dev - ref).reshape(4, it // 2, P).max(axis=2) / max(1.0, np.abs(ref).max())
        for c in range(4):
            assert np.all(err[c, :5] < 1e-3), (k, c, err[c, :5])
            if np.all(err[c] < 1e-4):
                n_full += 1
                assert cs[k, c, 3] == st_o[k, c, 3]
    print('layout 6: chains equal to the oracle to the end: %d of 12' % n_full)
    record_slack('layout 6 run vs oracle %s D=%d n=%d: chains equal to the end' % (model, D, n), n_full, '>= 9', 12)
    assert n_full >= 9, n_full


def test_layout_policy_prefers_the_duo_kernel_for_large_batches():
    """Auto layout: batches that fill the chip run the row-wave / state-wave kernel when the shape is
    instantiated (D padded to 16 or 32), the one-wave-per-chain kernel otherwise; an explicit layout
    is honoured; a lead split keeps working."""
    mod = models.m4b(330, 16, 60)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=12)
    eng = M.engine
    seeds = np.arange(330) + 1
    ref = {}
    for layout in (0, 1, 5, 2, 7):
        eng.set_site_order(None)
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=12, init='random', layout=layout, max_depth=6))
        ref[layout] = (_all_draws(eng, 330), eng.last_layout())
    assert [ref[l][1] for l in (0, 1, 5, 2, 7)] == [7, 1, 5, 2, 7]
    np.testing.assert_array_equal(ref[5][0], ref[1][0])
    np.testing.assert_array_equal(ref[0][0], ref[7][0])
    # split launch: the lead sites of the order run one workgroup per chain (layout 2's draws)
    order = np.arange(330, dtype=np.int32)[::-1].copy()
    eng.set_site_order(order)
    eng.set_site_split(9)
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=12, init='random', max_depth=6))
    m = eng.last_split()
    assert eng.last_layout() == 7 and 1 <= m <= 9
    dr = _all_draws(eng, 330)
    lead = order[:m]
    rest = order[m:]
    np.testing.assert_array_equal(dr[lead], ref[2][0][lead])
    np.testing.assert_array_equal(dr[rest], ref[7][0][rest])
    eng.set_site_order(None)
    eng.set_site_split(0)


# ------------------------------------------------------------------ Gaussian-likelihood family, streamed (layout 3)
@pytest.mark.parametrize('model,D,n', [('m1a_sg', 40, 150), ('m4a_sg', 128, 300), ('m3a_sg', 64, 100), ('m2a_sg', 100, 257),
                                       ('m5a_sg', 33, 90), ('m4a_sg', 16, 2500), ('m1a_sg', 8, 90)])
def test_gaussian_family_streams_when_the_site_is_not_lds_resident(model, D, n):
    """experiment/models/m{1..5}a_sg.stan on sites the resident kernels cannot hold (D > 32, or rows beyond the
    LDS): served by the streaming layout since round 2 (refused before).  Gradient against the oracle; the
    last shape is resident-size and takes the streaming layout on request."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 100 + D)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    rng = np.random.RandomState(5)
    for k in range(2):
        for trial in range(2):
            theta = rng.randn(P) * (0.2 + 0.4 * trial)
            lo, hi = k_lim[k], k_lim[k + 1]
            lp_o, g_o = no.logdensity_grad(model, X[lo:hi], y[lo:hi], mu_dev[k], Om_dev[k], theta)
            lp, g = eng.logdensity_grad(k, theta, layout=3 if D <= 8 else 0)
            assert eng.last_layout() == 3
            assert abs(lp - lp_o) <= 1e-10 * max(1.0, abs(lp_o)), (lp, lp_o)
            np.testing.assert_allclose(g, g_o, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(g_o).max()))


@pytest.mark.parametrize('model,D,n,chains', [('m4a_sg', 40, 100, 4), ('m1a_sg', 16, 120, 3)])
def test_gaussian_family_streamed_run_follows_the_oracle(model, D, n, chains):
    """Whole site updates of the Gaussian family through layout 3, chain by chain against the C restatement."""
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 9 + D, K=3, tight=1000.)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.array([11, 22, 33], dtype=np.int64)
    it = 44
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=chains, iter=it, init='random', layout=3))
    assert eng.last_layout() == 3
    draws_o, _, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=chains, iter=it)
    cs = eng.get_chain_stats(chains)
    n_full = 0
    for k in range(3):
        dev = eng.get_draws(k, all_params=True)
        ref = draws_o[k].reshape(-1, P)
        err = np.abs(dev - ref).reshape(chains, it // 2, P).max(axis=2) / max(1.0, np.abs(ref).max())
        for c in range(chains):
            assert np.all(err[c, :5] < 1e-3), (k, c, err[c, :5])
            if np.all(err[c] < 1e-4):
                n_full += 1
                assert cs[k, c, 3] == st_o[k, c, 3]
    print('Gaussian family, layout 3: chains equal to the oracle to the end: %d of %d' % (n_full, 3 * chains))
    record_slack('Gaussian family streamed vs oracle: chains equal to the end', n_full, '>= %d' % ((3 * chains * 3) // 4), 3 * chains)
    assert n_full >= (3 * chains * 3) // 4, n_full


def test_gaussian_family_with_many_groups_per_site_streams():
    """m4a with 8 groups per site: 35 + 8 * 17 = 171 sampled coordinates, beyond the resident multi-group kernel:
    gradient and a short site update through layout 3 against the oracle (experiment/models/m4a.stan)."""
    rng = np.random.RandomState(3)
    D, groups = 16, [[20] * 8, [15, 25, 20, 20, 20, 20, 20, 20]]
    sizes = [sum(g) for g in groups]
    N = sum(sizes)
    X = rng.randn(N, D)
    y = 0.3 + X.dot(rng.randn(D) * 0.4) + 0.7 * rng.randn(N)
    k_lim = np.concatenate(([0], np.cumsum(sizes)))
    g_cnt = np.array([len(g) for g in groups], dtype=np.int32)
    g_lim = np.concatenate(([0], np.cumsum([n for g in groups for n in g])))
    eng = HipEngine('m4a', X, y, k_lim, g_cnt=g_cnt, g_lim=g_lim)
    d = eng.d
    A = rng.randn(d, d + 3)
    Om = A.dot(A.T) / (d + 3) + 0.5 * np.eye(d)
    mu = 0.3 * rng.randn(d)
    for k in range(2):
        assert eng.cavity_site(k, Om + np.eye(d), Om.dot(mu), np.eye(d), np.zeros(d))
    Om_dev = np.stack([eng.get_cavity(k)[0] for k in range(2)])
    mu_dev = np.stack([eng.get_cavity(k)[1] for k in range(2)])
    assert eng.P == d + 8 * (1 + D)
    off = np.concatenate(([0], np.cumsum(g_cnt)))
    for k in range(2):
        theta = 0.3 * rng.randn(eng.P)
        gl = g_lim[off[k]:off[k + 1] + 1] - k_lim[k]
        lp_o, g_o = no.logdensity_grad('m4a', X[k_lim[k]:k_lim[k + 1]], y[k_lim[k]:k_lim[k + 1]], mu_dev[k], Om_dev[k], theta, gl=gl)
        lp, g = eng.logdensity_grad(k, theta, layout=0)
        assert eng.last_layout() == 3
        assert abs(lp - lp_o) <= 1e-10 * max(1.0, abs(lp_o))
        np.testing.assert_allclose(g, g_o, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(g_o).max()))
    stats, ms = eng.sample_batch(np.array([5, 6]), HipEngine.sampler_opts(chains=2, iter=24, init='random'))
    assert eng.last_layout() == 3 and np.all(np.isfinite(stats)) and stats[:, 7].sum() == 0


# ------------------------------------------------------------------ adapt = 'carry'
def test_carried_adaptation_history_and_second_call_follow_the_oracle():
    """`adapt='carry'` (opt-in, not the reference's behaviour): (1) without history the call IS a
    fresh one; (2) the history the library keeps (final step sizes, pooled regularised variances)
    equals the NumPy statement of it; (3) the next call, started from that history, follows the C
    restatement given the same history, chain by chain until they part."""
    model, D, n, it = 'm4b_sg', 16, 200, 60
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 23, K=3, tight=1000.)
    seeds = np.array([41, 42, 43], dtype=np.int64)
    for layout in (1, 2, 5):
        eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
        eng2, _, _ = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
        eng2.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=layout))
        fresh = _all_draws(eng2, 3)
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=layout, adapt='carry'))
        np.testing.assert_array_equal(_all_draws(eng, 3), fresh)                 # (1): no history yet
        cs = eng.get_chain_stats(4)
        eps_h, met_h = no.carry_history(_all_draws(eng, 3).reshape(3, 4, it // 2, P), cs)
        for k in range(3):                                                      # (2)
            e, m = eng.get_adapt(k, 4)
            np.testing.assert_array_equal(e, cs[k, :, 1])
            np.testing.assert_allclose(e, eps_h[k], rtol=0)
            np.testing.assert_allclose(m, met_h[k], rtol=1e-11)
        hist = [eng.get_adapt(k, 4) for k in range(3)]
        last = _all_draws(eng, 3).reshape(3, 4, it // 2, P)[:, :, -1, :]
        eng.sample_batch(seeds + 50, HipEngine.sampler_opts(chains=4, iter=it, init='prev', layout=layout, adapt='carry'))
        cs2 = eng.get_chain_stats(4)
        draws_o, _, st_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds + 50, chains=4, iter=it, init=last,
                                         carry_eps=np.stack([h[0] for h in hist]),
                                         carry_metric=np.stack([h[1] for h in hist]))
        n_full = 0
        for k in range(3):                                                      # (3)
            dev = eng.get_draws(k, all_params=True)
            ref = draws_o[k].reshape(-1, P)
            err = np.abs(dev - ref).reshape(4, it // 2, P).max(axis=2) / max(1.0, np.abs(ref).max())
            for c in range(4):
                assert np.all(err[c, :5] < 1e-3), (layout, k, c, err[c, :5])
                if np.all(err[c] < 1e-4):
                    n_full += 1
                    assert cs2[k, c, 3] == st_o[k, c, 3]
        print('carry, layout %d: chains equal to the oracle to the end: %d of 12' % (layout, n_full))
        record_slack('carried adaptation vs oracle, layout %d: chains equal to the end' % layout, n_full, '>= 9', 12)
        assert n_full >= 9, (layout, n_full)
        assert np.all(np.isfinite(cs2[:, :, 1])) and np.all(cs2[:, :, 1] > 0)


def test_ep_with_carried_adaptation_agrees_with_the_fresh_path_and_needs_fewer_leapfrogs():
    """Same target distribution: the EP posterior with adapt='carry' agrees with adapt='fresh' within
    the Monte-Carlo spread of two fresh seeds; the carried runs need fewer gradients per iteration."""
    mod = models.m4b(24, 8, 120)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()

    def run(adapt, seed):
        M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=400,
                   df0=models.default_df0(24), adapt=adapt)
        info, (m_s, S_s) = M.run(8, verbose=False, seed=seed)
        assert info == 0
        return m_s[-1], np.sqrt(np.diag(S_s[-1])), np.array(M.ngrad_log)

    m_a, s_a, g_a = run('fresh', 1)
    m_b, s_b, g_b = run('fresh', 2)
    m_c, s_c, g_c = run('carry', 1)
    spread = np.abs(m_a - m_b) / s_a
    dev = np.abs(m_c - m_a) / s_a
    print('EP mean, |fresh(1) - fresh(2)| / sd: max %.2f ; |carry - fresh| / sd: max %.2f ; gradients per iteration '
          'fresh %.3g, carry %.3g' % (spread.max(), dev.max(), g_a[2:].mean(), g_c[2:].mean()))
    assert dev.max() < max(3.0 * spread.max(), 0.6)
    np.testing.assert_allclose(s_c, s_a, rtol=0.5)
    assert g_c[2:].mean() < g_a[2:].mean()


# ------------------------------------------------------------------ BASELINE configs at their own size
def test_one_ep_iteration_at_c3_size_with_invariants():
    """C3 = C4 per GPU: J = 512 sites, D = 32, n_j = 500, m4b, 4 x 200.  Two EP iterations through the
    default path (layout 7: row team on the matrix pipe, cold store in HBM, fused update): finite, positive definite,
    symmetric, site sums consistent with the global approximation, one site's moment stage against
    the oracle from the device's own draws, and the split launch reproduces the unsplit draws."""
    J, D, n = 512, 32, 500
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
               df0=models.default_df0(J))
    eng = M.engine
    info, (m_s, S_s), (st, ms, rh, ot) = M.run(2, verbose=False, return_analytics=True, seed=1)
    assert info == 0 and eng.last_layout() == 7
    assert np.all(np.isfinite(m_s)) and np.all(np.isfinite(S_s))
    for S in S_s:
        np.testing.assert_allclose(S, S.T, rtol=1e-10, atol=1e-14)
        assert np.linalg.eigvalsh(S)[0] > 0
    np.testing.assert_allclose(M.Q, M.Q0 + M.Qi.sum(axis=2), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(M.r, M.r0 + M.ri.sum(axis=1), rtol=1e-9, atol=1e-8)
    stats = M.last_site_stats
    assert stats[:, 7].sum() == 0 and np.all(stats[:, 2] > 0)
    print('C3: leapfrogs per transition %.0f, sampling launches %s ms, update phase %s ms'
          % (stats[:, 2].sum() / (J * 4 * 200), np.round(M.sampling_ms, 1), np.round(ot * 1e3, 2)))
    # one site's moment stage from the device's own draws (method.py:413-458)
    k = 137
    samp = eng.get_draws(k)
    Mat, vec, nsamp = eng.get_tilted(k)
    assert nsamp == 400
    np.testing.assert_allclose(vec, samp.mean(axis=0), rtol=1e-10, atol=1e-12)
    c = samp - samp.mean(axis=0)
    np.testing.assert_allclose(Mat, c.T.dot(c), rtol=1e-8, atol=1e-9)
    # the split launch (lead sites one workgroup per chain) against explicit layouts on the same state
    seeds = np.arange(J) + 77
    ref = {}
    eng.set_site_order(None)
    for layout in (7, 2):
        eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=16, init='random', layout=layout, max_depth=7))
        ref[layout] = _all_draws(eng, J)
    order = np.argsort(-stats[:, 2]).astype(np.int32)
    eng.set_site_order(order)
    eng.set_site_split(12)
    eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=16, init='random', max_depth=7))
    m = eng.last_split()
    assert m >= 1
    dr = _all_draws(eng, J)
    np.testing.assert_array_equal(dr[order[:m]], ref[2][order[:m]])
    np.testing.assert_array_equal(dr[order[m:]], ref[7][order[m:]])


def test_one_damped_iteration_at_c5_site_size_through_the_sweep():
    """C5 site shape (D = 128, n_j = 2000, d = 258; rows streamed from HBM, dense kernels on the
    global workspace) on 64 sites: one EP iteration through run(..., sweep=) with prec_estim='olse'
    (S = 400 draws against d = 258: the `sample` estimator's S > d + 2 barely holds)."""
    J, D, n = 64, 128, 2000
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(rng=100)                     # uncorrelated covariates (HISTORY.md section 6, footnote)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
               prec_estim='olse', df0=0.2)
    d = M.dphi
    assert d == 258
    S0, m0 = M.cur_approx()
    damps = 0.85 ** np.arange(31)
    sw = dict(damps=damps, m_target=m0, S_target=S0)
    info, (m_s, S_s) = M.run(1, verbose=False, seed=5, sweep=sw)
    assert info == 0 and M.engine.last_layout() == 3
    log = M.sweep_log[0]
    assert log['global_pd'].shape == (31,) and np.all(np.isfinite(log['kls'][log['cav_pd']]))
    assert log['cav_pd'].any()
    np.testing.assert_allclose(S_s[0], S_s[0].T, rtol=1e-9, atol=1e-13)
    assert np.linalg.eigvalsh(S_s[0])[0] > 0
    np.testing.assert_allclose(M.Q, M.Q0 + M.Qi.sum(axis=2), rtol=1e-9, atol=1e-8)
    # the accepted factor's criteria against a host computation on the downloaded state
    i = int(np.argmin(np.abs(damps - M.df_log[-1])))
    if abs(damps[i] - M.df_log[-1]) < 1e-12 and log['cav_pd'][i]:
        mse = np.mean((m_s[0] - m0) ** 2)
        np.testing.assert_allclose(log['mses'][i], mse, rtol=1e-6)


# ------------------------------------------------------------------ host contract on the device
def test_master_init_golden_partition_and_init_site_on_gpu(golden_dir):
    """G5 (vectors produced by the imported reference's Master.__init__): partition arrays, initial
    global approximation and cavities with init_site, on the DEVICE engine (method.py:696-730, 853-882)."""
    alg = np.load(os.path.join(golden_dir, 'algebra.npz'))
    X, y, sizes = alg['g5_X'], alg['g5_y'], alg['g5_sizes']
    M = Master('m1b_sg', X, y, site_sizes=sizes, dphi=4, init_site=3.0)
    assert isinstance(M.engine, HipEngine)
    np.testing.assert_array_equal(M.k_lim, alg['g5_k_lim'])
    np.testing.assert_array_equal(M.k_ind, alg['g5_k_ind'])
    np.testing.assert_allclose(M.Q, alg['g5_Q'], rtol=1e-12)
    np.testing.assert_allclose(M.Qi, alg['g5_Qi'], rtol=1e-12)
    S, m = M.cur_approx()
    np.testing.assert_allclose(S, alg['g5_S'], rtol=1e-11)
    ind_ord = np.repeat(np.arange(3), sizes)
    M2 = Master('m1b_sg', X, y, site_ind_ord=ind_ord, dphi=4)
    np.testing.assert_array_equal(M2.k_lim, alg['g5_ord_k_lim'])
    np.testing.assert_allclose(M2.workers[1].Mat, alg['g5_w1_Mat'], rtol=1e-12)
    np.testing.assert_allclose(M2.workers[1].vec, alg['g5_w1_vec'], atol=1e-14)
    perm = np.random.RandomState(0).permutation(30)
    M3 = Master('m1b_sg', X[perm], y[perm], site_ind=ind_ord[perm], dphi=4)
    np.testing.assert_array_equal(M3.k_lim, alg['g5_k_lim'])
    with pytest.raises(ValueError):
        Master('m1b_sg', X, y, site_sizes=np.array([7, 0, 23]), dphi=4)


def test_force_pd_keeps_the_tilted_moments(golden_dir):
    """the force-pd fallback used the tilted means as scratch.  After an iteration that goes
    through it, mix_phi and Worker.vec still equal the pooled moments of the injected draws."""
    import injectors
    runs = np.load(os.path.join(golden_dir, 'master_run.npz'))
    Nj = runs['g6_Nj'][:3]
    nrow = int(Nj.sum())
    M = Master('m1b_sg', runs['g6_X'][:nrow], runs['g6_y'][:nrow], site_sizes=Nj,
               prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']}, A_k={'site_id': np.arange(3)}, chains=4, iter=200,
               df0=1.0, df_treshold=0.9)
    inj = injectors.GaussianTilted('wide_first')
    M._sample_injector = inj
    seen = []
    M._sample_injector = lambda data, sp: seen.append(inj(data, sp)) or seen[-1]
    info = M.run(1, verbose=False, calc_moments=False, seed=1)
    assert info == 0
    samp = seen[-3:]
    S_mix, m_mix = M.mix_phi()
    pooled = np.concatenate(samp)
    means = np.array([s.mean(axis=0) for s in samp])
    np.testing.assert_allclose(m_mix, means.mean(axis=0), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(S_mix, np.cov(pooled.T, ddof=1) if False else
                               (sum((s - s.mean(0)).T.dot(s - s.mean(0)) for s in samp)
                                + samp[0].shape[0] * (means - means.mean(0)).T.dot(means - means.mean(0))) / (pooled.shape[0] - 1),
                               rtol=1e-9, atol=1e-11)


def test_named_parameter_draws_from_the_device():
    """Worker.tilted(save_samples=...) / Master.run(save_last_param=...) by parameter name
    (method.py:352-359, 387-392; experiment/fit.py:366 passes ('alpha', 'beta'))."""
    mod = models.m4b(4, 3, 60)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=80, df0=0.4)
    info = M.run(2, verbose=False, calc_moments=False, save_last_param=('alpha', 'beta'), seed=2)
    assert info == 0
    for k, w in enumerate(M.workers):
        th = M.engine.get_draws(k, all_params=True)
        assert set(w.saved_samp) == {'alpha', 'beta'}
        assert w.saved_samp['alpha'].shape == (160,) and w.saved_samp['beta'].shape == (160, 3)
        np.testing.assert_allclose(w.saved_samp['alpha'], th[:, 0] + th[:, 8] * np.exp(th[:, 1]), rtol=1e-13)
        np.testing.assert_allclose(w.saved_samp['beta'], th[:, 2:5] + th[:, 9:12] * np.exp(th[:, 5:8]), rtol=1e-13)
    w = M.workers[1]
    dQ, dr = np.zeros((8, 8), order='F'), np.zeros(8)
    assert w.tilted(dQ, dr, save_samples=['phi', 'sigma_a'], seed=3)
    np.testing.assert_array_equal(w.saved_samp['phi'], M.engine.get_draws(1))
    with pytest.raises(ValueError, match='not defined'):
        w.cavity(M.Q, M.r, M.Qi[:, :, 1], M.ri[:, 1])
        w.tilted(dQ, dr, save_samples=['gamma'], seed=3)


# ---------------------------------------------------------------------------------------------
# pieced launches of layout 5: a chain stops at a transition boundary, leaves a checkpoint record and goes on
# in another workgroup -- the draws must be those of the uncut run, bit for bit
def _pieced_problem(D, n, it, J=6):
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    return M, M.engine, np.arange(J) + 11


def _run(eng, seeds, opts, J):
    stats, ms = eng.sample_batch(seeds, opts)
    return _all_draws(eng, J), eng.get_chain_stats(4).copy(), stats.copy()


@pytest.mark.parametrize('layout', [5, 7])
@pytest.mark.parametrize('piece_len,rate,D,n', [(7, None, 16, 120), (1, 'skewed', 16, 120), (25, 'skewed', 12, 90),
                                                 (500, None, 16, 120), (13, 'skewed', 32, 500), (26, None, 32, 500)])
def test_piece_queue_gives_the_draws_of_the_uncut_launch(piece_len, rate, D, n, layout):
    """epx_set_piece_queue: one workgroup per piece claims a site by largest remaining predicted work, runs piece_len
    transitions of it from the checkpoint the piece before left (warm-up windows and metric updates on either side of
    a cut: it = 50, windows end at transitions 11 and 22), and puts it back; whatever the claims, the draws, the
    chain and the site statistics are those of one workgroup per site -- also of a warm start from them."""
    it = 50
    M, eng, seeds = _pieced_problem(D, n, it, J=9)
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=layout)
    ref = _run(eng, seeds, opts, 9)
    assert eng.last_layout() == layout
    r = None if rate is None else np.array([9.0, 1.0, 1.0, 5.0, 1.0, 1.0, 2.0, 1.0, 30.0])
    eng.set_piece_queue(piece_len, r)
    got = _run(eng, seeds, opts, 9)
    assert eng.last_segments() == -((it + piece_len - 1) // piece_len)
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)
    warm = HipEngine.sampler_opts(chains=4, iter=it, init='prev', layout=layout)
    w_got = _run(eng, seeds + 1, warm, 9)
    eng.set_piece_queue(0)
    _run(eng, seeds, opts, 9)
    assert eng.last_segments() == 0
    w_ref = _run(eng, seeds + 1, warm, 9)
    np.testing.assert_array_equal(w_ref[0], w_got[0])
    with pytest.raises(_lib.EpxError, match='not positive'):
        eng.set_piece_queue(5, np.zeros(9))


@pytest.mark.parametrize('model,D,n,piece_len', [('m4b', 40, 260, 9), ('m4b', 70, 150, 50), ('m1b', 40, 300, 1),
                                                  ('m4a', 40, 150, 4), ('m1a', 40, 150, 5)])        # (the last two: Gaussian family)
def test_piece_queue_on_the_streaming_layout(model, D, n, piece_len):
    """The same mechanism in k_nuts_stream (rows streamed from HBM, the four chains of a site in lock step): a piece
    ends when every chain of the site has reached the boundary; draws and statistics equal the uncut launch's."""
    it, J = 36, 7
    mod = models.MODELS[model](J, D, n)
    data = mod.simulate_data(rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    eng, seeds = M.engine, np.arange(J) + 5
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random')
    ref = _run(eng, seeds, opts, J)
    assert eng.last_layout() == 3 and eng.last_segments() == 0
    eng.set_piece_queue(piece_len, np.linspace(1.0, 3.0, J))
    got = _run(eng, seeds, opts, J)
    assert eng.last_layout() == 3 and eng.last_segments() == -((it + piece_len - 1) // piece_len)
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)
    warm = HipEngine.sampler_opts(chains=4, iter=it, init='prev')
    w_got = _run(eng, seeds + 1, warm, J)
    eng.set_piece_queue(0)
    _run(eng, seeds, opts, J)
    w_ref = _run(eng, seeds + 1, warm, J)
    np.testing.assert_array_equal(w_ref[0], w_got[0])


def test_ep_with_the_piece_queue_equals_ep_without(monkeypatch):
    """Master at a site size that fills the LDS and more sites than CUs runs its launches from the piece queue: the
    whole EP trajectory equals the one-workgroup-per-site run (the dispatch does not touch a single draw)."""
    J = 300
    monkeypatch.setattr(Master, 'LEAD_FRACTION', 2.0)      # no lead sites: a split launch would take the queue's place
    mod = models.m4b(J, 20, 340)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    out = []
    for queue in (True, False):
        M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=64,
                   df0=models.default_df0(J), sync_sites=False)
        assert M._one_workgroup_per_cu()
        if not queue:
            M.engine.set_piece_queue = lambda *a, **k: None
        info = M.run(3, verbose=False, calc_moments=False, seed=5)
        out.append((M.Q.copy(), M.r.copy(), M.engine.last_segments(), M.engine.last_layout(), info))
    print('EP with / without the piece queue: sum|Q| = %.6f / %.6f' % (np.abs(out[0][0]).sum(), np.abs(out[1][0]).sum()))
    assert out[0][3] == out[1][3] == 7 and out[0][4] == out[1][4] == 0
    assert out[0][2] == -16 and out[1][2] == 0          # iter = 64: 16 pieces of 4 transitions
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize('D,n,layout', [(16, 120, 5), (16, 120, 7), (40, 150, 3)])
def test_piece_queue_with_a_chain_that_fails_at_its_start(D, n, layout):
    """A site whose density is not finite at the initial point fails in its FIRST piece, where its draws and its
    statistics are written; the later pieces of that site find the mark in the checkpoint record and end at once.
    Everything -- the failed site's record included -- equals the uncut launch."""
    it, J = 30, 9
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(rng=100)
    X = data.X.copy()
    X[int(np.cumsum(data.Nj)[3]) + 2, 1] = np.nan          # one row of site 4
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    eng, seeds = M.engine, np.arange(J) + 3
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=layout)
    ref = _run(eng, seeds, opts, J)
    assert eng.last_layout() == layout
    assert np.all(ref[1][4, :, 7] > 0) and np.all(np.delete(ref[1], 4, axis=0)[:, :, 7] == 0)
    eng.set_piece_queue(7, None)
    got = _run(eng, seeds, opts, J)
    assert eng.last_segments() == -5
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)


def test_piece_queue_runs_repeat_bit_for_bit(monkeypatch):
    """The claims of a pieced launch depend on timing; its draws must not.  Eight EP runs from the queue in one
    process (device memory full of the earlier runs' data) give the same global parameters bit for bit -- the check
    that found the two-word site state of the first form (scripts/leak_check.py is the long version)."""
    J = 300
    monkeypatch.setattr(Master, 'LEAD_FRACTION', 2.0)
    mod = models.m4b(J, 20, 340)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    ref = None
    for rep in range(8):
        M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=64,
                   df0=models.default_df0(J), sync_sites=False)
        assert M.run(2, verbose=False, calc_moments=False, seed=5) == 0 and M.engine.last_segments() == -16
        cur = (M.Q.copy(), M.r.copy(), M.engine.get_chain_stats(4).copy())
        if ref is None:
            ref = cur
        for a, b in zip(ref, cur):
            np.testing.assert_array_equal(a, b)
        del M


def test_piece_queue_on_streamed_sites_with_several_groups():
    """K < J on the streaming layout (the sites sample different numbers of coordinates): pieces of 4 and of 7
    transitions give the draws and chain statistics of the uncut launch."""
    from epstan_amd.util import distribute_groups
    J, K, D, n, it = 24, 8, 40, 90, 30
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(rng=100)
    _, _, Q0, r0 = mod.get_prior()
    Nk, Nj_k, j_ind_k = distribute_groups(J, K, data.Nj)
    M = Master('m4b', data.X, data.y, site_sizes=Nk, A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k + 1},
               prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    eng, seeds = M.engine, np.arange(K) + 5
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random')
    out = []
    for q in (0, 4, 7):
        eng.set_piece_queue(q, None)
        eng.sample_batch(seeds, opts)
        out.append(([eng.get_draws(k, all_params=True).copy() for k in range(K)], eng.get_chain_stats(4).copy(),
                    eng.last_layout(), eng.last_segments()))
    assert [o[2] for o in out] == [3, 3, 3] and [o[3] for o in out] == [0, -8, -5]
    for o in out[1:]:
        for a, b in zip(o[0], out[0][0]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(o[1], out[0][1])
