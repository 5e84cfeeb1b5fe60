"""The C sampler restatement (oracle/nuts_oracle.c) against answers that do not
depend on it: finite differences and a NumPy re-derivation for the densities,
exact Gaussian moments and 3-D quadrature for the sampler.  CPU only."""

import numpy as np
import pytest

from oracle import ep_oracle as eo
from oracle import nuts_oracle as no

MODELS = ['m1b_sg', 'm2b_sg', 'm3b_sg', 'm4b_sg', 'm5b_sg',
          'm1a_sg', 'm2a_sg', 'm3a_sg', 'm4a_sg', 'm5a_sg']     # a: Gaussian likelihood, real responses


def _problem(model, D, n, seed):
    rng = np.random.RandomState(seed)
    X = rng.randn(n, D)
    y = (rng.rand(n) < 0.5).astype(np.int32)
    if no.is_gauss(model):
        y = rng.randn(n) * 1.3 + 0.2
    d, P = no.dims(model, D)
    A = rng.randn(d, d + 3)
    Om = A.dot(A.T) / (d + 3) + 0.5 * np.eye(d)
    return X, y, rng.randn(d) * 0.5, Om, d, P, rng


@pytest.mark.parametrize('model', MODELS)
def test_logdensity_and_gradient(model):
    X, y, mu, Om, d, P, rng = _problem(model, 5, 40, 1)
    for _ in range(3):
        th = rng.randn(P) * 0.6
        lp, g = no.logdensity_grad(model, X, y, mu, Om, th)
        lp2, g2 = eo.site_logdensity(model, th, X, y, mu, Om)      # independent NumPy restatement
        assert abs(lp - lp2) < 1e-11 * max(1, abs(lp2))
        np.testing.assert_allclose(g, g2, rtol=1e-11, atol=1e-11)
        fd = np.zeros(P)
        for i in range(P):
            e = np.zeros(P); e[i] = 1e-6
            fd[i] = (no.logdensity_grad(model, X, y, mu, Om, th + e)[0]
                     - no.logdensity_grad(model, X, y, mu, Om, th - e)[0]) / 2e-6
        mask = np.ones(P, bool)
        if model in ('m5b_sg', 'm5a_sg'):
            mask = np.abs(th) > 1e-4                                   # |.| is not smooth at 0
        np.testing.assert_allclose(g[mask], fd[mask], rtol=2e-6, atol=2e-6)


def test_sampler_gaussian_known_answer():
    """Without data rows the tilted distribution is the Gaussian cavity times N(0,1)
    nuisance terms: exact moments."""
    model, D = 'm4b_sg', 4
    _, _, mu, Om, d, P, rng = _problem(model, D, 1, 3)
    X0 = np.zeros((0, D)); y0 = np.zeros(0, dtype=np.int32)
    draws, _, stats = no.nuts_sites(model, X0, y0, [0, 0], mu[None], Om[None], [42], chains=4,
                                    iter=6000, warmup=1000)
    x = draws[0].reshape(-1, P)
    S = np.linalg.inv(Om)
    sd = np.sqrt(np.diag(S))
    assert np.all(np.abs(x[:, :d].mean(0) - mu) < 5 * sd / np.sqrt(5000))
    np.testing.assert_allclose(np.cov(x[:, :d].T), S, atol=0.06 * np.abs(S).max())
    np.testing.assert_allclose(x[:, d:].var(0), 1.0, atol=0.08)
    assert 0.75 < stats[0, :, 5].mean() < 0.97 and stats[0, :, 4].sum() == 0


def test_sampler_gaussian_family_known_answer():
    """m1a_sg with log sigma and log sigma_a pinned by the cavity: the tilted distribution of
    (beta, eta) is Gaussian with a closed-form mean and covariance (the analytic check of the
    Gaussian-likelihood family)."""
    rng = np.random.RandomState(12)
    D, n = 3, 25
    d, P = no.dims('m1a_sg', D)                 # phi = [log sigma, log sigma_a, beta(3)], eta
    X = rng.randn(n, D)
    ls, lsa = np.log(0.7), np.log(1.4)
    y = 0.3 + X.dot([0.5, -1.0, 0.2]) + 0.7 * rng.randn(n)
    A = rng.randn(D, D + 2)
    Ob = A.dot(A.T) / (D + 2) + 0.4 * np.eye(D)
    Om = np.zeros((d, d)); Om[0, 0] = Om[1, 1] = 1e8; Om[2:, 2:] = Ob
    mu = np.concatenate(([ls, lsa], rng.randn(D) * 0.3))
    draws, _, stats = no.nuts_sites('m1a_sg', X, y, [0, n], mu[None], Om[None], [7], chains=4,
                                    iter=5000, warmup=1000)
    x = draws[0].reshape(-1, P)
    assert np.abs(x[:, 0] - ls).max() < 1e-3 and np.abs(x[:, 1] - lsa).max() < 1e-3
    # exact: z = (beta, eta), f = [X | sigma_a 1] z, prior precision diag(Ob, 1), noise sigma
    Z = np.hstack((X, np.full((n, 1), np.exp(lsa))))
    Pz = np.zeros((D + 1, D + 1)); Pz[:D, :D] = Ob; Pz[D, D] = 1.0
    prec = Pz + Z.T.dot(Z) / np.exp(2 * ls)
    rhs = np.concatenate((Ob.dot(mu[2:]), [0.0])) + Z.T.dot(y) / np.exp(2 * ls)
    S = np.linalg.inv(prec); m = S.dot(rhs)
    z = x[:, 2:]
    sd = np.sqrt(np.diag(S))
    assert np.all(np.abs(z.mean(0) - m) < 5 * sd / np.sqrt(4000))
    np.testing.assert_allclose(np.cov(z.T), S, atol=0.08 * np.abs(S).max())
    assert stats[0, :, 4].sum() == 0


def test_sampler_logistic_against_quadrature():
    """m1b_sg with one covariate: theta = (log sigma_a, beta, eta); posterior moments by
    brute-force quadrature on a 3-D grid vs a long NUTS run (4 sigma of the MCSE)."""
    model = 'm1b_sg'
    rng = np.random.RandomState(11)
    n = 12
    X = rng.randn(n, 1) * 1.3
    y = (rng.rand(n) < 0.6).astype(np.int32)
    mu = np.array([0.2, -0.3])
    Om = np.array([[1.4, 0.3], [0.3, 0.9]])
    g = np.linspace(-6, 6, 161)
    A, B, E = np.meshgrid(g + mu[0], g + mu[1], g, indexing='ij')
    v0, v1 = A - mu[0], B - mu[1]
    lp = -0.5 * (Om[0, 0] * v0 * v0 + 2 * Om[0, 1] * v0 * v1 + Om[1, 1] * v1 * v1) - 0.5 * E * E
    alpha = E * np.exp(A)
    for i in range(n):
        f = alpha + X[i, 0] * B
        lp += y[i] * f - np.logaddexp(0.0, f)
    w = np.exp(lp - lp.max())
    w /= w.sum()
    ref_mean = np.array([(w * A).sum(), (w * B).sum(), (w * E).sum()])
    ref_var = np.array([(w * A * A).sum(), (w * B * B).sum(), (w * E * E).sum()]) - ref_mean**2
    draws, _, stats = no.nuts_sites(model, X, y, [0, n], mu[None], Om[None], [7], chains=4,
                                    iter=12000, warmup=2000)
    x = draws[0].reshape(-1, 3)
    rhat = max(no.split_rhat(draws[0, :, :, e]) for e in range(3))
    assert rhat < 1.01
    ess = 4000.0                                   # conservative for 40000 NUTS draws
    assert np.all(np.abs(x.mean(0) - ref_mean) < 4 * np.sqrt(ref_var / ess)), (x.mean(0), ref_mean)
    assert np.all(np.abs(x.var(0) / ref_var - 1) < 4 * np.sqrt(2 / ess) * 1.5), (x.var(0), ref_var)


def test_rng_is_counter_based_and_reproducible():
    a = no.rng_probe(123, 1, 5, 4, 7, 2)
    assert a == no.rng_probe(123, 1, 5, 4, 7, 2)
    assert a != no.rng_probe(123, 2, 5, 4, 7, 2)
    u = np.array([no.rng_probe(9, 0, t, 1, 0, 0)[0] for t in range(2000)])
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.03
    z = np.array([no.rng_probe(9, 0, t, 1, 0, 0)[2:] for t in range(2000)]).ravel()
    assert abs(z.mean()) < 0.06 and abs(z.std() - 1) < 0.05


def test_thin_and_warm_start_shapes():
    X, y, mu, Om, d, P, rng = _problem('m1b_sg', 3, 20, 5)
    dr, last, st = no.nuts_sites('m1b_sg', X, y, [0, 20], mu[None], Om[None], [1], chains=2, iter=40,
                                 warmup=10, thin=3)
    assert dr.shape == (1, 2, 10, P)
    np.testing.assert_array_equal(last[0], dr[0, :, -1] if (40 - 10 - 1) % 3 == 0 else last[0])
    dr2, _, _ = no.nuts_sites('m1b_sg', X, y, [0, 20], mu[None], Om[None], [1], chains=2, iter=40,
                              warmup=10, thin=3)
    np.testing.assert_array_equal(dr, dr2)


# ---------------------------------------------------------------- multi-group sites (K < J)
@pytest.mark.parametrize('model', ['m1b', 'm2b', 'm3b', 'm4b', 'm5b'])
@pytest.mark.parametrize('D,sizes', [(3, [5, 1, 9]), (6, [20, 13]), (4, [7])])
def test_multigroup_density_matches_stan_program_and_finite_differences(model, D, sizes):
    rng = np.random.RandomState(D + len(sizes))
    n = int(np.sum(sizes)); ng = len(sizes)
    X = rng.randn(n, D); y = (rng.rand(n) < 0.5).astype(int)
    gl = np.concatenate(([0], np.cumsum(sizes)))
    j_ind = np.repeat(np.arange(ng), sizes)
    d, P = no.dims(model, D, ng)
    assert P == d + ng * (1 if model == 'm1b' else 1 + D)
    A = rng.randn(d, d + 2); Om = A.dot(A.T) / d + np.eye(d); mu = rng.randn(d) * 0.3
    th = rng.randn(P) * 0.4
    lp, g = no.logdensity_grad(model, X, y, mu, Om, th, gl=gl)
    assert abs(lp - eo.site_logdensity_groups(model, th, X, y, j_ind, mu, Om)) < 1e-10 * max(1.0, abs(lp))
    h = 1e-6
    for e in range(P):
        tp, tm = th.copy(), th.copy()
        tp[e] += h; tm[e] -= h
        fd = (eo.site_logdensity_groups(model, tp, X, y, j_ind, mu, Om)
              - eo.site_logdensity_groups(model, tm, X, y, j_ind, mu, Om)) / (2 * h)
        if model == 'm5b' and e >= d and abs(th[e]) < 2 * h:
            continue                                    # |x| is not differentiable at 0
        assert abs(fd - g[e]) < 2e-6 * max(1.0, abs(g[e])), (e, fd, g[e])
    if ng == 1:                                         # one group = the `_sg` program
        lp1, g1 = no.logdensity_grad(model + '_sg', X, y, mu, Om, th)
        assert lp1 == lp and np.array_equal(g1, g)


def test_multigroup_sampler_against_importance_sampling():
    """Moments of the multi-group sampler against self-normalised importance sampling of the
    independent NumPy density (a Gaussian proposal fitted to the draws and widened)."""
    rng = np.random.RandomState(5)
    D, sizes = 2, [6, 3, 8]
    n = int(np.sum(sizes))
    X = rng.randn(n, D); y = (rng.rand(n) < 0.5).astype(int)
    k_lim = np.array([0, n]); g_cnt = np.array([3]); g_lim = np.concatenate(([0], np.cumsum(sizes)))
    j_ind = np.repeat(np.arange(3), sizes)
    d, P = no.dims('m4b', D, 3)
    Om = np.diag(rng.uniform(3.0, 8.0, size=d)); mu = rng.randn(d) * 0.2
    draws, last, stats = no.nuts_sites('m4b', X, y, k_lim, mu[None], Om[None], np.array([3]), chains=4, iter=3000,
                                       g_cnt=g_cnt, g_lim=g_lim)
    assert stats[0, :, 7].sum() == 0 and stats[0, :, 4].sum() < 30
    samp = draws[0].reshape(-1, P)
    m, C = samp.mean(0), np.cov(samp.T)
    L = np.linalg.cholesky(C * 1.6)
    z = rng.randn(30000, P)
    prop = m + z.dot(L.T)
    logq = -0.5 * np.sum(z * z, axis=1)
    logp = np.array([eo.site_logdensity_groups('m4b', t, X, y, j_ind, mu, Om) for t in prop])
    w = np.exp(logp - logq - np.max(logp - logq)); w /= w.sum()
    ess = 1.0 / np.sum(w * w)
    assert ess > 2000, ess
    m_is = w.dot(prop)
    sd = np.sqrt(np.diag(C))
    assert np.all(np.abs(m_is - m) < 0.08 * sd), np.abs(m_is - m) / sd
    v_is = w.dot((prop - m_is)**2)
    assert np.all(np.abs(v_is / np.diag(C) - 1.0) < 0.15), v_is / np.diag(C)
    # one group through the groups entry point = the `_sg` program, bit for bit
    d1 = no.nuts_sites('m4b_sg', X, y, k_lim, mu[None], Om[None], np.array([3]), chains=2, iter=60)[0]
    d2 = no.nuts_sites('m4b', X, y, k_lim, mu[None], Om[None], np.array([3]), chains=2, iter=60,
                       g_cnt=np.array([1]), g_lim=k_lim)[0]
    assert np.array_equal(d1, d2)


@pytest.mark.parametrize('model', ['m1a', 'm2a', 'm3a', 'm4a', 'm5a'])
def test_gaussian_family_multigroup_density(model):
    """Several groups per site with the Gaussian likelihood (models/m1a.stan etc.): the oracle's value is the
    sum of the single-group values minus the cavity term counted once per extra group, and its gradient
    matches finite differences."""
    rng = np.random.RandomState(21)
    D, sizes = 4, [7, 12, 5]
    n, ng = sum(sizes), len(sizes)
    X = rng.randn(n, D); y = rng.randn(n) * 1.1 + 0.3
    d, P = no.dims(model, D, ng)
    d1, P1 = no.dims(model + '_sg', D)
    assert d == d1 and P == d + ng * (P1 - d)
    A = rng.randn(d, d + 2)
    Om = A.dot(A.T) / (d + 2) + 0.4 * np.eye(d); mu = 0.3 * rng.randn(d)
    th = rng.randn(P) * 0.4
    gl = np.concatenate(([0], np.cumsum(sizes)))
    lp, g = no.logdensity_grad(model, X, y, mu, Om, th, gl=gl)
    v = th[:d] - mu
    cav = -0.5 * v.dot(Om).dot(v)
    pg = P1 - d
    total = 0.0
    for j in range(ng):
        eta = th[d + j:d + j + 1]
        etb = th[d + ng + j * D:d + ng + (j + 1) * D] if pg > 1 else np.zeros(0)
        th_j = np.concatenate((th[:d], eta, etb))
        total += no.logdensity_grad(model + '_sg', X[gl[j]:gl[j + 1]], y[gl[j]:gl[j + 1]], mu, Om, th_j)[0] - cav
    assert abs(lp - (total + cav)) < 1e-10 * max(1.0, abs(lp))
    fd = np.zeros(P)
    for i in range(P):
        e = np.zeros(P); e[i] = 1e-6
        fd[i] = (no.logdensity_grad(model, X, y, mu, Om, th + e, gl=gl)[0] - no.logdensity_grad(model, X, y, mu, Om, th - e, gl=gl)[0]) / 2e-6
    mask = np.abs(th) > 1e-4 if model == 'm5a' else np.ones(P, bool)
    np.testing.assert_allclose(g[mask], fd[mask], rtol=2e-6, atol=2e-6)


def _overflowing_cavity(d):
    """A cavity whose log density and gradient are finite only for |phi_0| < 1: about half of the U(-2, 2) starts fail."""
    Om = np.eye(d)
    Om[0, 0] = np.finfo(float).max
    return Om, np.zeros(d)


def test_random_init_is_drawn_again_until_it_is_finite():
    """PyStan's init='random' (util.py:716 -> stan::services::util::initialize): a start whose log density or gradient is
    not finite is drawn again, up to 100 times.  With a cavity that is finite only for |phi_0| < 1 the chains whose first
    draw lies outside still run, from a later draw of their Philox stream; a given start gets one attempt."""
    model, D, n, K = 'm1b_sg', 3, 20, 4
    rng = np.random.RandomState(3)
    X = rng.randn(K * n, D); y = (rng.rand(K * n) < 0.5).astype(int)
    d, P = no.dims(model, D)
    Om, mu = _overflowing_cavity(d)
    seeds = np.arange(K, dtype=np.int64) + 40
    draws, last, st = no.nuts_sites(model, X, y, np.arange(K + 1) * n, np.tile(mu, (K, 1)), np.tile(Om, (K, 1, 1)), seeds,
                                    chains=4, iter=4, warmup=2)
    first = np.array([[-2.0 + 4.0 * no.rng_probe(int(s), c, 0, 0, 0, 0)[0] for c in range(4)] for s in seeds])
    retried = np.abs(first) >= 1.0
    assert retried.any() and (~retried).any()
    assert np.all(st[:, :, 7] == 0)                                  # nobody failed
    assert np.all(np.abs(draws[:, :, :, 0]) < 1.0)                   # every chain sits at a finite start
    # the chains whose first draw was fine start from it; the others from a later one
    np.testing.assert_allclose(draws[~retried][:, 0, 0], first[~retried], atol=1e-100)
    assert np.all(np.abs(draws[retried][:, 0, 0] - first[retried]) > 1e-6)
    # a given start is not replaced: the chain fails
    bad = np.zeros((K, 4, P)); bad[:, :, 0] = 1.5
    _, _, st2 = no.nuts_sites(model, X, y, np.arange(K + 1) * n, np.tile(mu, (K, 1)), np.tile(Om, (K, 1, 1)), seeds,
                              chains=4, iter=4, warmup=2, init=bad)
    assert np.all(st2[:, :, 7] == 1)


def test_timing_build_agrees_with_the_strict_build():
    """bench.py's cpu_baseline times a second build of the same source (oracle/Makefile FAST_LIB: -O3 -march=native,
    contraction allowed, vectorised logistic terms); the strict build stays the checker.  The two agree on the log density
    and its gradient at 1e-9, their traces of a site update agree transition by transition until a decision parts them,
    and the tilted moments of a whole site update agree statistically."""
    model, D, n = 'm4b_sg', 12, 150
    X, y, mu, Om, d, P, rng = _problem(model, D, n, 11)
    for _ in range(4):
        th = rng.randn(P) * 0.7
        lp, g = no.logdensity_grad(model, X, y, mu, Om, th)
        with no.timing_build():
            lp2, g2 = no.logdensity_grad(model, X, y, mu, Om, th)
        assert abs(lp - lp2) <= 1e-9 * max(1.0, abs(lp))
        np.testing.assert_allclose(g2, g, rtol=1e-9, atol=1e-9)
    args = (model, X, y, [0, n], (0.3 * mu)[None], (Om * 50.0)[None], [5])
    d1, _, s1, t1 = no.nuts_sites(*args, chains=4, iter=120, trace_sites=1)
    with no.timing_build():
        d2, _, s2, t2 = no.nuts_sites(*args, chains=4, iter=120, trace_sites=1)
    # transition by transition until a decision parts them (a dominant cavity: most chains stay together to the end)
    err = np.abs(t1[0, :, :, 8:] - t2[0, :, :, 8:]).max(axis=2)
    part = (err > 1e-6) | (t1[0, :, :, 1] != t2[0, :, :, 1])
    first = np.where(part.any(axis=1), part.argmax(axis=1), 120)
    assert np.median(first) >= 30, first
    for c in range(4):
        assert first[c] == 0 or err[c, :first[c]].max() < 1e-6
    # ... and statistically as a whole (mean within 5 MCSE-ish, sd within 35 %)
    a, b = d1[0].reshape(-1, P)[:, :d], d2[0].reshape(-1, P)[:, :d]
    sd = 0.5 * (a.std(axis=0) + b.std(axis=0))
    assert np.all(np.abs(a.mean(axis=0) - b.mean(axis=0)) < 5 * sd / np.sqrt(30.0))
    assert np.all(np.abs(a.std(axis=0) / b.std(axis=0) - 1.0) < 0.35)
