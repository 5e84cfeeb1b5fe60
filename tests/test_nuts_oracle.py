"""The C sampler restatement (oracle/nuts_oracle.c) against answers that do not
depend on it: finite differences and a NumPy re-derivation for the densities,
exact Gaussian moments and 3-D quadrature for the sampler.  CPU only."""

import numpy as np
import pytest

from oracle import ep_oracle as eo
from oracle import nuts_oracle as no

MODELS = ['m1b_sg', 'm2b_sg', 'm3b_sg', 'm4b_sg', 'm5b_sg']


def _problem(model, D, n, seed):
    rng = np.random.RandomState(seed)
    X = rng.randn(n, D)
    y = (rng.rand(n) < 0.5).astype(np.int32)
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
        if model == 'm5b_sg':
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
