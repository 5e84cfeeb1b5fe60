"""Deterministic stand-in samplers shared by make_golden.py and the tests.

The reference's per-site sampler is PyStan (absent here), so the
golden `Master.run` trajectories were captured with the reference's
`epstan.method._sample_stan` replaced by the closed-form Gaussian sampler
below. The tests feed *the same function* to this repo's `Master` through its
sample-injection test hook, so both sides see identical draws (up to the
1e-15 differences of the cavity parameters they are computed from).

Signature mirrors the data/stan_params dicts the reference hands to Stan
(/root/reference/epstan/method.py:217-227, 154-160, 346): `data` holds
`X, y, N, D, mu_phi, Omega_phi` plus `site_id` (passed through `A_k`),
`stan_params` holds `chains, iter, warmup, thin, init, seed`.
"""

import numpy as np


def n_draws(stan_params):
    """S = chains * (iter - warmup) / thin, warmup=None -> iter // 2."""
    it = stan_params['iter']
    wu = stan_params['warmup']
    if wu is None:
        wu = it // 2
    return stan_params['chains'] * ((it - wu) // stan_params['thin'])


def gaussian_site_natural(X, y, tau=0.05):
    """Pseudo-likelihood natural parameters of one site, a pure function of
    its data: Z = [1, X] (n, D+1), A = tau Z'Z, b = tau Z'(2y-1)."""
    Z = np.concatenate([np.ones((X.shape[0], 1)), X], axis=1)
    A = tau * Z.T.dot(Z)
    b = tau * Z.T.dot(2.0 * np.asarray(y, dtype=np.float64) - 1.0)
    return A, b


class GaussianTilted(object):
    """Exact draws from N(Qt^-1 rt, Qt^-1), Qt = Omega + A_k, rt = Omega mu + b_k.

    scenario:
      'smooth'       plain closed form
      'wide_first'   site 0 returns draws with `factor` times the covariance on
                     its first call (init == 'random'): drives the
                     "Non pos. def. cavity ... reducing df" branch
                     (/root/reference/epstan/method.py:1160-1176)
      'degenerate'   every site returns draws whose first coordinate is constant
                     -> singular scatter -> every site fails (info 4, :1033-1040)
      'wide_all'     every site returns draws with `factor` times the covariance
                     on the first call -> non-pd global Q on iteration 1
                     (info 1, :1092-1101)
    """

    def __init__(self, scenario='smooth', factor=60.0, tau=0.05):
        self.scenario = scenario
        self.factor = factor
        self.tau = tau

    def __call__(self, data, stan_params):
        X = np.asarray(data['X'])
        if X.ndim == 1:
            X = X[:, None]
        Omega = np.array(data['Omega_phi'], dtype=np.float64)
        mu = np.array(data['mu_phi'], dtype=np.float64)
        d = mu.shape[0]
        A, b = gaussian_site_natural(X, data['y'], self.tau)
        A = A[:d, :d]
        b = b[:d]
        Qt = 0.5 * (Omega + Omega.T) + A
        rt = Omega.dot(mu) + b
        St = np.linalg.inv(Qt)
        St = 0.5 * (St + St.T)
        mt = St.dot(rt)
        first = isinstance(stan_params['init'], str)
        site = int(data.get('site_id', -1))
        if self.scenario == 'wide_first' and first and site == 0:
            St = St * self.factor
        if self.scenario == 'wide_all' and first:
            St = St * self.factor
        L = np.linalg.cholesky(St)
        S = n_draws(stan_params)
        rng = np.random.RandomState(stan_params['seed'])
        z = rng.randn(S, d)
        samp = mt + z.dot(L.T)
        if self.scenario == 'degenerate':
            samp[:, 0] = 1.25
        return np.asfortranarray(samp)
