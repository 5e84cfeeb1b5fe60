"""Synthetic inputs of the BASELINE configs: hierarchical logistic regression.

Host-side restatement of the reference's simulators for the logistic family
(/root/reference/experiment/models/m1b.py:77-213, m4b.py:88-218,
common.py:33-78, 132-317) and of `fit.py`'s default damping schedule
(fit.py:171-186).  Pure NumPy input generation -- no numerical work of the EP
path happens here.  The generators consume `numpy.random.RandomState` in exactly
the reference's order, so `seed_data=100` reproduces the reference's data
(checked against tests/golden/simulators.npz).
"""

import numpy as np
from scipy.linalg import cholesky
from scipy.special import erfinv, logit

# common.py:21-30
P_0 = 0.2
GAMMA_0 = 0.01
SIGMA_F0 = 0.25
ERFINVGAMMA0 = erfinv(2*GAMMA_0 - 1)
LOGITP0 = logit(P_0)
DELTA_MAX = np.sqrt(2)*SIGMA_F0*ERFINVGAMMA0 - LOGITP0
B_ABS_MIN_SUM = 1e-4


def rand_corr_vine(d, alpha=2, beta=2, pmin=-0.8, pmax=0.8, seed=None):
    """Random correlation matrix by the C-vine construction (the reference's common.py:33-78).

    Partial correlations rho[k, i] (k < i), Beta(alpha, beta) draws stretched to [pmin, pmax], are
    turned into correlations by conditioning out the variables k = i-1, ..., 0 one after the other;
    a random permutation of the variables follows.  Random numbers are consumed and combined in the
    reference's order, so a given seed gives the reference's matrix."""
    rs = seed if isinstance(seed, np.random.RandomState) else np.random.RandomState(seed)
    upper = np.triu_indices(d, 1)
    draws = rs.beta(alpha, beta, size=upper[0].shape[0])
    draws *= pmax - pmin
    draws += pmin
    rho = np.zeros((d, d))                     # rho[k, i], k < i: partial correlation of i and k given 0..k-1
    rho[upper] = draws
    rho_sq = np.square(rho)
    C = np.eye(d)
    for i, j in zip(*upper):
        corr = rho[i, j]
        for k in reversed(range(i)):
            corr = corr * np.sqrt((1 - rho_sq[k, i]) * (1 - rho_sq[k, j])) + rho[k, i] * rho[k, j]
        C[i, j] = C[j, i] = corr
    order = rs.permutation(d)
    return C[order][:, order]


def calc_input_param_classification(alpha, beta, Sigma_x=None):
    """Input location and scale (mu_x, sigma_x) per group such that the class probabilities stay
    away from 0 and 1: alpha (J,), beta (D,) or (J, D), D > 1 (the multi-group branches of the
    reference's common.py:260-315).  With sd(x beta) = sigma_x * s, s^2 = beta' Sigma_x beta:
    a group whose intercept is small (|alpha| < DELTA_MAX) keeps mu_x = 0 and takes the largest
    scale that leaves P(f beyond logit(P_0)) = GAMMA_0; otherwise the inputs are shifted so that the
    mean of f sits at +-DELTA_MAX and the scale gives sd(f) = SIGMA_F0."""
    alpha = np.asarray(alpha, dtype=np.float64)
    beta = np.asarray(beta, dtype=np.float64)
    J = alpha.shape[0]
    total = np.broadcast_to(np.sum(beta, axis=-1), (J,))
    quad = np.square(beta) if Sigma_x is None else beta.dot(Sigma_x) * beta
    s = np.broadcast_to(np.sqrt(np.sum(quad, axis=-1)), (J,))
    small = np.abs(alpha) < DELTA_MAX
    target = np.where(alpha > 0, DELTA_MAX, -DELTA_MAX)
    mu_x = np.where(small, 0.0, (target - alpha) / total)
    sigma_x = np.where(small, (LOGITP0 + np.abs(alpha)) / (np.sqrt(2) * ERFINVGAMMA0 * s), SIGMA_F0 / s)
    return mu_x, sigma_x


R_SQUARED = 0.5            # common.py:16: target coefficient of determination of the regression models


def calc_input_param_lin_reg(beta, sigma, Sigma_x=None):
    """Input scale of the linear-regression simulators (common.py:81-131): sigma_x such that
    Var(x beta) / (Var(x beta) + sigma^2) = R_SQUARED, per group when beta is (J, D)."""
    beta = np.asarray(beta, dtype=np.float64)
    one_dim = beta.ndim == 0 or beta.shape[-1] == 1
    if Sigma_x is not None and one_dim:
        raise ValueError("Input dimension has to be greater than 1 if Sigma is provided")
    if one_dim:
        b = np.abs(beta.reshape(-1)) if beta.ndim == 2 else np.abs(beta.reshape(-1)[0])
        return np.sqrt(R_SQUARED/(1 - R_SQUARED))*sigma/b
    quad = np.sum(np.square(beta), axis=-1) if Sigma_x is None else np.sum(beta.dot(Sigma_x)*beta, axis=-1)
    return np.sqrt(R_SQUARED/(quad*(1 - R_SQUARED)))*sigma


class Data(object):
    """Simulated data set (common.py:320-404, the fields the EP path uses)."""

    def __init__(self, X, y, Nj, j_lim, phi_true, X_param):
        self.X, self.y, self.Nj, self.j_lim = X, y, Nj, j_lim
        self.N = int(np.sum(Nj))
        self.J = Nj.shape[0]
        self.phi_true = phi_true
        self.X_param = X_param


def _sizes(rng, J, npg):
    if hasattr(npg, '__getitem__') and len(npg) == 2:
        Nj = rng.randint(npg[0], npg[1] + 1, size=J)
    else:
        Nj = npg*np.ones(J, dtype=np.int64)
    return Nj, np.concatenate(([0], np.cumsum(Nj)))


def _draw_X(rng, Nj, j_lim, D, mu_x_j, sigma_x_j, Sigma_x):
    X = np.empty((int(np.sum(Nj)), D))
    if Sigma_x is None:
        for j in range(len(Nj)):
            X[j_lim[j]:j_lim[j+1], :] = mu_x_j[j] + rng.randn(Nj[j], D)*sigma_x_j[j]
    else:
        cho_x = cholesky(Sigma_x)
        for j in range(len(Nj)):
            X[j_lim[j]:j_lim[j+1], :] = mu_x_j[j] + rng.randn(Nj[j], D).dot(sigma_x_j[j]*cho_x)
    return X


def _bernoulli(rng, f):
    return (rng.rand(f.shape[0]) < 1/(1 + np.exp(-f))).astype(int)


def _regulate_rows(rng, beta_j, redraw):
    """Keep |sum(beta_j)| away from zero, group by group (e.g. m4b.py:150-158)."""
    for j in range(beta_j.shape[0]):
        beta_sum = np.sum(beta_j[j])
        while np.abs(beta_sum) < B_ABS_MIN_SUM:
            index = rng.randint(beta_j.shape[1])
            beta_sum -= beta_j[j, index]
            beta_j[j, index] = redraw(index)
            beta_sum += beta_j[j, index]


class _LogisticBase(object):
    """Shared skeleton of the remaining logistic simulators: the model-specific part draws the
    parameters (in the reference's order of random draws) and returns (alpha_j, beta or beta_j,
    phi_true)."""

    def __init__(self, J, D, npg):
        self.J, self.D, self.npg = J, D, npg
        self.dphi = self._dphi(D)

    def simulate_data(self, Sigma_x=None, rng=None):
        J, D = self.J, self.D
        if not isinstance(rng, np.random.RandomState):
            rng = np.random.RandomState(rng)
        seed_input_cov = rng.randint(2**31 - 1)
        if isinstance(Sigma_x, str) and Sigma_x == 'rand':
            Sigma_x = rand_corr_vine(D, seed=seed_input_cov)
        Nj, j_lim = _sizes(rng, J, self.npg)
        alpha_j, beta, phi_true = self._draw_parameters(rng)
        mu_x_j, sigma_x_j = calc_input_param_classification(alpha_j, beta, Sigma_x)
        X = _draw_X(rng, Nj, j_lim, D, mu_x_j, sigma_x_j, Sigma_x)
        j_ind = np.repeat(np.arange(J), Nj)
        f = alpha_j[j_ind] + (X.dot(beta) if beta.ndim == 1 else np.einsum('nd,nd->n', X, beta[j_ind]))
        return Data(X, _bernoulli(rng, f), Nj, j_lim, phi_true,
                    {'mu_x': mu_x_j, 'sigma_x': sigma_x_j, 'Sigma_x': Sigma_x})

    def get_prior(self):
        v = self._prior_var()
        m0 = np.zeros(self.dphi)
        return np.diag(v).T, m0, np.diag(1/v).T, m0/v


class m1b(_LogisticBase):
    """y ~ bernoulli_logit(alpha_j + x beta), alpha_j ~ N(0, sigma_a), beta shared by the groups,
    phi = [log sigma_a, beta] (models/m1b.py; density m1b_sg.stan)."""
    SIGMA_A = 1            # m1b.py:36-40
    SIGMA_AH = None
    SIGMA_B = 1.0
    M0_A, V0_A, M0_B, V0_B = 0, 1.5**2, 0, 1.5**2     # m1b.py:46-50
    site_model = 'm1b_sg'

    def _dphi(self, D):
        return D + 1

    def _draw_parameters(self, rng):
        beta = rng.randn(self.D)*self.SIGMA_B
        _regulate_rows(rng, beta[None, :], lambda index: rng.randn()*self.SIGMA_B)
        alpha_j = rng.randn(self.J)*self.SIGMA_A
        return alpha_j, beta, np.append(np.log(self.SIGMA_A), beta)

    def get_prior(self):
        D = self.D
        var = np.append(self.V0_A, np.full(D, self.V0_B))
        mean = np.append(self.M0_A, np.full(D, self.M0_B))
        return np.diag(var).T, mean, np.diag(1./var).T, np.append(self.M0_A/self.V0_A, np.ones(D)*(self.M0_B/self.V0_B))


class m4b(_LogisticBase):
    """y ~ bernoulli_logit(alpha_j + x beta_j), alpha_j ~ N(mu_a, sigma_a),
    beta_jd ~ N(mu_b_d, sigma_b_d), phi = [mu_a, log sigma_a, mu_b, log sigma_b]
    (models/m4b.py; the paper's model)."""
    MU_A = 1.5                      # m4b.py:44-47
    MU_B = (-2.0, 2.0)
    LOG_SIGMA_A = 0.4
    LOG_SIGMA_B = (-0.5, 0.5)
    V0_MA, V0_SA, V0_MB, V0_SB = 4**2, 2**2, 4**2, 2**2      # m4b.py:50-61
    site_model = 'm4b_sg'

    def _dphi(self, D):
        return 2*D + 2

    def _prior_var(self):
        D = self.D
        return np.concatenate(([self.V0_MA, self.V0_SA], np.full(D, self.V0_MB), np.full(D, self.V0_SB))).astype(np.float64)

    def _draw_parameters(self, rng):
        J, D = self.J, self.D
        lo, hi = self.MU_B
        mu_b = rng.rand(D)*(hi - lo) + lo
        lo, hi = self.LOG_SIGMA_B
        sigma_a = np.exp(self.LOG_SIGMA_A)
        sigma_b = np.exp(rng.rand(D)*(hi - lo) + lo)
        alpha_j = self.MU_A + rng.randn(J)*sigma_a
        beta_j = mu_b + rng.randn(J, D)*sigma_b
        _regulate_rows(rng, beta_j, lambda index: mu_b[index] + rng.randn()*sigma_b[index])
        phi_true = np.concatenate(([self.MU_A, np.log(sigma_a)], mu_b, np.log(sigma_b)))
        return alpha_j, beta_j, phi_true


class m2b(_LogisticBase):
    """alpha_j ~ N(0, sigma_a), beta ~ N(0, sigma_b) shared by the groups, phi = [log sigma_a, log sigma_b]
    (models/m2b.py; density m2b_sg.stan)."""
    SIGMA_A, SIGMA_B = 1, 1                            # m2b.py:38-42
    site_model = 'm2b_sg'

    def _dphi(self, D):
        return 2

    def _prior_var(self):
        return np.array([1.5**2, 1.5**2])             # m2b.py:46-50

    def _draw_parameters(self, rng):
        sigma_a, sigma_b = self.SIGMA_A, self.SIGMA_B
        alpha_j = rng.randn(self.J)*sigma_a
        beta = rng.randn(self.D)*sigma_b
        _regulate_rows(rng, beta[None, :], lambda index: rng.randn()*sigma_b)
        return alpha_j, beta, np.append(np.log(sigma_a), np.log(sigma_b))


class m3b(_LogisticBase):
    """alpha_j ~ N(0, sigma_a), beta_jd ~ N(0, sigma_b_d), phi = [log sigma_a, log sigma_b]
    (models/m3b.py; density m3b_sg.stan)."""
    SIGMA_A, SIGMA_BH = 1, 1                           # m3b.py:39-42
    site_model = 'm3b_sg'

    def _dphi(self, D):
        return D + 1

    def _prior_var(self):
        return np.full(self.dphi, 1.5**2)              # m3b.py:45-50

    def _draw_parameters(self, rng):
        sigma_a = self.SIGMA_A
        sigma_b = np.exp(rng.randn(self.D)*self.SIGMA_BH)
        alpha_j = rng.randn(self.J)*sigma_a
        beta_j = rng.randn(self.J, self.D)*sigma_b
        _regulate_rows(rng, beta_j, lambda index: rng.randn()*sigma_b[index])
        return alpha_j, beta_j, np.append(np.log(sigma_a), np.log(sigma_b))


class m5b(_LogisticBase):
    """m4b with Laplace group effects and half-Cauchy scales, phi = [mu_a, log sigma_a, mu_b, log sigma_b]
    (models/m5b.py; density m5b_sg.stan)."""
    MU_A, SIGMA_A, SIGMA_MB, SIGMA_SB = 0.1, 1, 0, 1   # m5b.py:43-50
    site_model = 'm5b_sg'

    def _dphi(self, D):
        return 2*D + 2

    def _prior_var(self):
        return np.full(self.dphi, 1.5**2)              # m5b.py:53-64

    def _draw_parameters(self, rng):
        J, D = self.J, self.D
        sigma_a, mu_a = self.SIGMA_A, self.MU_A
        sigma_b = np.abs(rng.standard_cauchy(D)*self.SIGMA_SB)
        mu_b = rng.laplace(size=D)*self.SIGMA_MB
        alpha_j = mu_a + rng.laplace(size=J)*sigma_a
        beta_j = mu_b + rng.laplace(size=(J, D))*sigma_b
        _regulate_rows(rng, beta_j, lambda index: mu_b[index] + rng.randn()*sigma_b[index])
        phi_true = np.concatenate(([mu_a, np.log(sigma_a)], mu_b, np.log(sigma_b)))
        return alpha_j, beta_j, phi_true


class m1a(object):
    """y ~ N(alpha_j + x beta, sigma), alpha_j ~ N(0, sigma_a), phi = [log sigma, log sigma_a, beta]
    (models/m1a.py; site density models/m1a_sg.stan)."""
    SIGMA, SIGMA_A, SIGMA_B = 1, 1, 1                  # m1a.py:38-46
    V0_S = V0_A = V0_B = 1.5**2                        # m1a.py:49-58 (zero prior means)
    site_model = 'm1a_sg'

    def __init__(self, J, D, npg):
        self.J, self.D, self.npg = J, D, npg
        self.dphi = D + 2

    def simulate_data(self, Sigma_x=None, rng=None):
        J, D = self.J, self.D
        if not isinstance(rng, np.random.RandomState):
            rng = np.random.RandomState(rng)
        seed_input_cov = rng.randint(2**31 - 1)
        if isinstance(Sigma_x, str) and Sigma_x == 'rand':
            Sigma_x = rand_corr_vine(D, seed=seed_input_cov)
        Nj, j_lim = _sizes(rng, J, self.npg)
        N = int(np.sum(Nj))
        sigma, sigma_a = self.SIGMA, self.SIGMA_A
        beta = rng.randn(D)*self.SIGMA_B
        beta_sum = np.sum(beta)
        while np.abs(beta_sum) < B_ABS_MIN_SUM:
            index = rng.randint(D)
            beta_sum -= beta[index]
            beta[index] = rng.randn()*self.SIGMA_B
            beta_sum += beta[index]
        alpha_j = rng.randn(J)*sigma_a
        phi_true = np.concatenate(([np.log(sigma), np.log(sigma_a)], beta))
        sigma_x = calc_input_param_lin_reg(beta, sigma, Sigma_x)
        if Sigma_x is None:
            X = rng.randn(N, D)*sigma_x
        else:
            X = rng.randn(N, D).dot(sigma_x*cholesky(Sigma_x))
        y = alpha_j[np.repeat(np.arange(J), Nj)] + X.dot(beta)
        y = y + rng.randn(N)*sigma
        return Data(X, y, Nj, j_lim, phi_true, {'sigma_x': sigma_x, 'Sigma_x': Sigma_x})

    def get_prior(self):
        v = np.concatenate(([self.V0_S, self.V0_A], np.full(self.D, self.V0_B)))
        m0 = np.zeros(self.dphi)
        return np.diag(v).T, m0, np.diag(1/v).T, m0/v


class m4a(object):
    """y ~ N(alpha_j + x beta_j, sigma), alpha_j ~ N(mu_a, sigma_a), beta_jd ~ N(mu_b_d, sigma_b_d),
    phi = [log sigma, mu_a, log sigma_a, mu_b, log sigma_b] (models/m4a.py; density m4a_sg.stan)."""
    SIGMA, MU_A, SIGMA_A, SIGMA_MB, SIGMA_SB = 1, 0.1, 1, 1, 1        # m4a.py:45-55
    V0 = 1.5**2                                                       # m4a.py:57-72, all five blocks
    site_model = 'm4a_sg'

    def __init__(self, J, D, npg):
        self.J, self.D, self.npg = J, D, npg
        self.dphi = 2*D + 3

    def simulate_data(self, Sigma_x=None, rng=None):
        J, D = self.J, self.D
        if not isinstance(rng, np.random.RandomState):
            rng = np.random.RandomState(rng)
        seed_input_cov = rng.randint(2**31 - 1)
        if isinstance(Sigma_x, str) and Sigma_x == 'rand':
            Sigma_x = rand_corr_vine(D, seed=seed_input_cov)
        Nj, j_lim = _sizes(rng, J, self.npg)
        sigma, sigma_a, mu_a = self.SIGMA, self.SIGMA_A, self.MU_A
        sigma_b = np.exp(rng.randn(D)*self.SIGMA_SB)
        mu_b = rng.randn(D)*self.SIGMA_MB
        alpha_j = mu_a + rng.randn(J)*sigma_a
        beta_j = mu_b + rng.randn(J, D)*sigma_b
        for j in range(J):
            beta_sum = np.sum(beta_j[j])
            while np.abs(beta_sum) < B_ABS_MIN_SUM:
                index = rng.randint(D)
                beta_sum -= beta_j[j, index]
                beta_j[j, index] = mu_b[index] + rng.randn()*sigma_b[index]
                beta_sum += beta_j[j, index]
        phi_true = np.concatenate(([np.log(sigma), mu_a, np.log(sigma_a)], mu_b, np.log(sigma_b)))
        sigma_x_j = calc_input_param_lin_reg(beta_j, sigma, Sigma_x)
        X = _draw_X(rng, Nj, j_lim, D, np.zeros(J), sigma_x_j, Sigma_x)
        j_ind = np.repeat(np.arange(J), Nj)
        y = alpha_j[j_ind] + np.einsum('nd,nd->n', X, beta_j[j_ind])
        y = y + rng.randn(X.shape[0])*sigma
        return Data(X, y, Nj, j_lim, phi_true, {'sigma_x': sigma_x_j, 'Sigma_x': Sigma_x})

    def get_prior(self):
        v = np.full(self.dphi, self.V0)
        m0 = np.zeros(self.dphi)
        return np.diag(v).T, m0, np.diag(1/v).T, m0/v


class _GaussBase(object):
    """Shared skeleton of the remaining linear-regression simulators (noise sigma = 1, all prior
    variances 1.5^2, zero prior means): the model-specific part draws the parameters in the
    reference's order of random draws and returns (alpha_j, beta or beta_j, phi_true[1:])."""
    SIGMA = 1

    def __init__(self, J, D, npg):
        self.J, self.D, self.npg = J, D, npg
        self.dphi = self._dphi(D)

    def simulate_data(self, Sigma_x=None, rng=None):
        J, D = self.J, self.D
        if not isinstance(rng, np.random.RandomState):
            rng = np.random.RandomState(rng)
        seed_input_cov = rng.randint(2**31 - 1)
        if isinstance(Sigma_x, str) and Sigma_x == 'rand':
            Sigma_x = rand_corr_vine(D, seed=seed_input_cov)
        Nj, j_lim = _sizes(rng, J, self.npg)
        N = int(np.sum(Nj))
        sigma = self.SIGMA
        alpha_j, beta, phi_rest = self._draw_parameters(rng)
        sigma_x = calc_input_param_lin_reg(beta, sigma, Sigma_x)
        j_ind = np.repeat(np.arange(J), Nj)
        if beta.ndim == 1:                      # one coefficient vector: one input scale, one draw of X
            X = rng.randn(N, D)*sigma_x if Sigma_x is None else rng.randn(N, D).dot(sigma_x*cholesky(Sigma_x))
            y = alpha_j[j_ind] + X.dot(beta)
        else:
            X = _draw_X(rng, Nj, j_lim, D, np.zeros(J), sigma_x, Sigma_x)
            y = alpha_j[j_ind] + np.einsum('nd,nd->n', X, beta[j_ind])
        y = y + rng.randn(N)*sigma
        return Data(X, y, Nj, j_lim, np.append(np.log(sigma), phi_rest), {'sigma_x': sigma_x, 'Sigma_x': Sigma_x})

    def get_prior(self):
        v = np.full(self.dphi, 1.5**2)
        m0 = np.zeros(self.dphi)
        return np.diag(v).T, m0, np.diag(1/v).T, m0/v


class m2a(_GaussBase):
    """phi = [log sigma, log sigma_a, log sigma_b], beta ~ N(0, sigma_b) shared (models/m2a.py; density
    m2a_sg.stan).  The reference's simulator names its generator `rnd_data` but receives it as
    `rng` (m2a.py:91-174) and raises NameError; this is the evident intent."""
    site_model = 'm2a_sg'

    def _dphi(self, D):
        return 3

    def _draw_parameters(self, rng):
        alpha_j = rng.randn(self.J)
        beta = rng.randn(self.D)
        _regulate_rows(rng, beta[None, :], lambda index: rng.randn())
        return alpha_j, beta, np.zeros(2)


class m3a(_GaussBase):
    """phi = [log sigma, log sigma_a, log sigma_b(D)] (models/m3a.py; density m3a_sg.stan)."""
    site_model = 'm3a_sg'

    def _dphi(self, D):
        return D + 2

    def _draw_parameters(self, rng):
        sigma_b = np.exp(rng.randn(self.D))
        alpha_j = rng.randn(self.J)
        beta_j = rng.randn(self.J, self.D)*sigma_b
        _regulate_rows(rng, beta_j, lambda index: rng.randn()*sigma_b[index])
        return alpha_j, beta_j, np.append(0.0, np.log(sigma_b))


class m5a(_GaussBase):
    """m4a with Laplace group effects and half-Cauchy scales (models/m5a.py; density m5a_sg.stan)."""
    MU_A, SIGMA_A, SIGMA_MB, SIGMA_SB = 0.1, 1, 0, 1   # m5a.py:49-55
    site_model = 'm5a_sg'

    def _dphi(self, D):
        return 2*D + 3

    def _draw_parameters(self, rng):
        J, D = self.J, self.D
        sigma_b = np.abs(rng.standard_cauchy(D)*self.SIGMA_SB)
        mu_b = rng.laplace(size=D)*self.SIGMA_MB
        alpha_j = self.MU_A + rng.laplace(size=J)*self.SIGMA_A
        beta_j = mu_b + rng.laplace(size=(J, D))*sigma_b
        _regulate_rows(rng, beta_j, lambda index: mu_b[index] + rng.randn()*sigma_b[index])
        return alpha_j, beta_j, np.concatenate(([self.MU_A, np.log(self.SIGMA_A)], mu_b, np.log(sigma_b)))


MODELS = {'m1b': m1b, 'm2b': m2b, 'm3b': m3b, 'm4b': m4b, 'm5b': m5b,
          'm1a': m1a, 'm2a': m2a, 'm3a': m3a, 'm4a': m4a, 'm5a': m5a}


def default_df0(K):
    """fit.py:171-186: exponential decay from 0.5 to min(1/K, 0.2)."""
    start, end, decay_at_k = 0.5, min(1/K, 0.2), 0.9
    t = -np.log(1 - decay_at_k)/(K - 1)
    a = start - end
    return lambda curiter: a*np.exp(-t*(curiter - 1)) + end
