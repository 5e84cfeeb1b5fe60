"""CPU restatement (NumPy/SciPy) of ep-stan's EP hot path -- TEST INFRASTRUCTURE.

This file is the oracle of the repository: it restates, function by function,
what /root/reference/epstan/method.py and util.py compute on the hot path
(SURVEY.md §8a rows a1, a4-a12). Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product (ep-stan_amd/) never
does, and fails loudly when its HIP library is missing.

Parity status
  * everything downstream of the samples (this file): PINNED -- checked against
    the golden vectors of tests/golden/*.npz, which were captured by importing
    the reference itself (tests/golden/make_golden.py).
  * the sampler (oracle/nuts_oracle.c): PARITY UNPINNED -- the reference's
    sampler is PyStan 2.17.0.0 / Stan C++ NUTS (README.md:10, util.py:34,716),
    a third-party dependency that is absent from /root/reference and cannot be
    installed; nuts_oracle.c restates Stan 2.17's published algorithm.

All matrices are float64; (d,d,K) site arrays are Fortran-ordered like the
reference's (method.py:838-851).
"""

import numpy as np
from scipy import linalg

MAX_UINT = 2**31 - 1          # pystan.constants.MAX_UINT (method.py:40)

INFO_OK = 0                    # method.py:620-624
INFO_INVALID_PRIOR = 1
INFO_DF_TRESHOLD_REACHED_GLOBAL = 2
INFO_DF_TRESHOLD_REACHED_CAVITY = 3
INFO_ALL_SITES_FAIL = 4
MIN_EIG_TRESHOLD = 1e-5        # method.py:627-628
MIN_EIG = 0.5


class NotPosDef(Exception):
    """Stands for scipy.linalg.LinAlgError in the reference."""


# ----------------------------------------------------------------------------
# util.py:51-125
def invert_normal_params(A, b=None, cho_form=False):
    """(A, b) -> (A^-1, A^-1 b) for SPD A, or for A given as its UPPER Cholesky
    factor when cho_form (util.py:110-125: cho_factor, cho_solve, dpotri,
    copy_triu_to_tril). Returns new arrays; raises NotPosDef."""
    A = np.array(A, dtype=np.float64, order='F')
    if not cho_form:
        try:
            U = linalg.cholesky(A, lower=False)
        except linalg.LinAlgError as ex:
            raise NotPosDef(str(ex))
    else:
        U = np.triu(A)
        if np.any(np.diag(U) == 0.0) or not np.all(np.isfinite(np.diag(U))):
            # dpotri returns info > 0 for an exactly singular factor (util.py:119-122)
            raise NotPosDef('singular Cholesky factor')
    out_b = None
    if b is not None:
        out_b = linalg.cho_solve((U, False), np.array(b, dtype=np.float64))
    Uinv = linalg.solve_triangular(U, np.eye(U.shape[0]), lower=False)
    out_A = Uinv.dot(Uinv.T)
    out_A = np.asfortranarray(0.5 * (out_A + out_A.T))
    return out_A, out_b


def fro_norm_squared(A):
    """cython_util.pyx:17-40."""
    return float(np.sum(np.square(A)))


# util.py:128-194
def olse(S, n, P=None):
    """Optimal linear shrinkage precision estimate (Bodnar et al.,
    arXiv:1308.0931) of a sample covariance S from n draws."""
    d = S.shape[0]
    out, _ = invert_normal_params(S)
    tr = np.trace(out)
    tr2 = tr**2
    f2 = fro_norm_squared(out)
    if P is None:
        alpha = 1 - (d + tr2 / (f2 - tr2 / d)) / n           # util.py:181
        beta = tr * (1 - d / n - alpha)                       # util.py:182
        out = out * alpha
        out[np.diag_indices(d)] += beta / d                   # util.py:184
    else:
        f2p = fro_norm_squared(P)
        trSP = np.sum(out * P)                                # util.py:188-189
        alpha = 1 - (d + tr2 * f2p / (f2 * f2p - trSP**2)) / n
        beta = (trSP / f2p) * (1 - d / n - alpha)
        out = out * alpha + beta * P                          # util.py:192-193
    return np.asfortranarray(out)


# method.py:267-302
def cavity(Q, r, Qi, ri):
    """Cavity precision Mat = Q - Qi and cavity MEAN vec = Mat^-1 (r - ri).
    Returns (Mat, vec, posdef). When not posdef, vec holds r - ri (the
    reference leaves the un-solved vector in place)."""
    Mat = np.asfortranarray(Q - Qi)
    vec = np.array(r - ri, dtype=np.float64)
    try:
        cho = linalg.cho_factor(Mat.copy(order='F'))
        vec = linalg.cho_solve(cho, vec)
    except linalg.LinAlgError:
        return Mat, vec, False
    return Mat, vec, True


# method.py:410-475
def tilted_moments(samp, Q, r, prec_estim='sample'):
    """Moment stage of Worker.tilted: draws (S,d) -> site deltas.

    Returns (dQi, dri, mt, scatter, posdef). `scatter` is the un-normalised
    C'C of the centred draws (what Worker.Mat holds in the 'olse' branch,
    method.py:446; the 'sample' branch holds its QR factor R with R'R equal to
    it, :420-427)."""
    samp = np.array(samp, dtype=np.float64)
    S, d = samp.shape
    mt = samp.mean(axis=0)                                    # :415 / :442
    C = samp - mt                                             # :417 / :444
    scatter = C.T.dot(C)
    try:
        if prec_estim == 'sample':
            # QR route of :420-431 == inverse of the scatter matrix
            R = np.linalg.qr(C, mode='r')
            dQi, dri = invert_normal_params(R, mt, cho_form=True)
            unbias_k = S - d - 2                              # :433-435
            dQi = dQi * unbias_k
            dri = dri * unbias_k
        elif prec_estim == 'olse':
            dQi = olse(scatter / S, S, P=Q)                   # :448-450
            dri = dQi.dot(mt)                                 # :451
        else:
            raise ValueError('Invalid value for option `prec_estim`')
        dQi = np.asfortranarray(dQi - Q)                      # :457-458
        dri = dri - r
    except NotPosDef:
        return (np.zeros((d, d), order='F'), np.zeros(d), mt, scatter, False)
    return dQi, dri, mt, scatter, True


# method.py:342-346 and :956-960
def kl_mvn(m0, S0, m1, S1, sum_log_diag_cho_S0=None):
    """KL(N(m0,S0) || N(m1,S1)) (experiment/find_damp.py:38-56)."""
    choS1 = np.linalg.cholesky(S1)
    if sum_log_diag_cho_S0 is None:
        sum_log_diag_cho_S0 = np.sum(np.log(np.diag(np.linalg.cholesky(S0))))
    dm = m1 - m0
    Q1 = np.linalg.inv(S1)
    return (0.5 * (np.trace(Q1.dot(S0)) + dm.dot(Q1.dot(dm)) - len(m0))
            - sum_log_diag_cho_S0 + np.sum(np.log(np.diag(choS1))))


def damp_criteria(Q, r, m_target, S_target, samp_target=None):
    """Selection criteria of one damping trial (find_damp.py:155-163): mse, kl, ll of the
    proposal N(m, S), S = Q^-1, m = S r, against the target."""
    S, m = invert_normal_params(Q, r)
    mse = np.mean((m - m_target)**2)
    kl = kl_mvn(m_target, S_target, m, S)
    ll = np.nan
    if samp_target is not None:
        d = m.shape[0]
        dx = samp_target - m
        ll = -0.5 * (samp_target.shape[0] * (d * np.log(2 * np.pi) + np.linalg.slogdet(S)[1])
                     + np.einsum('si,ij,sj->', dx, Q, dx))
    return mse, kl, ll


def damp_sweep(Q0, r0, Qi, ri, dQi, dri, damps, m_target, S_target, samp_target=None):
    """The loop `for di, df in enumerate(damps)` of find_damp.py:146-173 -> (ndf, 5) array
    [global_pd, cav_pd, mse, kl, ll], criteria NaN unless both flags hold."""
    out = np.full((len(damps), 5), np.nan)
    out[:, :2] = 0.0
    K = Qi.shape[2]
    for di, df in enumerate(damps):
        Qi2 = Qi + df * dQi
        ri2 = ri + df * dri
        Q = Qi2.sum(2) + Q0
        r = ri2.sum(1) + r0
        try:
            np.linalg.cholesky(Q)
        except np.linalg.LinAlgError:
            continue
        out[di, 0] = 1.0
        if all(cavity(Q, r, Qi2[:, :, k], ri2[:, k])[2] for k in range(K)):
            out[di, 1] = 1.0
            out[di, 2:] = damp_criteria(Q, r, m_target, S_target, samp_target)
    return out


def run_seeds(seed, niter, K):
    if isinstance(seed, np.random.RandomState):
        rng = seed
    else:
        rng = np.random.RandomState(seed=seed)
    return rng.randint(0, MAX_UINT, size=(niter, K))


def stan_seed(seed):
    if isinstance(seed, np.random.RandomState):
        rng = seed
    else:
        rng = np.random.RandomState(seed)
    return rng.randint(0, MAX_UINT)


# ----------------------------------------------------------------------------
class OracleWorker(object):
    """State of one site (method.py:121-475) without the sampler."""

    def __init__(self, index, dphi, X, y, A=None, prec_estim='sample',
                 chains=4, iter=1000, warmup=None, thin=1, init='random',
                 init_prev=True):
        self.index = index
        self.dphi = dphi
        self.Mat = np.zeros((dphi, dphi), order='F')
        self.vec = np.zeros(dphi)
        self.phase = 0
        self.nsamp = None
        self.prec_estim = prec_estim
        self.stan_params = dict(chains=chains, iter=iter, warmup=warmup,
                                thin=thin, init=init)
        self.init_prev = init_prev
        self.data = dict(N=X.shape[0], X=X, y=y, mu_phi=self.vec,
                         Omega_phi=self.Mat)
        if X.ndim == 2:
            self.data['D'] = X.shape[1]
        if A:
            self.data.update(A)
        self.Q = None
        self.r = None
        self.last_time = None
        self.last_msteps = None
        self.last_mrhat = None

    def cavity(self, Q, r, Qi, ri):
        self.Q = Q
        self.r = r
        Mat, vec, ok = cavity(Q, r, Qi, ri)
        self.Mat[...] = Mat
        self.vec[...] = vec
        self.phase = 1 if ok else 0
        return ok

    def tilted(self, dQi, dri, sampler, seed=None):
        """sampler(data, stan_params) -> (samp (S,d), lastsamp, time, msteps, mrhat)."""
        if self.phase != 1:
            raise RuntimeError('Cavity has to be calculated before tilted.')
        self.stan_params['seed'] = stan_seed(seed)
        samp, lastsamp, dur, msteps, mrhat = sampler(self.data, self.stan_params)
        self.last_time, self.last_msteps, self.last_mrhat = dur, msteps, mrhat
        if self.init_prev:
            self.stan_params['init'] = lastsamp
        self.nsamp = samp.shape[0]
        d1, d2, mt, scatter, ok = tilted_moments(samp, self.Q, self.r,
                                                 self.prec_estim)
        self.vec[...] = mt
        self.Mat[...] = scatter
        dQi[...] = d1
        dri[...] = d2
        self.phase = 2 if ok else 0
        return ok


class OracleMaster(object):
    """Master.__init__ + Master.run (method.py:647-882, 899-1247) restated.

    `sampler(data, stan_params)` replaces the Stan subprocess and returns
    `(samp, lastsamp, time, mean_stepsize, max_rhat)`.
    """

    def __init__(self, X, y, site_sizes, sampler, dphi=None, prior=None,
                 init_site=None, df0=None, df_decay=0.8, df_treshold=1e-6,
                 A_k=None, **worker_opts):
        self.X = np.ascontiguousarray(X)
        self.y = np.ascontiguousarray(y)
        self.N = X.shape[0]
        self.Nk = np.asarray(site_sizes)
        self.K = len(self.Nk)
        self.k_lim = np.concatenate(([0], np.cumsum(self.Nk)))   # :700
        if self.k_lim[-1] != self.N:
            raise ValueError('Site definition does not match with `X`')
        if np.any(self.Nk == 0):
            raise ValueError('Empty sites')
        if self.K < 2:
            raise ValueError('Distributed EP should be run with at least two sites.')
        self.sampler = sampler
        if prior is None:
            if dphi is None:
                raise ValueError('If arg. `prior` is not provided, arg. `dphi` has to be given')
            self.Q0 = np.asfortranarray(np.eye(dphi))
            self.r0 = np.zeros(dphi)
        elif 'Q' in prior:
            self.Q0 = np.asfortranarray(prior['Q'])
            self.r0 = np.asarray(prior['r'], dtype=np.float64)
        else:
            self.Q0, self.r0 = invert_normal_params(prior['S'], prior['m'])
        self.dphi = d = self.Q0.shape[0]
        self.df_decay = df_decay
        self.df_treshold = df_treshold
        if df0 is None:
            default_df = 1 / self.K
            self.df0 = lambda i: default_df                      # :802-805
        elif isinstance(df0, (float, int)):
            if df0 <= 0 or df0 > 1:
                raise ValueError('Constant initial damping factor has to be in (0,1]')
            self.df0 = lambda i: df0
        else:
            self.df0 = df0
        self.workers = []
        for k in range(self.K):
            A = {}
            if A_k:
                for key, val in A_k.items():
                    A[key] = val[k]
            self.workers.append(OracleWorker(
                k, d, self.X[self.k_lim[k]:self.k_lim[k + 1]],
                self.y[self.k_lim[k]:self.k_lim[k + 1]], A=A, **worker_opts))
        K = self.K
        self.S = np.zeros((d, d), order='F')
        self.m = np.zeros(d)
        self.Qi = np.zeros((d, d, K), order='F')
        self.ri = np.zeros((d, K), order='F')
        self.Qi2 = np.zeros((d, d, K), order='F')
        self.ri2 = np.zeros((d, K), order='F')
        self.dQi = np.zeros((d, d, K), order='F')
        self.dri = np.zeros((d, K), order='F')
        if init_site is not None:                                # :853-861
            if isinstance(init_site, np.ndarray):
                for k in range(K):
                    self.Qi[:, :, k] = init_site
            else:
                for k in range(K):
                    self.Qi[:, :, k][np.diag_indices(d)] = K / (init_site**2)
        self.iter = 0
        self.Q = np.asfortranarray(self.Qi.sum(2) + self.Q0)     # :867-868
        self.r = self.ri.sum(1) + self.r0
        try:
            linalg.cho_factor(self.Q.copy())
        except linalg.LinAlgError as ex:
            raise ValueError('Initial approximation is not pos.def.') from ex
        for k, w in enumerate(self.workers):                     # :877-882
            if not w.cavity(self.Q, self.r, self.Qi[:, :, k], self.ri[:, k]):
                raise ValueError('Initial cavity is not pos.def.')
        self.df_log = []          # (iter, accepted df) -- oracle-only diagnostics

    def cur_approx(self):
        return invert_normal_params(self.Q, self.r)

    def _force_pd(self, posdefs):
        """method.py:1119-1129 / 1194-1204 (min-eig shift is applied to Qi)."""
        posdefs[:] = False
        d = self.dphi
        for k in range(self.K):
            min_eig = np.linalg.eigvalsh(self.Qi2[:, :, k])[0]
            if min_eig < MIN_EIG_TRESHOLD:
                self.Qi[:, :, k][np.diag_indices(d)] += MIN_EIG - min_eig
                posdefs[k] = True

    def run(self, niter, calc_moments=True, seed=None):
        """Returns (info, (m_phi_s, cov_phi_s), (stimes, msteps, mrhats))."""
        K, d = self.K, self.dphi
        seeds = run_seeds(seed, niter, K)
        posdefs = np.empty(K, dtype=bool)
        m_phi_s = np.zeros((niter, d))
        cov_phi_s = np.zeros((niter, d, d))
        stimes = np.zeros(niter)
        msteps = np.zeros(niter)
        mrhats = np.zeros(niter)
        for cur_iter in range(niter):
            self.iter += 1
            for k in range(K):                                   # :1005-1023
                posdefs[k] = self.workers[k].tilted(
                    self.dQi[:, :, k], self.dri[:, k], self.sampler,
                    seed=seeds[cur_iter, k])
            if not np.any(posdefs):                              # :1033-1040
                return INFO_ALL_SITES_FAIL, (m_phi_s, cov_phi_s), (stimes, msteps, mrhats)
            stimes[cur_iter] = max(w.last_time for w in self.workers)
            msteps[cur_iter] = max(w.last_msteps for w in self.workers)
            mrhats[cur_iter] = max(w.last_mrhat for w in self.workers)
            df = self.df0(self.iter)                             # :1060
            failed_force_pos_def = False
            while True:                                          # :1067
                self.Qi2[...] = self.Qi + df * self.dQi          # :1071-1074
                self.ri2[...] = self.ri + df * self.dri
                self.Q[...] = self.Qi2.sum(2) + self.Q0
                self.r[...] = self.ri2.sum(1) + self.r0
                try:
                    cho_Q = linalg.cho_factor(self.Q.copy(order='F'))
                    global_ok = True
                except linalg.LinAlgError:
                    global_ok = False
                cav_ok = False
                if global_ok:
                    cav_ok = True
                    for k in range(K):                           # :1138-1143
                        if not self.workers[k].cavity(
                                self.Q, self.r, self.Qi2[:, :, k], self.ri2[:, k]):
                            cav_ok = False
                            break
                    if cav_ok:                                   # :1145-1158
                        self.Qi, self.Qi2 = self.Qi2, self.Qi
                        self.ri, self.ri2 = self.ri2, self.ri
                        self.df_log.append((self.iter, df))
                        break
                df *= self.df_decay                              # :1083 / :1163
                if not global_ok and self.iter == 1:             # :1092-1101
                    return INFO_INVALID_PRIOR, (m_phi_s, cov_phi_s), (stimes, msteps, mrhats)
                if df < self.df_treshold:                        # :1102-1132 / :1177-1207
                    df = self.df0(self.iter)
                    self.Qi2[...] = self.Qi + df * self.dQi
                    self.ri2[...] = self.ri + df * self.dri
                    if failed_force_pos_def:
                        return (INFO_DF_TRESHOLD_REACHED_CAVITY,
                                (m_phi_s, cov_phi_s), (stimes, msteps, mrhats))
                    failed_force_pos_def = True
                    self._force_pd(posdefs)
            if calc_moments:                                     # :1211-1219
                self.S, self.m = invert_normal_params(self.Q, self.r)
                m_phi_s[cur_iter] = self.m
                cov_phi_s[cur_iter] = self.S.T
        return INFO_OK, (m_phi_s, cov_phi_s), (stimes, msteps, mrhats)


# ----------------------------------------------------------------------------
# Site log-densities (experiment/models/m*b_sg.stan).
MODELS = {'m1b_sg': 0, 'm2b_sg': 1, 'm3b_sg': 2, 'm4b_sg': 3, 'm5b_sg': 4}


def model_dims(model, D):
    """(dphi, P) of the single-group models; the Gaussian-likelihood family m*a_sg.stan has one
    more shared parameter in front, log sigma."""
    if model[2] == 'a':
        d, P = model_dims(model.replace('a', 'b', 1), D)
        return d + 1, P + 1
    if model == 'm1b_sg':
        return D + 1, D + 2               # m1b_sg.stan:19-22
    if model == 'm2b_sg':
        return 2, D + 3                   # m2b_sg.stan:19-23
    if model == 'm3b_sg':
        return D + 1, 2 * D + 2           # m3b_sg.stan:19-23
    if model in ('m4b_sg', 'm5b_sg'):
        return 2 * D + 2, 3 * D + 3       # m4b_sg.stan:19-23
    raise ValueError(model)


def site_logdensity(model, theta, X, y, mu, Omega):
    """lp(theta) and its gradient, NumPy, for cross-checking nuts_oracle.c."""
    theta = np.asarray(theta, dtype=np.float64)
    n, D = X.shape
    if model[2] == 'a':
        # experiment/models/m*a_sg.stan: theta = [log sigma | theta of the b-model], y ~ normal(f, sigma).
        # Written on its own (value by the Stan program's formula, gradient by hand) so that the
        # shared code below stays the b-family's.
        return _site_logdensity_gauss(model, theta, X, y, mu, Omega)
    d, P = model_dims(model, D)
    phi = theta[:d]
    eta = theta[d]
    etb = theta[d + 1:] if P > d + 1 else None
    g = np.zeros(P)
    if model == 'm1b_sg':
        sa = np.exp(phi[0]); alpha = eta * sa; beta = phi[1:]
    elif model == 'm2b_sg':
        sa = np.exp(phi[0]); sb = np.exp(phi[1]); alpha = eta * sa; beta = etb * sb
    elif model == 'm3b_sg':
        sa = np.exp(phi[0]); sb = np.exp(phi[1:]); alpha = eta * sa; beta = etb * sb
    else:
        sa = np.exp(phi[1]); sb = np.exp(phi[2 + D:]); alpha = phi[0] + eta * sa
        beta = phi[2:2 + D] + etb * sb
    f = alpha + X.dot(beta)
    yy = np.asarray(y, dtype=np.float64)
    ll = np.sum(yy * f - np.logaddexp(0.0, f))
    gf = yy - 1.0 / (1.0 + np.exp(-f))
    da = np.sum(gf)
    db = X.T.dot(gf)
    v = phi - mu
    Ov = Omega.dot(v)
    lp = -0.5 * v.dot(Ov) + ll
    g[:d] = -Ov
    laplace = model == 'm5b_sg'
    if laplace:
        lp += -abs(eta) - (np.sum(np.abs(etb)))
    else:
        lp += -0.5 * eta**2 - (0.5 * np.sum(etb**2) if etb is not None else 0.0)
    if model == 'm1b_sg':
        g[0] += da * eta * sa; g[1:d] += db; g[d] = da * sa - eta
    elif model == 'm2b_sg':
        g[0] += da * eta * sa; g[1] += db.dot(etb) * sb
        g[d] = da * sa - eta; g[d + 1:] = db * sb - etb
    elif model == 'm3b_sg':
        g[0] += da * eta * sa; g[1:d] += db * etb * sb
        g[d] = da * sa - eta; g[d + 1:] = db * sb - etb
    else:
        g[0] += da; g[1] += da * eta * sa
        g[2:2 + D] += db; g[2 + D:d] += db * etb * sb
        if laplace:
            g[d] = da * sa - np.sign(eta); g[d + 1:] = db * sb - np.sign(etb)
        else:
            g[d] = da * sa - eta; g[d + 1:] = db * sb - etb
    return lp, g


def _site_logdensity_gauss(model, theta, X, y, mu, Omega):
    n, D = X.shape
    d, P = model_dims(model, D)
    base = model[:2]
    ls = theta[0]
    phi = theta[1:d]                                    # the b-model's phi
    eta = theta[d]
    etb = theta[d + 1:] if P > d + 1 else None
    if base == 'm1':
        sa = np.exp(phi[0]); alpha = eta * sa; beta = phi[1:]
    elif base == 'm2':
        sa = np.exp(phi[0]); sb = np.exp(phi[1]); alpha = eta * sa; beta = etb * sb
    elif base == 'm3':
        sa = np.exp(phi[0]); sb = np.exp(phi[1:]); alpha = eta * sa; beta = etb * sb
    else:
        sa = np.exp(phi[1]); sb = np.exp(phi[2 + D:]); alpha = phi[0] + eta * sa
        beta = phi[2:2 + D] + etb * sb
    res = np.asarray(y, dtype=np.float64) - (alpha + X.dot(beta))
    s2 = np.exp(2 * ls)
    v = theta[:d] - mu
    Ov = Omega.dot(v)
    lp = -0.5 * v.dot(Ov) - n * ls - 0.5 * res.dot(res) / s2
    laplace = base == 'm5'
    if laplace:
        lp += -abs(eta) - np.sum(np.abs(etb))
    else:
        lp += -0.5 * eta**2 - (0.5 * np.sum(etb**2) if etb is not None else 0.0)
    gf = res / s2
    da, db = gf.sum(), X.T.dot(gf)
    g = np.zeros(P)
    g[:d] = -Ov
    g[0] += res.dot(res) / s2 - n
    gp = g[1:d]                                         # view: the b-model's phi block
    pr = (lambda t: np.sign(t)) if laplace else (lambda t: t)
    if base == 'm1':
        gp[0] += da * eta * sa; gp[1:] += db; g[d] = da * sa - eta
    elif base == 'm2':
        gp[0] += da * eta * sa; gp[1] += db.dot(etb) * sb
        g[d] = da * sa - eta; g[d + 1:] = db * sb - etb
    elif base == 'm3':
        gp[0] += da * eta * sa; gp[1:] += db * etb * sb
        g[d] = da * sa - eta; g[d + 1:] = db * sb - etb
    else:
        gp[0] += da; gp[1] += da * eta * sa
        gp[2:2 + D] += db; gp[2 + D:] += db * etb * sb
        g[d] = da * sa - pr(eta); g[d + 1:] = db * sb - pr(etb)
    return lp, g


def site_logdensity_groups(model, theta, X, y, j_ind, mu, Omega):
    """lp(theta) of the multi-group programs experiment/models/m{1..5}b.stan (K < J): `J` groups
    in the site, `j_ind[n]` (0-based) the group of row n; theta = [phi, eta (J), etb (J x D)].
    Value only (the tests differentiate it numerically)."""
    theta = np.asarray(theta, dtype=np.float64)
    n, D = X.shape
    base = model.replace('_sg', '')
    d = model_dims(base + '_sg', D)[0]
    j_ind = np.asarray(j_ind)
    J = int(j_ind.max()) + 1
    phi = theta[:d]
    eta = theta[d:d + J]
    etb = theta[d + J:].reshape(J, D) if base != 'm1b' else None
    if base == 'm1b':                                   # m1b.stan:24-38
        alpha = eta * np.exp(phi[0]); beta = np.tile(phi[1:], (J, 1))
    elif base == 'm2b':                                 # m2b.stan
        alpha = eta * np.exp(phi[0]); beta = etb * np.exp(phi[1])
    elif base == 'm3b':                                 # m3b.stan
        alpha = eta * np.exp(phi[0]); beta = etb * np.exp(phi[1:])[None, :]
    else:                                               # m4b.stan:33-41, m5b.stan
        alpha = phi[0] + eta * np.exp(phi[1]); beta = phi[2:2 + D][None, :] + etb * np.exp(phi[2 + D:])[None, :]
    f = alpha[j_ind] + np.einsum('nd,nd->n', X, beta[j_ind])
    yy = np.asarray(y, dtype=np.float64)
    lp = np.sum(yy * f - np.logaddexp(0.0, f))
    v = phi - mu
    lp += -0.5 * v.dot(Omega.dot(v))
    if base == 'm5b':
        lp += -np.sum(np.abs(eta)) - np.sum(np.abs(etb))
    else:
        lp += -0.5 * np.sum(eta**2) - (0.5 * np.sum(etb**2) if etb is not None else 0.0)
    return lp
