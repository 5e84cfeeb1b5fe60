"""CPU stand-in for ep-stan_amd.engine.HipEngine -- TEST INFRASTRUCTURE.

Implements the engine interface with the NumPy / C oracle so that
  * tests can drive `Master`'s host logic (partitioning, damping state machine,
    multi-rank reduction over gloo) without a GPU, and
  * bench.py's cpu_baseline leg can time the same EP iteration on host cores.
It is never imported by the package; `Master` only receives it through the
private `_engine_factory` argument from tests/ and bench.py.
"""

import time

import numpy as np

from . import ep_oracle as eo
from . import nuts_oracle as no

QI, QI2, DQI = 0, 1, 2


class OracleEngine(object):
    def __init__(self, model, X, y, k_lim, nthreads=0, g_cnt=None, g_lim=None):
        self.model = model
        self.g_cnt = None if g_cnt is None else np.ascontiguousarray(g_cnt, dtype=np.int32)
        self.g_lim = None if g_lim is None else np.ascontiguousarray(g_lim, dtype=np.int64)
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64 if no.is_gauss(model) else np.int32)
        self.k_lim = np.ascontiguousarray(k_lim, dtype=np.int64)
        self.K = self.k_lim.shape[0] - 1
        self.D = self.X.shape[1]
        self.d, self.P = no.dims(model, self.D, 1 if g_cnt is None else int(np.max(g_cnt)))
        d, K = self.d, self.K
        self.packed_len = 2 * (d * d + d)
        self.nthreads = nthreads
        self.Q0 = np.zeros((d, d), order='F'); self.r0 = np.zeros(d)
        self.Q = np.zeros((d, d), order='F'); self.r = np.zeros(d)
        self.Qi = np.zeros((d, d, K), order='F'); self.ri = np.zeros((d, K), order='F')
        self.dQi = np.zeros((d, d, K), order='F'); self.dri = np.zeros((d, K), order='F')
        self.cav_Om = np.zeros((K, d, d)); self.cav_mu = np.zeros((K, d))
        self.tilt_scatter = np.zeros((K, d, d)); self.tilt_mean = np.zeros((K, d))
        self.last_df = 0.0
        self.nsamp = 0
        self.draws = None
        self.last = None
        self.chain_stats = None
        self.carry_eps = None
        self.carry_metric = None

    def close(self):
        pass

    # ---- state
    def set_prior(self, Q0, r0):
        self.Q0[...] = Q0; self.r0[...] = r0

    def _arr(self, which):
        if which == QI:
            return self.Qi, self.ri
        if which == DQI:
            return self.dQi, self.dri
        return (np.asfortranarray(self.Qi + self.last_df * self.dQi),
                np.asfortranarray(self.ri + self.last_df * self.dri))

    def set_sites(self, which, Q=None, r=None):
        A, a = self._arr(which)
        if Q is not None:
            A[...] = Q
        if r is not None:
            a[...] = r

    def get_sites(self, which, Q=None, r=None):
        A, a = self._arr(which)
        return A.copy(order='F'), a.copy(order='F')

    def set_site(self, which, k, Q=None, r=None):
        A, a = self._arr(which)
        if Q is not None:
            A[:, :, k] = Q
        if r is not None:
            a[:, k] = r

    def get_site(self, which, k):
        A, a = self._arr(which)
        return A[:, :, k].copy(order='F'), a[:, k].copy()

    def set_global(self, Q, r):
        self.Q[...] = Q; self.r[...] = r

    def get_global(self):
        return self.Q.copy(order='F'), self.r.copy()

    # ---- cavity
    def _cavity(self, k, Qi, ri):
        Mat, vec, ok = eo.cavity(self.Q, self.r, Qi, ri)
        self.cav_Om[k] = Mat
        self.cav_mu[k] = vec
        return ok

    def cavity_batch(self, which, k0=0, count=None):
        count = self.K - k0 if count is None else count
        A, a = self._arr(which)
        return np.array([self._cavity(k, A[:, :, k], a[:, k]) for k in range(k0, k0 + count)])

    def cavity_site(self, k, Q, r, Qi, ri):
        self.set_global(Q, r)
        return self._cavity(k, np.asarray(Qi), np.asarray(ri))

    def get_cavity(self, k):
        return np.asfortranarray(self.cav_Om[k]), self.cav_mu[k].copy()

    # ---- tilted
    @staticmethod
    def sampler_opts(chains=4, iter=1000, warmup=None, thin=1, init='random', max_depth=10, layout=0, flags=0,
                     adapt='fresh'):
        return dict(chains=chains, iter=iter, warmup=warmup, thin=thin, init=init, max_depth=max_depth,
                    carry=(adapt == 'carry'))

    def sample_batch(self, seeds, opts, k0=0, count=None):
        count = self.K - k0 if count is None else count
        o = opts if isinstance(opts, dict) else dict(
            chains=opts.chains, iter=opts.iter, warmup=None if opts.warmup < 0 else opts.warmup,
            thin=opts.thin, init={0: 'random', 1: '0', 2: 'prev'}[opts.init], max_depth=opts.max_depth,
            carry=bool(opts.reserved & 2))
        sl = slice(k0, k0 + count)
        lim = self.k_lim[k0:k0 + count + 1]
        init = None
        if o['init'] == 'prev':
            init = self.last[sl]
        elif o['init'] in ('0', 0):
            init = np.zeros((count, o['chains'], self.P))
        t0 = time.time()
        grp = {}
        if self.g_cnt is not None:
            off = np.concatenate(([0], np.cumsum(self.g_cnt)))
            grp = dict(g_cnt=self.g_cnt[sl], g_lim=self.g_lim[off[k0]:off[k0 + count] + 1] - lim[0])
        draws, last, stats = no.nuts_sites(
            self.model, self.X[lim[0]:lim[-1]], self.y[lim[0]:lim[-1]], lim - lim[0],
            self.cav_mu[sl], self.cav_Om[sl], seeds, chains=o['chains'], iter=o['iter'],
            warmup=o['warmup'], thin=o['thin'], max_depth=o['max_depth'], init=init,
            nthreads=self.nthreads, **grp, **self._carry_args(o, sl, draws_shape=None))
        ms = (time.time() - t0) * 1e3
        if self.draws is None or self.draws.shape[1:] != draws.shape[1:]:
            self.draws = np.zeros((self.K,) + draws.shape[1:])
            self.last = np.zeros((self.K,) + last.shape[1:])
            self.chain_stats = np.zeros((self.K,) + stats.shape[1:])
        self.draws[sl] = draws; self.last[sl] = last; self.chain_stats[sl] = stats
        ce, cm = no.carry_history(draws, stats)                     # what the device keeps after every call
        if self.carry_eps is None or self.carry_eps.shape[1] != ce.shape[1]:
            self.carry_eps = -np.ones((self.K, ce.shape[1])); self.carry_metric = np.ones((self.K, cm.shape[1]))
        self.carry_eps[sl] = ce; self.carry_metric[sl] = cm
        site = np.zeros((count, 8))
        for j in range(count):
            cs = stats[j]
            site[j, 0] = cs[:, 0].mean()
            site[j, 1] = max(no.split_rhat(draws[j, :, :, e]) for e in range(self.P))
            site[j, 2] = cs[:, 2].sum(); site[j, 3] = cs[:, 3].sum(); site[j, 4] = cs[:, 4].sum()
            site[j, 5] = cs[:, 5].mean(); site[j, 6] = cs[:, 6].mean(); site[j, 7] = cs[:, 7].sum()
        return site, ms

    def _carry_args(self, o, sl, draws_shape=None):
        if not o.get('carry') or self.carry_eps is None or self.carry_eps.shape[1] != o['chains']:
            return {}
        return dict(carry_eps=self.carry_eps[sl], carry_metric=self.carry_metric[sl])

    def get_adapt(self, k, chains):
        return self.carry_eps[k].copy(), self.carry_metric[k].copy()

    def _moments(self, k, samp, prec_estim):
        dQ, dr, mt, scatter, ok = eo.tilted_moments(samp, self.Q, self.r, prec_estim)
        self.dQi[:, :, k] = dQ; self.dri[:, k] = dr
        self.tilt_mean[k] = mt; self.tilt_scatter[k] = scatter
        self.nsamp = samp.shape[0]
        return ok

    def tilted_batch(self, seeds, opts, prec_estim, k0=0, count=None):
        count = self.K - k0 if count is None else count
        site, ms = self.sample_batch(seeds, opts, k0, count)
        flags = np.array([self._moments(k, self.get_draws(k), prec_estim)
                          for k in range(k0, k0 + count)])
        return flags, site, ms

    def moments_batch(self, samples, prec_estim, k0=0, count=None):
        count = self.K - k0 if count is None else count
        return np.array([self._moments(k0 + j, samples[:, :, j], prec_estim) for j in range(count)])

    def get_tilted(self, k):
        return np.asfortranarray(self.tilt_scatter[k]), self.tilt_mean[k].copy(), self.nsamp

    def num_draws(self):
        return self.draws.shape[1] * self.draws.shape[2]

    def get_draws(self, k, all_params=False):
        dr = self.draws[k].reshape(-1, self.P)
        return np.asfortranarray(dr if all_params else dr[:, :self.d])

    def last_layout(self):
        return 0

    def set_site_order(self, order=None):
        pass

    def set_site_split(self, n_lead):
        pass

    def last_split(self):
        return 0

    def cu_count(self):
        return 256

    def row_passes(self, chains, k0=0, count=None):
        return self.get_chain_stats(chains, k0, count)[:, :, 3].sum(axis=1)

    def get_chain_stats(self, chains, k0=0, count=None):
        count = self.K - k0 if count is None else count
        return self.chain_stats[k0:k0 + count].copy()

    def logdensity_grad(self, k, theta, layout=0):
        lo, hi = self.k_lim[k], self.k_lim[k + 1]
        gl = None
        if self.g_cnt is not None:
            off = np.concatenate(([0], np.cumsum(self.g_cnt)))
            gl = self.g_lim[off[k]:off[k + 1] + 1] - lo
            theta = np.asarray(theta)[:no.dims(self.model, self.D, self.g_cnt[k])[1]]
        return no.logdensity_grad(self.model, self.X[lo:hi], self.y[lo:hi], self.cav_mu[k],
                                  self.cav_Om[k], theta, gl=gl)

    def invert_normal_params(self, A, b):
        return eo.invert_normal_params(A, b)

    # ---- global update
    def site_sums(self, out_tensor=None):
        out = np.concatenate([self.Qi.sum(2).ravel(order='F'), self.ri.sum(1),
                              self.dQi.sum(2).ravel(order='F'), self.dri.sum(1)])
        if out_tensor is not None:
            import torch
            out_tensor.copy_(torch.from_numpy(out))
            return out_tensor
        return out

    def damped_trial(self, df, packed):
        d = self.d
        if not isinstance(packed, np.ndarray):
            packed = packed.cpu().numpy()
        sQ = packed[:d * d].reshape(d, d, order='F'); sr = packed[d * d:d * d + d]
        sdQ = packed[d * d + d:2 * d * d + d].reshape(d, d, order='F'); sdr = packed[2 * d * d + d:]
        self.last_df = df
        self.Q[...] = self.Q0 + sQ + df * sdQ
        self.r[...] = self.r0 + sr + df * sdr
        try:
            np.linalg.cholesky(self.Q)
        except np.linalg.LinAlgError:
            return False, False, -1
        flags = self.cavity_batch(QI2)
        bad = np.nonzero(~flags)[0]
        return True, bad.size == 0, (int(bad[0]) if bad.size else -1)

    def damp_sweep(self, damps, packed, m_target, S_target, samp_target=None):
        d = self.d
        if not isinstance(packed, np.ndarray):
            packed = packed.cpu().numpy()
        sQ = packed[:d * d].reshape(d, d, order='F'); sr = packed[d * d:d * d + d]
        sdQ = packed[d * d + d:2 * d * d + d].reshape(d, d, order='F'); sdr = packed[2 * d * d + d:]
        out = np.full((len(damps), 5), np.nan)
        out[:, :2] = 0.0
        for di, df in enumerate(damps):
            Q = self.Q0 + sQ + df * sdQ
            r = self.r0 + sr + df * sdr
            try:
                np.linalg.cholesky(Q)
            except np.linalg.LinAlgError:
                continue
            out[di, 0] = 1.0
            ok = all(eo.cavity(Q, r, self.Qi[:, :, k] + df * self.dQi[:, :, k],
                               self.ri[:, k] + df * self.dri[:, k])[2] for k in range(self.K))
            if ok:
                out[di, 1] = 1.0
                out[di, 2:] = eo.damp_criteria(Q, r, m_target, S_target, samp_target)
        return out

    def mix_sums(self):
        mm = np.einsum('ki,kj->ij', self.tilt_mean, self.tilt_mean)
        return np.concatenate([self.tilt_scatter.sum(0).ravel(order='F'), self.tilt_mean.sum(0), mm.ravel(order='F')])

    def accept(self, df):
        self.Qi += df * self.dQi
        self.ri += df * self.dri

    def global_moments(self):
        S, m = eo.invert_normal_params(self.Q, self.r)
        return S, m

    def force_pd(self, df, thresh, target):
        forced = np.zeros(self.K, dtype=bool)
        for k in range(self.K):
            min_eig = np.linalg.eigvalsh(self.Qi[:, :, k] + df * self.dQi[:, :, k])[0]
            if min_eig < thresh:
                self.Qi[:, :, k][np.diag_indices(self.d)] += target - min_eig
                forced[k] = True
        return forced
