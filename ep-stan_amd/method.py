"""Distributed EP driver with the API of /root/reference/epstan/method.py.

`Master` and `Worker` keep the reference's constructor arguments, public
attributes (`S m Q r Qi ri Qi2 ri2 dQi dri Q0 r0 K workers iter`, Fortran-ordered
`(d,d,K)` site arrays), methods (`run`, `cur_approx`, `Worker.cavity`,
`Worker.tilted`), return codes and error behaviour (method.py:121-1247), but
every numerical step runs on the GPU through libepx.so:

  reference (CPU)                                   here (MI355X)
  ------------------------------------------------  ------------------------------------
  for k: Worker.tilted -> Stan subprocess (:1005)   ONE batched NUTS kernel over all sites
  mean / dgeqrf / potri per site (:413-437)         batched scatter(MFMA f64)+Cholesky kernel
  Qi2 = Qi + df dQi ; Q = sum_k Qi2 + Q0 (:1071)    site sums once per iteration (affine in df)
                                                    + one RCCL all-reduce across GPUs
  cho_factor(Q) (:1080), for k: Worker.cavity       one small kernel + one batched cavity kernel,
  invert_normal_params(cho_Q, r) (:1215)            queued behind the all-reduce, one host sync

The host mirrors (`master.Qi` ...) are refreshed when `run` returns; inside `run`
the device copies are authoritative.  Sites are sharded contiguously over the
ranks of an optional communicator (`comm=dist.EpxComm()`: RCCL inside libepx.so).
"""

__all__ = ['Worker', 'Master']

import os
import sys
import time

import numpy as np
from numpy.linalg import LinAlgError

from . import dist as _dist
from . import engine as _engine
from . import site_params as _site_params
from .seeds import MAX_UINT, run_seeds, stan_seed, stan_seeds
from .util import invert_normal_params


def _model_name(site_model):
    """'.../m4b_sg', 'm4b_sg.stan' or 'm4b_sg.pkl' -> 'm4b_sg' (util.load_stan
    accepts all three spellings, util.py:642-689)."""
    if not isinstance(site_model, str):
        raise TypeError('site_model has to be a path/name string selecting a built-in '
                        'GPU site model (StanModel instances cannot run on the GPU)')
    base = os.path.basename(site_model)
    for ext in ('.stan', '.pkl'):
        if base.endswith(ext):
            base = base[:-len(ext)]
    return base


def _sort_keywords(given, tables, unknown):
    """Distribute keyword arguments over option tables.

    `tables` is a sequence of dicts of defaults; returns one dict per table holding the
    table's keys with the given value or the default.  A keyword that no table knows raises
    TypeError with the message `unknown` (formatted with the keyword)."""
    known = set()
    for table in tables:
        known.update(table)
    for kw in given:
        if kw not in known:
            raise TypeError(unknown.format(kw))
    return [dict((kw, given.get(kw, default)) for kw, default in table.items()) for table in tables]


class Worker(object):
    """Per-site state and the per-site entry points `cavity` / `tilted`.

    Same constructor and attributes as the reference's Worker
    (method.py:121-265).  `Mat`/`vec` hold the cavity precision / MEAN after
    `cavity` (phase 1) and the unnormalised scatter matrix / tilted mean after
    `tilted` (phase 2); they are fetched from the device on access.

    `last_time` is the device time of the sampling launch the site took part in: inside
    `Master.run` all sites of a rank are sampled by ONE launch, so every worker of the rank
    reports the same value (the reference reports each site's own Stan time; their maximum,
    which is what `run` records in `stimes`, has the same meaning in both).
    """

    DEFAULT_OPTIONS = {
        'init_prev'       : True,
        'prec_estim'      : 'sample',
        'prec_estim_skip' : 0,
        'verbose'         : False
    }

    DEFAULT_STAN_PARAMS = {
        'chains'          : 4,
        'iter'            : 1000,
        'warmup'          : None,
        'thin'            : 1,
        'init'            : 'random'
    }

    PREC_ESTIM_OPTIONS = ('sample', 'olse')

    RESERVED_STAN_PARAMETER_NAMES = ['X', 'y', 'N', 'D', 'mu_phi', 'Omega_phi']

    def __init__(self, index, stan_model, dphi, X, y, A=None, _master=None, **options):
        opt, sampler = _sort_keywords(options, (self.DEFAULT_OPTIONS, self.DEFAULT_STAN_PARAMS),
                                      "Unexpected option '{}'")
        # sampler settings first: `init_prev` constrains `init`
        self.stan_params = sampler
        self.init_prev = opt['init_prev']
        if self.init_prev:
            self.init_orig = sampler['init']
            if not isinstance(self.init_orig, str):
                raise ValueError("Arg. `init` has to be a string if "
                                 "`init_prev` is True")
        self.prec_estim = opt['prec_estim']
        if self.prec_estim not in self.PREC_ESTIM_OPTIONS:
            raise ValueError("Invalid value for option `prec_estim`")
        self.prec_estim_skip = opt['prec_estim_skip'] if self.prec_estim != 'sample' else 0
        self.verbose = opt['verbose']

        self.index = index
        self.stan_model = stan_model
        self.dphi = dphi
        self.data = dict(N=X.shape[0], X=X, y=y, **(A or {}))
        if X.ndim == 2:
            self.data['D'] = X.shape[1]
        self.phase = 0
        self.iteration = 0
        self.nsamp = None
        self.Q = self.r = None
        self.last_time = self.last_msteps = self.last_mrhat = None
        self.saved_samples = None
        self._Mat = np.zeros((dphi, dphi), order='F')
        self._vec = np.zeros(dphi)
        self._stale = False
        # binding to the device engine (a stand-alone Worker owns a 1-site engine)
        self._master = _master
        self._eng = None
        self._k = None           # local site index inside the engine
        self._has_sampled = False
        if _master is None:
            name = _model_name(stan_model)
            self._eng = _engine.HipEngine(name, np.ascontiguousarray(X), y,
                                          np.array([0, X.shape[0]], dtype=np.int64))
            self._k = 0
            if self._eng.d != dphi:
                raise ValueError('dphi = {} does not match model {} (dphi {})'
                                 .format(dphi, name, self._eng.d))

    # ---- lazily mirrored device state
    def _refresh(self):
        if self._stale and self._eng is not None:
            if self.phase == 1:
                M, v = self._eng.get_cavity(self._k)
            elif self.phase == 2:
                M, v, _ = self._eng.get_tilted(self._k)
            else:
                M = v = None
            if M is not None:
                self._Mat[...] = M
                self._vec[...] = v
        self._stale = False

    @property
    def Mat(self):
        self._refresh()
        return self._Mat

    @property
    def vec(self):
        self._refresh()
        return self._vec

    def _sampler_opts(self):
        sp = self.stan_params
        init = sp['init']
        if not isinstance(init, str) and init != 0:
            init = 'prev'          # list of last draws in the reference (method.py:404-406)
        m = self._master
        return _engine.HipEngine.sampler_opts(
            chains=sp['chains'], iter=sp['iter'], warmup=sp['warmup'], thin=sp['thin'],
            init=init, max_depth=m.max_treedepth if m is not None else 10,
            layout=m.layout if m is not None else 0, adapt=m.adapt if m is not None else 'fresh')

    def _cur_estim(self):
        if self.prec_estim == 'sample' or self.prec_estim_skip > 0:
            return 'sample'
        return self.prec_estim

    def cavity(self, Q, r, Qi, ri):
        """Form the cavity distribution (method.py:267-302).

        Returns True if the cavity precision Q - Qi is positive definite."""
        if self._eng is None:
            raise RuntimeError('this site belongs to another rank')
        self.Q = Q
        self.r = r
        ok = self._eng.cavity_site(self._k, Q, r, Qi, ri)
        self._stale = True
        self.phase = 1 if ok else 0
        if not ok:
            # the reference leaves Mat = Q - Qi, vec = r - ri behind (:288-289)
            self._Mat[...], self._vec[...] = self._eng.get_cavity(self._k)
            self._stale = False
        return ok

    def _save_named(self, names):
        """`saved_samp = {name: draws}` for the requested parameter names (method.py:387-392)."""
        eng = self._eng
        theta = eng.get_draws(self._k, all_params=True)
        ng = 1 if eng.g_cnt is None else int(eng.g_cnt[self._k])
        mid = _engine.MODEL_IDS[eng.model] % 5
        self.saved_samp = _site_params.named_draws(
            mid, eng.D, ng, _engine.is_gauss(eng.model), eng.model.endswith('_sg'), theta, list(names))

    def tilted(self, dQi, dri, save_samples=None, seed=None):
        """Estimate the tilted distribution and write the site parameter update
        into `dQi`, `dri` (method.py:305-475).  Returns False if the precision
        estimate is not positive definite (then dQi, dri are zero).

        `save_samples`: parameter names of the site model whose draws are kept in
        `self.saved_samp` (chain-major, not permuted)."""
        if self.phase != 1:
            raise RuntimeError('Cavity has to be calculated before tilted.')
        if self._eng is None:
            raise RuntimeError('this site belongs to another rank')
        self.stan_params['seed'] = stan_seed(seed)
        m = self._master
        injector = m._sample_injector if m is not None else None
        if self.Q is not None:
            self._eng.set_global(self.Q, self.r)      # dQi -= Q, dri -= r use the aliased arrays (:457-458)
        if injector is not None:
            self._refresh_for_injection()
            samp = np.asfortranarray(injector(self.data, self.stan_params))
            ok = bool(self._eng.moments_batch(samp[:, :, None], self._cur_estim(),
                                              k0=self._k, count=1)[0])
            self.last_time, self.last_msteps, self.last_mrhat = 0.25, 0.125, 1.0625
            lastsamp = [{}] * self.stan_params['chains']
        else:
            flags, stats, ms = self._eng.tilted_batch(
                np.array([self.stan_params['seed']]), self._sampler_opts(), self._cur_estim(),
                k0=self._k, count=1)
            ok = bool(flags[0])
            if stats[0, 7] > 0:
                ok = self._void_update()
            self.last_time = ms * 1e-3
            self.last_msteps = stats[0, 0]
            self.last_mrhat = stats[0, 1]
            lastsamp = 'prev'
            self._has_sampled = True
            if save_samples:
                self._save_named(save_samples)
        if self.verbose:
            print('\n   sampling runtime: {:.4}'.format(self.last_time))
            print('    mean stepsize: {:.4}'.format(self.last_msteps))
            print('    max Rhat: {:.4}'.format(self.last_mrhat))
        if self.init_prev:
            self.stan_params['init'] = lastsamp
        self._finish_tilted(ok)
        dQ, dr = self._eng.get_site(_engine.DQI, self._k)
        dQi[...] = dQ
        dri[...] = dr
        return ok

    def _void_update(self):
        """A chain of this site started at a non-finite density (its draws are its initial point):
        the site update is dropped like a failed precision estimate (method.py:462-475)."""
        d = self.dphi
        self._eng.set_site(_engine.DQI, self._k, np.zeros((d, d), order='F'), np.zeros(d))
        return False

    def _refresh_for_injection(self):
        """Expose the device cavity as the Stan data of method.py:221-222."""
        M, v = self._eng.get_cavity(self._k)
        self.data['mu_phi'] = v
        self.data['Omega_phi'] = M.T

    def _finish_tilted(self, ok):
        self.nsamp = self._eng.get_tilted(self._k)[2]
        if self.prec_estim_skip > 0:
            self.prec_estim_skip -= 1
        if ok:
            self.phase = 2
        else:
            self.phase = 0
            if self.init_prev:
                self.init = self.init_orig        # sic: the reference does not reset stan_params (:468)
        self._stale = True
        self.iteration += 1


# ---------------------------------------------------------------------------------------------
# constructor helpers of Master: every check of method.py:674-814, one concern per function

def _partition(N, site_sizes, site_ind_ord, site_ind):
    """Rows -> sites.  Returns (Nk, k_lim, k_ind, order): rows per site, row limits, site of
    every (sorted) row, and the permutation that sorts the rows by site (None: already sorted).
    Precedence of the three descriptions as in the reference (method.py:696-722)."""
    order = None
    if site_sizes is not None:
        Nk = site_sizes
        k_ind = np.repeat(np.arange(len(Nk), dtype=np.int64), np.asarray(Nk, dtype=np.int64))
    elif site_ind_ord is not None:
        k_ind = site_ind_ord
        Nk = np.bincount(k_ind)
    elif site_ind is not None:
        order = np.argsort(site_ind, kind='mergesort')           # stable: rows keep their order inside a site
        k_ind = site_ind[order]
        Nk = np.bincount(k_ind)
    else:
        raise NotImplementedError("Auto clustering not yet implemented")
    k_lim = np.concatenate(([0], np.cumsum(Nk)))
    if k_lim[-1] != N:
        raise ValueError("Site definition does not match with `X`")
    empty = np.nonzero(np.asarray(Nk) == 0)[0]
    if empty.size:
        raise ValueError("Empty sites: {}. Index the sites from 1 to K-1".format(empty))
    if len(Nk) < 2:
        raise ValueError("Distributed EP should be run with at least "
                         "two sites.")
    return Nk, k_lim, k_ind, order


def _additional_data(A, A_n, A_k, N, K):
    """The three dictionaries of extra Stan data (shared, per row, per site; method.py:736-769):
    lengths must fit and no name may be used twice or shadow the built-in data names."""
    taken = list(Worker.RESERVED_STAN_PARAMETER_NAMES)

    def claim(name):
        if name in taken:
            raise ValueError("Additional data name {} clashes.".format(name))
        taken.append(name)

    for name in A:
        claim(name)
    rows = {}
    for name, val in A_n.items():
        if val.shape[0] != N:
            raise ValueError("The shapes of `A_n[{}]` and `X` does not "
                             "match".format(repr(name)))
        claim(name)
        rows[name] = val if val.flags['CARRAY'] else np.ascontiguousarray(val)
    for name, val in A_k.items():
        if len(val) != K:
            raise ValueError("Array-like length mismatch in `A_k` "
                             "(should be: {}, found: {})"
                             .format(K, len(val)))
        claim(name)
    return A, rows, A_k


def _prior_natural(prior, dphi, device):
    """(Q0, r0, dphi) from the `prior` argument: natural parameters, moment parameters or
    None = unit Gaussian of the given size (method.py:772-797)."""
    if prior is None:
        if dphi is None:
            raise ValueError("If arg. `prior` is not provided, "
                             "arg. `dphi` has to be given")
        return np.asfortranarray(np.eye(dphi)), np.zeros(dphi), dphi
    if not isinstance(prior, dict):
        raise TypeError("Argument `prior` is of wrong type")
    if 'Q' in prior and 'r' in prior:
        Q0 = np.asfortranarray(prior['Q'], dtype=np.float64)
        r0 = np.asarray(prior['r'], dtype=np.float64)
    elif 'S' in prior and 'm' in prior:
        try:
            Q0, r0 = invert_normal_params(np.asarray(prior['S'], dtype=np.float64),
                                          np.asarray(prior['m'], dtype=np.float64), device=device)
        except LinAlgError as ex:
            raise ValueError("Argument `prior` is not appropriate") from ex
    else:
        raise ValueError("Argument `prior` is not appropriate")
    if dphi is None:
        dphi = Q0.shape[0]
    if Q0.shape[0] != dphi or r0.shape[0] != dphi:
        raise ValueError("Arg. `dphi` does not match with `prior`")
    return Q0, r0, dphi


def _damping_schedule(df0, K):
    """Initial damping factor of iteration i as a function (method.py:800-814)."""
    if df0 is None:
        return lambda i, _v=1 / K: _v
    if isinstance(df0, (float, int)):
        if not 0 < df0 <= 1:
            raise ValueError("Constant initial damping factor has to be "
                             "in (0,1]")
        return lambda i, _v=df0: _v
    return df0


class Master(object):
    """Manages the distributed EP algorithm (method.py:478-1247).

    Parameters are those of the reference (site_model, X, y, A, A_n, A_k,
    site_ind, site_ind_ord, site_sizes, dphi, prior, init_site, df0, df_decay,
    df_treshold, overwrite_model, plus the worker options chains, iter, warmup,
    thin, init, init_prev, prec_estim, prec_estim_skip, verbose).  `site_model`
    is a path or name whose basename selects a built-in GPU site model
    (`m1b_sg` ... `m5b_sg`).

    GPU-only keyword arguments: `device` (HIP device index, default: rank's
    LOCAL_RANK or 0), `comm` (a `dist.EpxComm` to shard the sites over several
    GPUs), `max_treedepth` (default 10), `layout` (0 auto, 1 block per site,
    2 block per (site, chain), 3 streaming), `sync_sites` (gather the site arrays
    of all ranks into the host mirrors when `run` returns, default True),
    `balance_sites` (dispatch the sites of an iteration in decreasing order of the
    leapfrogs they took in the previous one; results do not depend on it, default True),
    `adapt`: 'fresh' (default; every site update adapts step size and metric from scratch like
    the reference's fresh `model.sampling` call, util.py:716) or 'carry' (NOT the reference's
    behaviour: a site update starts from the step sizes its chains ended the previous one with
    and from the site's pooled sample variances as metric, warm-up tunes the step size only --
    same target distribution, far fewer leapfrogs; SURVEY.md Appendix A sanctions it as a
    reported opt-in).
    """

    INFO_OK = 0
    INFO_INVALID_PRIOR = 1
    INFO_DF_TRESHOLD_REACHED_GLOBAL = 2
    INFO_DF_TRESHOLD_REACHED_CAVITY = 3
    INFO_ALL_SITES_FAIL = 4

    MIN_EIG_TRESHOLD = 1e-5
    MIN_EIG = 0.5

    DEFAULT_KWARGS = dict(
        A                 = {},
        A_n               = {},
        A_k               = {},
        site_ind          = None,
        site_ind_ord      = None,
        site_sizes        = None,
        dphi              = None,
        prior             = None,
        init_site         = None,
        df0               = None,
        df_decay          = 0.8,
        df_treshold       = 1e-6,
        overwrite_model   = False
    )

    GPU_KWARGS = dict(
        device            = None,
        comm              = None,
        max_treedepth     = 10,
        layout            = 0,
        sync_sites        = True,
        balance_sites     = True,
        adapt             = 'fresh',
        _engine_factory   = None,
    )

    def __init__(self, site_model, X, y, **kwargs):
        own, gpu, w_opt, w_stan = _sort_keywords(
            kwargs, (self.DEFAULT_KWARGS, self.GPU_KWARGS, Worker.DEFAULT_OPTIONS, Worker.DEFAULT_STAN_PARAMS),
            "Unexpected keyword argument '{}'")
        self.worker_options = dict(w_opt, **w_stan)
        self.site_model = site_model
        self.model_name = _model_name(site_model)
        self.max_treedepth = gpu['max_treedepth']
        self.layout = gpu['layout']
        self.sync_sites = gpu['sync_sites']
        self.balance_sites = gpu['balance_sites']
        if gpu['adapt'] not in ('fresh', 'carry'):
            raise ValueError("adapt must be 'fresh' or 'carry'")
        self.adapt = gpu['adapt']
        self.comm = gpu['comm'] if gpu['comm'] is not None else _dist.LocalComm()
        self._sample_injector = None        # test hook: f(data, stan_params) -> (S, d) draws
        self.last_site_stats = None         # sampler statistics of the last iteration (local sites)
        self.sampling_ms = []               # device time of every sampling launch (this rank)
        self.ngrad_log = []                 # gradient evaluations of every sampling launch (this rank)
        self.pass_log = []                  # passes over the site rows of every sampling launch (per site)
        self.team_pass_log = []             # layout 7: passes the row team made per launch (all sites), else None
        self.sweep_log = []                 # results of `run(..., sweep=...)`, one dict per iteration
        self.df_log = []                    # damping factor accepted in every iteration
        self.othertime_log = []             # host time of every update phase (this rank)
        device = gpu['device']
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0')) if self.comm.world > 1 else 0

        # ---- the data and its partition into sites
        if X.ndim not in (1, 2):
            raise ValueError("Argument `X` should be one or two dimensional")
        if y.ndim != 1:
            raise ValueError("Argument `y` should be one dimensional")
        if y.shape[0] != X.shape[0]:
            raise ValueError("The shapes of `y` and `X` does not match")
        self.N = X.shape[0]
        self.D = X.shape[1] if X.ndim == 2 else None
        self.Nk, self.k_lim, self.k_ind, order = _partition(
            self.N, own['site_sizes'], own['site_ind_ord'], own['site_ind'])
        self.K = len(self.Nk)
        # sorted, C-contiguous copies for the device; the Workers below still receive slices of the
        # caller's arrays, as in the reference (method.py:829-830)
        self.X = np.ascontiguousarray(X if order is None else X[order])
        self.y = np.ascontiguousarray(y if order is None else y[order])
        self.A, self.A_n, self.A_k = _additional_data(own['A'], own['A_n'], own['A_k'], self.N, self.K)
        self.Q0, self.r0, self.dphi = _prior_natural(own['prior'], own['dphi'], device)
        self.df_decay = own['df_decay']
        self.df_treshold = own['df_treshold']
        self.df0 = _damping_schedule(own['df0'], self.K)

        # ---- this rank's contiguous block of sites + its device engine
        self.k_lo, self.k_hi = _dist.site_range(self.K, self.comm.rank, self.comm.world)
        self.K_local = self.k_hi - self.k_lo
        if self.K_local < 1:
            raise ValueError("more ranks ({}) than sites ({})".format(self.comm.world, self.K))
        r0w, r1w = int(self.k_lim[self.k_lo]), int(self.k_lim[self.k_hi])
        k_lim_local = np.asarray(self.k_lim[self.k_lo:self.k_hi + 1], dtype=np.int64) - r0w
        groups = {}
        if not self.model_name.endswith('_sg'):
            # several groups per site (K < J, fit.py:310-324): the multi-group programs take the
            # number of groups `J` per site (A_k) and the 1-based group index `j_ind` per row (A_n)
            g_cnt, g_lim = self._site_groups()
            glo = int(np.sum(g_cnt[:self.k_lo]))
            ghi = glo + int(np.sum(g_cnt[self.k_lo:self.k_hi]))
            groups = dict(g_cnt=g_cnt[self.k_lo:self.k_hi], g_lim=g_lim[glo:ghi + 1] - r0w)
        factory = gpu['_engine_factory']
        if factory is None:
            self.engine = _engine.HipEngine(self.model_name, self.X[r0w:r1w], self.y[r0w:r1w],
                                            k_lim_local, device=device, **groups)
        else:
            self.engine = factory(self.model_name, self.X[r0w:r1w], self.y[r0w:r1w], k_lim_local, **groups)
        if self.engine.d != self.dphi:
            raise ValueError("Arg. `dphi`/`prior` ({}) does not match site model {} (dphi {})"
                             .format(self.dphi, self.model_name, self.engine.d))
        if hasattr(self.comm, 'bind'):
            self.comm.bind(self.engine)                   # RCCL communicator of the engine's context (collective)
        # the fused update (epx_update_trial) needs the engine's own communicator (or one rank)
        self._fused = bool(getattr(self.comm, 'native', False)) and hasattr(self.engine, 'update_trial')
        self._host_sums = None

        # ---- workers (method.py:817-834); slices of the ORIGINAL X, y like the reference
        self.workers = []
        for k in range(self.K):
            rows = slice(self.k_lim[k], self.k_lim[k + 1])
            A = dict((key, val[rows]) for (key, val) in self.A_n.items())
            A.update(self.A)
            A.update((key, val[k]) for (key, val) in self.A_k.items())
            w = Worker(k, self.site_model, self.dphi, X[rows], y[rows], A=A, _master=self, **self.worker_options)
            if self.k_lo <= k < self.k_hi:
                w._eng = self.engine
                w._k = k - self.k_lo
            self.workers.append(w)

        # ---- host mirrors (method.py:838-851)
        d, K = self.dphi, self.K
        self.S = np.empty((d, d), order='F')
        self.m = np.empty(d)
        self.Q = self.Q0.copy(order='F')
        self.r = self.r0.copy()
        self.Qi, self.Qi2, self.dQi = (np.zeros((d, d, K), order='F') for _ in range(3))
        self.ri, self.ri2, self.dri = (np.zeros((d, K), order='F') for _ in range(3))
        init_site = own['init_site']
        if isinstance(init_site, np.ndarray):
            self.Qi[...] = init_site[:, :, None]
        elif init_site is not None:
            self.Qi[np.arange(d), np.arange(d), :] = K / (init_site**2)
        self.iter = 0

        # ---- initial global approximation and cavities on the device (method.py:867-882)
        self.engine.set_prior(self.Q0, self.r0)
        self._upload_sites()
        g_pd, c_pd = self._trial(0.0, True)[:2]
        if not g_pd:
            raise ValueError("Initial approximation is not pos.def.")
        if not c_pd:
            raise ValueError("Initial cavity is not pos.def.")
        self.Q[...], self.r[...] = self.engine.get_global()
        for w in self.workers[self.k_lo:self.k_hi]:
            w.Q, w.r = self.Q, self.r
            w.phase = 1
            w._stale = True

    # ------------------------------------------------------------------
    def _trial(self, df, reduce_sums, stat_sum=(), stat_max=(), want_moments=False):
        """One damping trial over all ranks: (global_pd, cav_pd, first_bad (global site or -1),
        stat_sum, stat_max, S, m).  The device engine runs it as one stream-ordered batch with the
        all-reduce inside (epx_update_trial); other engines / transports compose it from the
        engine's primitives and the communicator's host collectives."""
        eng, comm = self.engine, self.comm
        if self._fused:
            return eng.update_trial(df, reduce_sums, self.k_lo, stat_sum, stat_max, want_moments)
        ss = np.array(stat_sum, dtype=np.float64)
        sm = np.array(stat_max, dtype=np.float64)
        if reduce_sums:
            self._host_sums = comm.allreduce_sum(np.array(eng.site_sums(), dtype=np.float64))
            if ss.size:
                ss = comm.allreduce_sum(ss)
            if sm.size:
                sm = comm.allreduce_max(sm)
        g_pd, c_pd, first_bad = eng.damped_trial(df, self._host_sums)
        c_pd = bool(g_pd and c_pd)
        first = self.k_lo + first_bad if (g_pd and first_bad >= 0) else self.K + 1
        if comm.world > 1:
            c_pd = bool(comm.allreduce_min_int(1 if c_pd else 0))
            first = comm.allreduce_min_int(first)
        S = m = None
        if want_moments and g_pd and c_pd:
            S, m = eng.global_moments()
        return g_pd, c_pd, (first if first <= self.K else -1), ss, sm, S, m

    def _upload_sites(self):
        lo, hi = self.k_lo, self.k_hi
        self.engine.set_sites(_engine.QI, np.asfortranarray(self.Qi[:, :, lo:hi]),
                              np.asfortranarray(self.ri[:, lo:hi]))
        self.engine.set_sites(_engine.DQI, np.asfortranarray(self.dQi[:, :, lo:hi]),
                              np.asfortranarray(self.dri[:, lo:hi]))

    def _download_sites(self):
        lo, hi = self.k_lo, self.k_hi
        for which, (Qa, ra) in ((_engine.QI, (self.Qi, self.ri)),
                                (_engine.QI2, (self.Qi2, self.ri2)),
                                (_engine.DQI, (self.dQi, self.dri))):
            Ql, rl = self.engine.get_sites(which)
            if self.comm.world > 1 and self.sync_sites:
                Qa[...] = self.comm.allgather_sites(Ql, self.K)
                ra[...] = self.comm.allgather_sites(rl, self.K)
            else:
                Qa[:, :, lo:hi] = Ql
                ra[:, lo:hi] = rl
        self.Q[...], self.r[...] = self.engine.get_global()

    def mix_phi(self, out_S=None, out_m=None):
        """Posterior approximation of phi from the pooled tilted samples of the last iteration
        (method.py:1250-1296): mean of the site means and the pooled covariance
        `(sum_k scatter_k + sum_k n_k (m_k - m)(m_k - m)') / (n_tot - 1)`.  The reference walks the
        workers' saved samples; here the per-site tilted means / scatters are already on the
        device, so three sums over the sites (and one all-reduce) give the same numbers."""
        if self.iter == 0:
            raise RuntimeError("Can not mix samples before at least one iteration has been done.")
        d, K = self.dphi, self.K
        sums = self.comm.allreduce_sum(self.engine.mix_sums())
        n = self.engine.get_tilted(0)[2]                # draws per site (the same for every site)
        sS = sums[:d * d].reshape(d, d, order='F')
        sm = sums[d * d:d * d + d]
        smm = sums[d * d + d:].reshape(d, d, order='F')
        m = sm / K
        S = (sS + n * (smm - K * np.outer(m, m))) / (n * K - 1)
        if out_S is None:
            out_S = np.zeros((d, d), order='F')
        if out_m is None:
            out_m = np.zeros(d)
        out_S[...] = S
        out_m[...] = m
        return out_S, out_m

    # a site whose slowest chain took more than this fraction of the iteration's slowest chain is
    # scheduled one workgroup per chain next time (measured leapfrog: 5.6 vs 10.3 us at D=32, n=500)
    LEAD_FRACTION = 0.4

    @classmethod
    def _site_schedule(cls, passes, chain_leapfrogs, n_cu=256):
        """Dispatch order and number of lead sites for the next sampling launch from the work of
        this one.  A launch ends with its slowest chain, and which sites hold the slow chains is
        stable from one EP iteration to the next (a property of the site's posterior geometry:
        measured rank correlation 0.96-0.98 at C3), so the sites within LEAD_FRACTION of the
        slowest come first, ordered by their slowest chain, and run at the shorter leapfrog of
        one workgroup per chain (engine.set_site_split); the others follow longest-first."""
        chain_leapfrogs = np.asarray(chain_leapfrogs)
        site_max = chain_leapfrogs.max(axis=1)
        lead = np.where(site_max > cls.LEAD_FRACTION * site_max.max())[0]
        # no tail to speak of: many sites near the slowest, or enough work to keep every CU (one
        # site, i.e. up to 4 chains, each) busy for most of the slowest chain's run anyway
        busy = site_max.sum() / n_cu
        if lead.size * 4 > site_max.size or busy > 0.7 * site_max.max():
            lead = lead[:0]
        lead = lead[np.argsort(-site_max[lead], kind='stable')]
        rest = np.setdiff1d(np.arange(site_max.size), lead)
        rest = rest[np.argsort(-np.asarray(passes)[rest], kind='stable')]
        return np.concatenate((lead, rest)).astype(np.int32), int(lead.size)

    PIECES_PER_SITE = int(os.environ.get('EPX_PIECES_PER_SITE', '16'))     # (the environment variable: A/B runs only)

    def _one_workgroup_per_cu(self):
        """The resident sampler keeps a site's rows (padded to 16 / 32 columns) in LDS: above half of the 160 KB
        only one workgroup fits a CU; the streaming sampler (D > 32, or rows beyond the LDS) always fills it.  Then
        a pieced launch (one resident workgroup per CU, looping over pieces) loses nothing to occupancy."""
        n_max = int(np.max(np.diff(self.k_lim[self.k_lo:self.k_hi + 1])))
        if self.D > 32:
            return True
        return n_max * (16 if self.D <= 16 else 32) * 8 > 80 * 1024

    def _site_groups(self):
        """Group structure of the sites from `A_k['J']` and `A_n['j_ind']` (the data the
        reference hands to m*b.stan): groups per site and the row limits of all groups.  The
        rows of a group have to be contiguous and the groups of a site in order -- what
        `util.distribute_groups` produces."""
        if 'J' not in self.A_k or 'j_ind' not in self.A_n:
            raise ValueError("site model {!r} holds several groups per site: give the number of groups per "
                             "site as A_k['J'] and the 1-based group index of every row as A_n['j_ind'] "
                             "(fit.py:310-324), or use the single-group model {!r}"
                             .format(self.model_name, self.model_name + '_sg'))
        g_cnt = np.asarray(self.A_k['J'], dtype=np.int32)
        j_ind = np.asarray(self.A_n['j_ind'])
        lims = [0]
        for k in range(self.K):
            jk = j_ind[self.k_lim[k]:self.k_lim[k + 1]]
            if jk[0] != 1 or jk[-1] != g_cnt[k] or np.any(np.diff(jk) < 0) or np.any(np.diff(jk) > 1):
                raise ValueError("A_n['j_ind'] of site {}: the rows of a group have to be contiguous and the "
                                 "groups numbered 1..J in order".format(k))
            change = np.nonzero(np.diff(jk))[0] + 1
            lims.extend((self.k_lim[k] + change).tolist())
            lims.append(int(self.k_lim[k + 1]))
        return g_cnt, np.asarray(lims, dtype=np.int64)

    def _score_damps(self, damps, m_target, S_target, samp_target):
        """The sweep itself, on the site sums the device / `_host_sums` hold."""
        eng, comm = self.engine, self.comm
        res = eng.damp_sweep(damps, None if self._fused else self._host_sums, m_target, S_target, samp_target)
        if comm.world > 1:
            res[:, 1] = -comm.allreduce_max(-res[:, 1])
        bad = res[:, 1] == 0.0
        res[bad, 2:] = np.nan
        return dict(damps=np.asarray(damps, dtype=np.float64), global_pd=res[:, 0] > 0, cav_pd=res[:, 1] > 0,
                    mses=res[:, 2], kls=res[:, 3], lls=res[:, 4])

    def damp_sweep(self, damps, m_target, S_target, samp_target=None):
        """Score damping factors for the pending site updates `dQi, dri` against a target
        posterior N(m_target, S_target): the loop `for di, df in enumerate(damps)` of
        experiment/find_damp.py:146-173 as ONE batched device call (the proposal is affine in
        `df`, so the site sums are reduced once).  Returns a dict with `mses`, `lls`, `kls`
        (NaN where the proposal or a cavity is not positive definite, as in the reference) and
        the flags `global_pd`, `cav_pd`.  Site parameters, the global approximation and the
        cavities are those of the accepted state again when it returns."""
        self._trial(0.0, True)                  # reduce the sums of the current (Qi, dQi)
        out = self._score_damps(damps, m_target, S_target, samp_target)
        self._trial(0.0, False)                 # the sweep left the last factor's Q, r and cavities behind
        return out

    def cur_approx(self):
        """Current posterior approximation moments (S, m) (method.py:884-896)."""
        return self.engine.invert_normal_params(self.Q, self.r)

    def _ret(self, info, calc_moments, return_analytics, moments, analytics, as_tuple=False):
        out = [info]
        if calc_moments:
            out.append(moments)
        if return_analytics:
            out.append(analytics)
        if len(out) == 1:
            return out[0]
        return tuple(out) if as_tuple else out

    def run(self, niter, calc_moments=True, save_last_param=None, verbose=True,
            return_analytics=False, seed=None, sweep=None):
        """Run the distributed EP algorithm (method.py:899-1247).

        Returns `info`, optionally followed by `(m_phi_s, cov_phi_s)` and
        `(stimes, msteps, mrhats, othertimes)` exactly like the reference
        (a list on the early-exit paths, a tuple on the normal path).

        `save_last_param`: parameter names whose draws of the LAST iteration every worker keeps
        in `saved_samp` (method.py:1011-1016).

        `sweep` (not in the reference's `run`; it is the body of experiment/find_damp.py): a dict
        `damps, m_target, S_target[, samp_target]`; every iteration scores these damping factors
        before its own damped update and appends the result to `self.sweep_log`."""
        if niter < 1:
            if verbose:
                print("Nothing to do here as provided arg. `niter` is {}".format(niter))
            return self._ret(self.INFO_OK, calc_moments, return_analytics,
                             (None, None), (None, None, None))

        seeds = run_seeds(seed, niter, self.K)                    # :956-960
        eng = self.engine
        lo, hi, K = self.k_lo, self.k_hi, self.K
        local_workers = self.workers[lo:hi]

        # host mirrors -> device (the caller may have edited them between runs)
        self._upload_sites()
        eng.set_global(self.Q, self.r)

        m_phi_s = cov_phi_s = None
        if calc_moments:
            m_phi_s = np.zeros((niter, self.dphi))
            cov_phi_s = np.zeros((niter, self.dphi, self.dphi))
        stimes = np.zeros(niter)
        msteps = np.zeros(niter)
        mrhats = np.zeros(niter)
        othertimes = np.zeros(niter)
        moments = (m_phi_s, cov_phi_s)
        analytics = (stimes, msteps, mrhats, othertimes)

        for cur_iter in range(niter):
            self.iter += 1
            if verbose:
                print("Iter {} starting. Process tilted distributions".format(self.iter))

            # ---- tilted distributions: ONE batched launch over this rank's sites (:1005-1023)
            w0 = local_workers[0]
            for w in local_workers:
                if w.phase != 1:
                    raise RuntimeError('Cavity has to be calculated before tilted.')
            sseeds = stan_seeds(seeds[cur_iter, lo:hi])             # :342-346
            estim = w0._cur_estim()
            if self._sample_injector is not None:
                posdefs_l, tl, ml, rl = self._tilted_injected(sseeds, estim)
            else:
                opts = w0._sampler_opts()
                if (self.balance_sites and not self.sampling_ms and hasattr(eng, 'set_piece_queue')
                        and self.K_local > eng.cu_count() and self._one_workgroup_per_cu()):
                    # no history yet: the piece queue with equal predicted work per transition (the library ignores
                    # it when the launch does not run the row-wave / state-wave kernel)
                    eng.set_piece_queue(max(1, w0.stan_params['iter'] // self.PIECES_PER_SITE), None)
                posdefs_l, stats, ms = eng.tilted_batch(sseeds, opts, estim)
                for j in np.nonzero(stats[:, 7] > 0)[0]:
                    posdefs_l[j] = local_workers[j]._void_update()
                self.last_site_stats = stats        # (K_local, 8): see epx_site_stat in include/epx.h
                self.sampling_ms.append(ms)
                self.ngrad_log.append(float(stats[:, 3].sum()))
                self.pass_log.append(eng.row_passes(w0.stan_params['chains']))
                # (layout 7: what the row team really did, yielded passes included -- a measurement aid, bench.py)
                self.team_pass_log.append(float(eng.team_passes().sum()) if eng.last_layout() == 7 else None)
                if self.balance_sites and self.K_local > 1:
                    # longest-first dispatch of the next iteration's workgroups (results unaffected)
                    order, n_lead = self._site_schedule(self.pass_log[-1],
                                                        eng.get_chain_stats(w0.stan_params['chains'])[:, :, 3],
                                                        eng.cu_count())
                    eng.set_site_order(order)
                    eng.set_site_split(n_lead)
                    if hasattr(eng, 'set_piece_queue'):
                        # when a site fills the LDS (one workgroup per CU) and there are more sites than CUs: run the
                        # sampler from a piece queue -- looping workgroups claim pieces of a site's transitions, the site with
                        # the largest predicted remaining work first (same draws; only the dispatch changes)
                        if (eng.last_layout() in (5, 7, 3) and n_lead == 0 and self.K_local > eng.cu_count()
                                and self._one_workgroup_per_cu()):
                            it_s = w0.stan_params['iter']
                            lf = eng.get_chain_stats(w0.stan_params['chains'])[:, :, 3]
                            eng.set_piece_queue(max(1, it_s // self.PIECES_PER_SITE), np.maximum(lf.max(axis=1), 1.0) / it_s)
                        else:
                            eng.set_piece_queue(0)
                tl = np.full(self.K_local, ms * 1e-3)
                ml, rl = stats[:, 0], stats[:, 1]
            nsamp = eng.get_tilted(0)[2]
            for j, w in enumerate(local_workers):
                w.stan_params['seed'] = int(sseeds[j])
                w.last_time, w.last_msteps, w.last_mrhat = tl[j], ml[j], rl[j]
                if w.init_prev:
                    w.stan_params['init'] = 'prev' if self._sample_injector is None \
                        else [{}] * w.stan_params['chains']
                w.nsamp = nsamp
                if w.prec_estim_skip > 0:
                    w.prec_estim_skip -= 1
                w.phase = 2 if posdefs_l[j] else 0
                if not posdefs_l[j] and w.init_prev:
                    w.init = w.init_orig                            # sic (:468)
                w._stale = True
                w.iteration += 1
                if save_last_param and cur_iter == niter - 1 and self._sample_injector is None:
                    w._save_named(save_last_param)
            n_ok = int(np.sum(posdefs_l))
            start_othertime = time.time()

            # ---- the one reduction per iteration (:1073-1074, affine in df) and the first damping
            # trial behind it; the site counts and the analytics of :1043-1045 ride on the same buffer
            df = self.df0(self.iter)                                # :1060
            g_pd, c_pd, first_bad, ssum, smax, S, m = self._trial(
                df, True, stat_sum=(n_ok, self.K_local - n_ok),
                stat_max=(np.max(tl), np.max(ml), np.max(rl)), want_moments=calc_moments)
            if verbose:
                if ssum[1] == 0.0:
                    print("\rAll sites ok")
                elif ssum[0] > 0:
                    print("\rSome sites failed and are not updated")
                else:
                    print("\rEvery site failed")
            if ssum[0] == 0.0:                                      # :1033-1040
                self._download_sites()
                return self._ret(self.INFO_ALL_SITES_FAIL, calc_moments, return_analytics,
                                 moments, analytics)
            stimes[cur_iter], msteps[cur_iter], mrhats[cur_iter] = smax[0], smax[1], smax[2]   # :1043-1045
            if verbose:
                print("Sampling done, max sampling time {}".format(stimes[cur_iter]))

            if sweep is not None:                                   # find_damp.py:146-173
                self.sweep_log.append(self._score_damps(sweep['damps'], sweep['m_target'], sweep['S_target'],
                                                        sweep.get('samp_target')))
                g_pd, c_pd, first_bad, _, _, S, m = self._trial(df, False, want_moments=calc_moments)

            if verbose:
                print("Iter {}, starting df {:.3g}".format(self.iter, df))
            fail_printline = False
            failed_force_pos_def = False
            while not (g_pd and c_pd):                              # :1067
                df *= self.df_decay                                 # :1083 / :1163
                if verbose:
                    fail_printline = True
                    if not g_pd:
                        sys.stdout.write("\rNon pos. def. posterior cov, " +
                                         "reducing df to {:.3}".format(df) + " "*5 + "\b"*5)
                    else:
                        sys.stdout.write("\rNon pos. def. cavity, " +
                                         "(first encountered in site {}), ".format(first_bad) +
                                         "reducing df to {:.3}".format(df) + " "*5 + "\b"*5)
                    sys.stdout.flush()
                if not g_pd and self.iter == 1:                     # :1092-1101
                    if verbose:
                        print("\nInvalid prior.")
                    self._download_sites()
                    return self._ret(self.INFO_INVALID_PRIOR, calc_moments, return_analytics,
                                     moments, analytics)
                refresh = False
                if df < self.df_treshold:                           # :1102-1132 / :1177-1207
                    if verbose:
                        print("\nDamping factor reached minimum.")
                    df = self.df0(self.iter)
                    if failed_force_pos_def:
                        if verbose:
                            print("Failed to force pos_def.")
                        self._download_sites()
                        return self._ret(self.INFO_DF_TRESHOLD_REACHED_CAVITY, calc_moments,
                                         return_analytics, moments, analytics)
                    failed_force_pos_def = True
                    forced = eng.force_pd(df, self.MIN_EIG_TRESHOLD, self.MIN_EIG)
                    if verbose:
                        print("Force sites {} pos_def.".format(np.nonzero(forced)[0] + lo))
                    refresh = True                                  # the shift changed Qi: reduce the sums again
                g_pd, c_pd, first_bad, _, _, S, m = self._trial(df, refresh, want_moments=calc_moments)
            if verbose and fail_printline:
                print()

            eng.accept(df)                                          # :1145-1158
            self.df_log.append(df)
            for w in local_workers:
                w.Q, w.r = self.Q, self.r
                w.phase = 1
                w._stale = True

            if calc_moments:                                        # :1211-1219
                self.S[...] = S
                self.m[...] = m
                np.copyto(m_phi_s[cur_iter], m)
                np.copyto(cov_phi_s[cur_iter], S.T)
                if verbose:
                    print("Mean and std of phi[0]: {:.3}, {:.3}".format(
                        m_phi_s[cur_iter, 0], np.sqrt(cov_phi_s[cur_iter, 0, 0])))
            othertimes[cur_iter] = time.time() - start_othertime   # :1230
            self.othertime_log.append(othertimes[cur_iter])
            if verbose:
                print("Iter {} done.".format(self.iter))

        self._download_sites()
        if verbose:
            print("{} iterations done\nTotal limiting sampling time: {}"
                  .format(niter, stimes.sum()))
        return self._ret(self.INFO_OK, calc_moments, return_analytics, moments, analytics,
                         as_tuple=True)

    # ------------------------------------------------------------------
    def _tilted_injected(self, sseeds, estim):
        """Test hook: draws come from `self._sample_injector(data, stan_params)`
        (same role as the reference's `_sample_stan`, method.py:43) and go
        through the device moment kernel."""
        eng = self.engine
        samples = None
        for j, w in enumerate(self.workers[self.k_lo:self.k_hi]):
            w.stan_params['seed'] = int(sseeds[j])
            w._refresh_for_injection()
            samp = np.asarray(self._sample_injector(w.data, w.stan_params), dtype=np.float64)
            if samples is None:
                samples = np.empty((samp.shape[0], samp.shape[1], self.K_local), order='F')
            samples[:, :, j] = samp
        posdefs = eng.moments_batch(samples, estim)
        n = self.K_local
        return posdefs, np.full(n, 0.25), np.full(n, 0.125), np.full(n, 1.0625)
