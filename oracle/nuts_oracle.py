"""ctypes loader for oracle/libepx_oracle.so (nuts_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libepx_oracle.so')

MODEL_IDS = {'m1b_sg': 0, 'm2b_sg': 1, 'm3b_sg': 2, 'm4b_sg': 3, 'm5b_sg': 4,
             # the multi-group programs (K < J, experiment/models/m*b.stan): same densities, several
             # (eta, etb) blocks per site
             'm1b': 0, 'm2b': 1, 'm3b': 2, 'm4b': 3, 'm5b': 4,
             # Gaussian-likelihood family (experiment/models/m*a_sg.stan): phi = [log sigma | b-model phi],
             # real-valued responses
             'm1a_sg': 5, 'm2a_sg': 6, 'm3a_sg': 7, 'm4a_sg': 8, 'm5a_sg': 9,
             'm1a': 5, 'm2a': 6, 'm3a': 7, 'm4a': 8, 'm5a': 9}
STAT_NAMES = ('stepsize_mean', 'stepsize_final', 'nleap', 'ngrad', 'ndiv',
              'accept_mean', 'depth_mean', 'fail')
_lib = None


def build(force=False):
    src = os.path.join(HERE, 'nuts_oracle.c')
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', HERE, 'libepx_oracle.so'])
    return LIB


def _restypes(L):
    L.epo_dphi.restype = ctypes.c_int
    L.epo_npar.restype = ctypes.c_int
    L.epo_npar_groups.restype = ctypes.c_int
    L.epo_num_threads.restype = ctypes.c_int
    return L


def lib():
    """The library the wrappers below call: the STRICT build (-O2, no contraction: the checker), unless a caller has
    switched to the fast build for a timing (`timing_build`)."""
    global _lib
    if _fast_on:
        return fast_lib()
    if _lib is None:
        build()
        _lib = _restypes(ctypes.CDLL(LIB))
    return _lib


_fast = None
_fast_on = False


def fast_lib():
    """The same source built -O3 -march=native with contraction allowed, ON this machine (the file name carries a hash
    of the CPU model: the GPU box's host is not this container's).  For TIMING only: bench.py's cpu_baseline."""
    global _fast
    if _fast is None:
        import hashlib
        cpu = 'unknown'
        try:
            for line in open('/proc/cpuinfo'):
                if line.startswith('model name') or line.startswith('flags'):
                    cpu += line
                    if line.startswith('flags'):
                        break
        except OSError:
            pass
        name = 'libepx_oracle_fast_%s.so' % hashlib.sha1(cpu.encode()).hexdigest()[:10]
        path = os.path.join(HERE, name)
        src = os.path.join(HERE, 'nuts_oracle.c')
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(['make', '-s', '-C', HERE, name, 'FAST_LIB=' + name])
        _fast = _restypes(ctypes.CDLL(path))
    return _fast


class timing_build:
    """`with timing_build():` -- the wrappers of this module call the fast build inside the block."""

    def __enter__(self):
        global _fast_on
        fast_lib()
        self.was, _fast_on = _fast_on, True
        return self

    def __exit__(self, *exc):
        global _fast_on
        _fast_on = self.was
        return False


def _p(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def is_gauss(model):
    return MODEL_IDS[model] >= 5


def _y(model, y):
    """Responses in the type the library expects for the model: int32 0/1, or float64 for the
    Gaussian family (nuts_oracle.c reads `y` as doubles for model ids 5..9)."""
    if is_gauss(model):
        return np.ascontiguousarray(y, dtype=np.float64)
    return np.ascontiguousarray(y, dtype=np.int32)


def _yp(y):
    return ctypes.cast(y.ctypes.data, ctypes.POINTER(ctypes.c_int32))


def dims(model, D, ng=1):
    """(dphi, sampled coordinates) of a site with ng groups."""
    L = lib()
    m = MODEL_IDS[model]
    return L.epo_dphi(m, D), L.epo_npar_groups(m, D, int(ng))


def _groups(k_lim, g_cnt, g_lim):
    """Normalise the group structure: g_cnt (K) groups per site, g_lim (sum+1) absolute row limits."""
    if g_cnt is None:
        return None, None, None
    g_cnt = np.ascontiguousarray(g_cnt, dtype=np.int32)
    g_lim = np.ascontiguousarray(g_lim, dtype=np.int64)
    assert g_lim.shape[0] == g_cnt.sum() + 1
    off = np.concatenate(([0], np.cumsum(g_cnt)))
    assert np.array_equal(g_lim[off], np.asarray(k_lim)), 'group limits must nest in the site limits'
    return g_cnt, g_lim, off


def logdensity_grad(model, X, y, mu, Omega, theta, gl=None):
    """gl: row limits of the site's groups relative to its first row (ng+1 entries), or None."""
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = _y(model, y)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    Om = np.ascontiguousarray(Omega, dtype=np.float64)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    n, D = X.shape
    lp = ctypes.c_double()
    g = np.zeros(th.shape[0])
    if gl is None:
        ng, glp = 1, None
    else:
        gl = np.ascontiguousarray(gl, dtype=np.int64)
        ng, glp = gl.shape[0] - 1, _p(gl, ctypes.c_int64)
    rc = L.epo_logdensity_grad_groups(MODEL_IDS[model], n, D, ng, glp, _p(X), _yp(y),
                                      _p(mu), _p(Om), _p(th), ctypes.byref(lp), _p(g))
    assert rc == 0
    return lp.value, g


def carry_history(draws, stats):
    """The `adapt = carry` history the device library keeps after a sampling call (k_carry_update,
    csrc/nuts.hip): final step size per chain (-1 for a site with a failed chain) and the site's
    pooled variances, regularised like Stan's windowed estimate.  draws (K, chains, nkeep, P)."""
    K, C, nk, P = draws.shape
    n = C * nk
    eps = stats[:, :, 1].copy()
    flat = draws.reshape(K, n, P)
    var = flat.var(axis=1, ddof=1)
    metric = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0))
    bad = stats[:, :, 7].sum(axis=1) > 0
    eps[bad] = -1.0
    return eps, metric


def nuts_sites(model, X, y, k_lim, mu, Omega, seeds, chains=4, iter=200, warmup=None,
               thin=1, max_depth=10, init=None, nthreads=0, g_cnt=None, g_lim=None,
               carry_eps=None, carry_metric=None, trace_sites=0, dump_at=None):
    """Sample every site; returns (draws (K,chains,nkeep,P), last (K,chains,P),
    stats (K,chains,8)). mu (K,d), Omega (K,d,d) symmetric.  With groups (g_cnt, g_lim) P is the
    largest coordinate count over the sites and shorter sites are zero padded.
    trace_sites > 0: a fourth result, the per-transition trace of the first sites (sites, chains, iter, 8 + P), warm-up
    included: [eps used, leapfrogs, accept, depth, divergent, eps after adaptation, sum of the metric, log density,
    sample] -- the counterpart of the device library's epx_set_trace / epx_get_trace.
    dump_at (with trace_sites): transition indices; a fifth result, the chains' state BEFORE each of them,
    (sites, chains, len(dump_at), 20 + 4 P): the 20 scalars of the device's checkpoint record (csrc/epx_pieces.h
    EPX_CK_LIST order), then sample, Welford mean, Welford sum of squares, metric."""
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = _y(model, y)
    k_lim = np.ascontiguousarray(k_lim, dtype=np.int64)
    K = k_lim.shape[0] - 1
    D = X.shape[1]
    g_cnt, g_lim, _ = _groups(k_lim, g_cnt, g_lim)
    d, P = dims(model, D, 1 if g_cnt is None else int(g_cnt.max()))
    mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(K, d)
    Om = np.ascontiguousarray(Omega, dtype=np.float64).reshape(K, d, d)
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    if warmup is None:
        warmup = iter // 2
    nkeep = (iter - warmup + thin - 1) // thin
    draws = np.zeros((K, chains, nkeep, P))
    last = np.zeros((K, chains, P))
    stats = np.zeros((K, chains, 8))
    ip = None
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.float64).reshape(K, chains, P)
        ip = _p(init)
    ce = cm = None
    if carry_eps is not None:
        carry_eps = np.ascontiguousarray(carry_eps, dtype=np.float64).reshape(K, chains)
        carry_metric = np.ascontiguousarray(carry_metric, dtype=np.float64).reshape(K, P)
        ce, cm = _p(carry_eps), _p(carry_metric)
    trace = dump = None
    if trace_sites > 0:
        trace = np.zeros((min(int(trace_sites), K), chains, iter, 8 + P))
        L.epo_set_trace(_p(trace), trace.shape[0])
        if dump_at is not None:
            dts = np.ascontiguousarray(dump_at, dtype=np.int32)
            dump = np.zeros((trace.shape[0], chains, dts.shape[0], 20 + 4 * P))
            L.epo_set_dump(_p(dts, ctypes.c_int32), int(dts.shape[0]), _p(dump), trace.shape[0])
    try:
        rc = _nuts_sites_call(L, model, K, D, k_lim, g_cnt, g_lim, X, y, mu, Om, seeds, chains, iter, warmup, thin, max_depth,
                              ip, draws, last, stats, nthreads, ce, cm)
    finally:
        if trace is not None:
            L.epo_set_trace(None, 0)
            L.epo_set_dump(None, 0, None, 0)
    if rc != 0:
        raise ValueError('epo_nuts_sites rc=%d' % rc)
    if dump is not None:
        return draws, last, stats, trace, dump
    if trace is not None:
        return draws, last, stats, trace
    return draws, last, stats


def _nuts_sites_call(L, model, K, D, k_lim, g_cnt, g_lim, X, y, mu, Om, seeds, chains, iter, warmup, thin, max_depth,
                     ip, draws, last, stats, nthreads, ce, cm):
    return L.epo_nuts_sites_carry(MODEL_IDS[model], K, D, _p(k_lim, ctypes.c_int64),
                                None if g_cnt is None else _p(g_cnt, ctypes.c_int32),
                                None if g_cnt is None else _p(g_lim, ctypes.c_int64), _p(X),
                                _yp(y), _p(mu), _p(Om), _p(seeds, ctypes.c_int64),
                                  chains, iter, warmup, thin, max_depth, ip, _p(draws), _p(last),
                                  _p(stats), nthreads, ce, cm)


def nuts_transitions(model, X, y, k_lim, mu, Omega, seeds, q0, eps, inv_e, nt=1, t_offset=0,
                     max_depth=10, g_cnt=None, g_lim=None):
    """TEST HOOK: nt un-adapted transitions per (site, chain) from q0 (K,chains,P)
    with step sizes eps (K,chains) and inverse metrics inv_e (K,chains,P)."""
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = _y(model, y)
    k_lim = np.ascontiguousarray(k_lim, dtype=np.int64)
    K = k_lim.shape[0] - 1
    D = X.shape[1]
    g_cnt, g_lim, _ = _groups(k_lim, g_cnt, g_lim)
    d, P = dims(model, D, 1 if g_cnt is None else int(g_cnt.max()))
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    chains = q0.shape[1]
    mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(K, d)
    Om = np.ascontiguousarray(Omega, dtype=np.float64).reshape(K, d, d)
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    eps = np.ascontiguousarray(eps, dtype=np.float64).reshape(K, chains)
    inv_e = np.ascontiguousarray(inv_e, dtype=np.float64).reshape(K, chains, P)
    draws = np.zeros((K, chains, nt, P))
    last = np.zeros((K, chains, P))
    stats = np.zeros((K, chains, 8))
    rc = L.epo_nuts_transitions_groups(MODEL_IDS[model], K, D, _p(k_lim, ctypes.c_int64),
                                None if g_cnt is None else _p(g_cnt, ctypes.c_int32),
                                None if g_cnt is None else _p(g_lim, ctypes.c_int64), _p(X),
                                _yp(y), _p(mu), _p(Om), _p(seeds, ctypes.c_int64),
                                chains, nt, t_offset, max_depth, _p(q0), _p(eps), _p(inv_e),
                                _p(draws), _p(last), _p(stats))
    if rc != 0:
        raise ValueError('epo_nuts_transitions rc=%d' % rc)
    return draws, stats


def rng_probe(seed, chain, t, kind, a, b):
    L = lib()
    out = [ctypes.c_double() for _ in range(4)]
    L.epo_rng_probe(ctypes.c_uint64(seed), chain, ctypes.c_uint32(t), ctypes.c_uint32(kind),
                    ctypes.c_uint32(a), ctypes.c_uint32(b), *[ctypes.byref(o) for o in out])
    return tuple(o.value for o in out)


def split_rhat(x):
    """PyStan 2.17 split R-hat of one scalar quantity, x: (chains, n)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[1] - (x.shape[1] % 2)
    h = n // 2
    halves = np.concatenate([x[:, :h], x[:, x.shape[1] - h:]], axis=0)
    means = halves.mean(axis=1)
    vars_ = halves.var(axis=1, ddof=1)
    var_between = h * means.var(ddof=1)
    var_within = vars_.mean()
    return np.sqrt((var_between / var_within + h - 1) / h)
