"""Device engine: one `epx_ctx` (one rank's sites on one MI355X) behind ctypes.

`HipEngine` is the only engine the package ships.  It raises when libepx.so or
a HIP device is missing -- there is no CPU path.  (Tests drive `Master`'s host
logic on CPU with an oracle-backed engine, oracle/engine_oracle.py, which only tests import.)
"""

import ctypes

import numpy as np

from . import _lib
from ._lib import SamplerOpts, check, dptr

MODEL_IDS = {'m1b_sg': 0, 'm2b_sg': 1, 'm3b_sg': 2, 'm4b_sg': 3, 'm5b_sg': 4,
             # the multi-group programs experiment/models/m{1..5}b.stan (K < J): same densities with
             # several (eta, etb) blocks per site; they need the group structure of the sites
             'm1b': 0, 'm2b': 1, 'm3b': 2, 'm4b': 3, 'm5b': 4,
             # Gaussian-likelihood family experiment/models/m{1..5}a_sg.stan: phi = [log sigma | the
             # b-model's phi], real responses
             'm1a_sg': 5, 'm2a_sg': 6, 'm3a_sg': 7, 'm4a_sg': 8, 'm5a_sg': 9,
             'm1a': 5, 'm2a': 6, 'm3a': 7, 'm4a': 8, 'm5a': 9}


def is_gauss(model):
    return MODEL_IDS[model] >= 5
PREC_ESTIM_IDS = {'sample': 0, 'olse': 1}
QI, QI2, DQI = 0, 1, 2
INIT_IDS = {'random': 0, '0': 1, 0: 1, 'prev': 2}
N_STAT = 8


def model_dims(model, D):
    """(dphi, P) of a single-group logistic model (m*b_sg.stan)."""
    lib = _lib.load()
    d, P = ctypes.c_int(), ctypes.c_int()
    check(lib.epx_model_dims(MODEL_IDS[model], int(D), ctypes.byref(d), ctypes.byref(P)))
    return d.value, P.value


def _f64(a, order='F'):
    return np.require(a, dtype=np.float64, requirements=['F' if order == 'F' else 'C', 'A'])


class HipEngine(object):
    """Sites k = 0..K_local-1 of one rank, resident in HBM."""

    def __init__(self, model, X, y, k_lim, device=0, g_cnt=None, g_lim=None):
        """g_cnt (K) groups per site and g_lim (sum+1) row limits of all groups: sites that hold
        several groups of the hierarchical model (util.distribute_groups, K < J)."""
        self.lib = _lib.load()
        if model not in MODEL_IDS:
            raise ValueError('unknown site model {!r}; built-in models: {}'
                             .format(model, sorted(MODEL_IDS)))
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError('the built-in site models need a two dimensional X')
        gauss = is_gauss(model)
        y32 = None if gauss else np.ascontiguousarray(y, dtype=np.int32)
        yd = np.ascontiguousarray(y, dtype=np.float64) if gauss else None
        k_lim = np.ascontiguousarray(k_lim, dtype=np.int64)
        self.model = model
        self.K = k_lim.shape[0] - 1
        self.D = X.shape[1]
        self.d, self.P = model_dims(model if model.endswith('_sg') else model + '_sg', self.D)
        self.device = device
        self._trace_sites = 0                  # (set_trace: sites whose transitions are recorded; 0 = off)
        ctx = ctypes.c_void_p()
        if gauss and g_cnt is None:
            if not model.endswith('_sg'):
                raise ValueError('site model {!r} needs the groups of every site (g_cnt, g_lim)'.format(model))
            self.site_P = np.full(self.K, self.P, dtype=np.int64)
            check(self.lib.epx_ctx_create_real(
                device, MODEL_IDS[model], self.K, self.D,
                k_lim.ctypes.data_as(_lib.c_int64_p), dptr(X), dptr(yd), ctypes.byref(ctx)))
        elif g_cnt is None:
            if not model.endswith('_sg'):
                raise ValueError('site model {!r} needs the groups of every site (g_cnt, g_lim)'.format(model))
            check(self.lib.epx_ctx_create(
                device, MODEL_IDS[model], self.K, self.D,
                k_lim.ctypes.data_as(_lib.c_int64_p), dptr(X),
                y32.ctypes.data_as(_lib.c_int32_p), ctypes.byref(ctx)))
        else:
            g_cnt = np.ascontiguousarray(g_cnt, dtype=np.int32)
            g_lim = np.ascontiguousarray(g_lim, dtype=np.int64)
            if g_cnt.shape[0] != self.K or g_lim.shape[0] != int(g_cnt.sum()) + 1:
                raise ValueError('g_cnt needs one entry per site and g_lim sum(g_cnt)+1 entries')
            pg = 1 if MODEL_IDS[model] % 5 == 0 else 1 + self.D
            self.P = self.d + int(g_cnt.max()) * pg          # record stride: the largest site
            self.site_P = self.d + g_cnt.astype(np.int64) * pg
            if gauss:
                check(self.lib.epx_ctx_create_real_groups(
                    device, MODEL_IDS[model], self.K, self.D,
                    k_lim.ctypes.data_as(_lib.c_int64_p), g_cnt.ctypes.data_as(_lib.c_int32_p),
                    g_lim.ctypes.data_as(_lib.c_int64_p), dptr(X), dptr(yd), ctypes.byref(ctx)))
            else:
                check(self.lib.epx_ctx_create_groups(
                    device, MODEL_IDS[model], self.K, self.D,
                    k_lim.ctypes.data_as(_lib.c_int64_p), g_cnt.ctypes.data_as(_lib.c_int32_p),
                    g_lim.ctypes.data_as(_lib.c_int64_p), dptr(X),
                    y32.ctypes.data_as(_lib.c_int32_p), ctypes.byref(ctx)))
        self.g_cnt, self.g_lim = g_cnt, g_lim
        self.ctx = ctx
        n = ctypes.c_int()
        check(self.lib.epx_packed_len(self.ctx, ctypes.byref(n)))
        self.packed_len = n.value

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.epx_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state transfer
    def set_prior(self, Q0, r0):
        check(self.lib.epx_set_prior(self.ctx, dptr(_f64(Q0)), dptr(_f64(r0))))

    def set_sites(self, which, Q=None, r=None):
        Qa = _f64(Q) if Q is not None else None
        ra = _f64(r) if r is not None else None
        check(self.lib.epx_set_sites(self.ctx, which, dptr(Qa), dptr(ra)))

    def get_sites(self, which, Q=None, r=None):
        """Copy into the given F-ordered arrays (allocated when None)."""
        d, K = self.d, self.K
        if Q is None:
            Q = np.empty((d, d, K), order='F')
        if r is None:
            r = np.empty((d, K), order='F')
        assert Q.flags['F_CONTIGUOUS'] and r.flags['F_CONTIGUOUS']
        check(self.lib.epx_get_sites(self.ctx, which, dptr(Q), dptr(r)))
        return Q, r

    def set_site(self, which, k, Q=None, r=None):
        Qa = _f64(Q) if Q is not None else None
        ra = _f64(r) if r is not None else None
        check(self.lib.epx_set_site(self.ctx, which, int(k), dptr(Qa), dptr(ra)))

    def get_site(self, which, k):
        Q = np.empty((self.d, self.d), order='F')
        r = np.empty(self.d)
        check(self.lib.epx_get_site(self.ctx, which, int(k), dptr(Q), dptr(r)))
        return Q, r

    def set_global(self, Q, r):
        check(self.lib.epx_set_global(self.ctx, dptr(_f64(Q)), dptr(_f64(r))))

    def get_global(self):
        Q = np.empty((self.d, self.d), order='F')
        r = np.empty(self.d)
        check(self.lib.epx_get_global(self.ctx, dptr(Q), dptr(r)))
        return Q, r

    # ---- cavity
    def cavity_batch(self, which, k0=0, count=None):
        count = self.K - k0 if count is None else count
        flags = np.zeros(count, dtype=np.uint8)
        check(self.lib.epx_cavity_batch(self.ctx, which, k0, count,
                                        flags.ctypes.data_as(_lib.c_uint8_p)))
        return flags.astype(bool)

    def cavity_site(self, k, Q, r, Qi, ri):
        flag = ctypes.c_uint8()
        check(self.lib.epx_cavity_site(self.ctx, int(k), dptr(_f64(Q)), dptr(_f64(r)),
                                       dptr(_f64(Qi)), dptr(_f64(ri)), ctypes.byref(flag)))
        return bool(flag.value)

    def get_cavity(self, k):
        Mat = np.empty((self.d, self.d), order='F')
        vec = np.empty(self.d)
        check(self.lib.epx_get_cavity(self.ctx, int(k), dptr(Mat), dptr(vec)))
        return Mat, vec

    # ---- tilted
    @staticmethod
    def sampler_opts(chains=4, iter=1000, warmup=None, thin=1, init='random',
                     max_depth=10, layout=0, flags=0, adapt='fresh'):
        o = SamplerOpts()
        if adapt not in ('fresh', 'carry'):
            raise ValueError("adapt must be 'fresh' (the reference's behaviour) or 'carry'")
        o.reserved = int(flags) | (2 if adapt == 'carry' else 0)    # bit 0: layout 2 without the speculative bookkeeping wave
        o.chains, o.iter, o.thin = int(chains), int(iter), int(thin)
        o.warmup = -1 if warmup is None else int(warmup)
        if init not in INIT_IDS:
            raise ValueError("init must be 'random', '0' or 'prev' for the GPU sampler")
        o.init = INIT_IDS[init]
        o.max_depth, o.layout = int(max_depth), int(layout)
        return o

    def tilted_batch(self, seeds, opts, prec_estim, k0=0, count=None):
        """Returns (posdef bool[count], stats (count, 8), sampling kernel ms)."""
        count = self.K - k0 if count is None else count
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        assert seeds.shape[0] == count
        flags = np.zeros(count, dtype=np.uint8)
        stats = np.zeros((count, N_STAT))
        ms = ctypes.c_double()
        check(self.lib.epx_tilted_batch(
            self.ctx, k0, count, seeds.ctypes.data_as(_lib.c_int64_p), ctypes.byref(opts),
            PREC_ESTIM_IDS[prec_estim], flags.ctypes.data_as(_lib.c_uint8_p), dptr(stats),
            ctypes.byref(ms)))
        return flags.astype(bool), stats, ms.value

    def sample_batch(self, seeds, opts, k0=0, count=None):
        count = self.K - k0 if count is None else count
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        stats = np.zeros((count, N_STAT))
        ms = ctypes.c_double()
        check(self.lib.epx_sample_batch(self.ctx, k0, count, seeds.ctypes.data_as(_lib.c_int64_p),
                                        ctypes.byref(opts), dptr(stats), ctypes.byref(ms)))
        return stats, ms.value

    # ---- test hook: one teacher-forced transition from checkpoint records (epx_sample_piece)
    CK_SCALARS = ('lps', 'eps', 'da_mu', 's_bar', 'x_bar', 'da_count', 'va_n', 'eps_sum', 'acc_sum', 'depth_sum', 'nleap_tot',
                  'ngrad', 't', 'va_counter', 'va_wsize', 'va_next', 'ndiv', 'npost', 'kept', 'failed')     # csrc/epx_pieces.h EPX_CK_LIST

    def pack_records(self, scalars, qs, wmean, wm2, inv_e):
        """Checkpoint records of the pieced launch from their parts: scalars (..., 20) in CK_SCALARS order and four
        (..., P) vectors -> (..., (4 NV + 1) * 64)."""
        nv = (self.P + 63) // 64
        rec = np.zeros(scalars.shape[:-1] + ((4 * nv + 1) * 64,))
        for j, v in enumerate((qs, wmean, wm2, inv_e)):
            rec[..., j * nv * 64:j * nv * 64 + self.P] = v
        rec[..., 3 * nv * 64 + self.P:4 * nv * 64] = 1.0            # (the metric's padding elements are 1, as the kernels keep them)
        rec[..., 4 * nv * 64:4 * nv * 64 + 20] = scalars
        return rec

    def unpack_records(self, rec):
        nv = (self.P + 63) // 64
        parts = [rec[..., j * nv * 64:j * nv * 64 + self.P] for j in range(4)]
        return rec[..., 4 * nv * 64:4 * nv * 64 + 20], parts[0], parts[1], parts[2], parts[3]

    def sample_piece(self, seeds, opts, t0, records_in):
        """Every chain takes transition t0 of a run with `opts` from the state in records_in (K, chains, record); returns
        the records of boundary t0 + 1."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        rin = np.ascontiguousarray(records_in, dtype=np.float64)
        out = np.zeros_like(rin)
        check(self.lib.epx_sample_piece(self.ctx, seeds.ctypes.data_as(_lib.c_int64_p), ctypes.byref(opts), int(t0),
                                        dptr(rin), dptr(out)))
        return out

    def moments_batch(self, samples, prec_estim, k0=0, count=None):
        """samples: (S, d, count) F-order -- the injected-draws test hook."""
        count = self.K - k0 if count is None else count
        samples = _f64(samples)
        S = samples.shape[0]
        assert samples.shape[1] == self.d
        flags = np.zeros(count, dtype=np.uint8)
        check(self.lib.epx_moments_batch(self.ctx, k0, count, dptr(samples), S,
                                         PREC_ESTIM_IDS[prec_estim],
                                         flags.ctypes.data_as(_lib.c_uint8_p)))
        return flags.astype(bool)

    def get_tilted(self, k):
        Mat = np.empty((self.d, self.d), order='F')
        vec = np.empty(self.d)
        n = ctypes.c_int()
        check(self.lib.epx_get_tilted(self.ctx, int(k), dptr(Mat), dptr(vec), ctypes.byref(n)))
        return Mat, vec, n.value

    def num_draws(self):
        n = ctypes.c_int()
        check(self.lib.epx_num_draws(self.ctx, ctypes.byref(n)))
        return n.value

    def get_draws(self, k, all_params=False):
        S = self.num_draws()
        out = np.empty((S, self.P if all_params else self.d), order='F')
        check(self.lib.epx_get_draws(self.ctx, int(k), 1 if all_params else 0, dptr(out)))
        return out

    def get_adapt(self, k, chains):
        """Adaptation history of site k: (final step size per chain, diagonal metric (P))."""
        eps = np.zeros(chains)
        metric = np.zeros(self.P)
        check(self.lib.epx_get_adapt(self.ctx, int(k), dptr(eps), dptr(metric)))
        return eps, metric

    def nuts_transitions(self, seeds, q0, eps, inv_e, nt=1, t_offset=0, layout=0, k0=0):
        """TEST HOOK: nt un-adapted transitions from q0 (count, chains, P)."""
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        count, chains, P = q0.shape
        assert P == self.P
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        eps = np.ascontiguousarray(eps, dtype=np.float64).reshape(count, chains)
        inv_e = np.ascontiguousarray(inv_e, dtype=np.float64).reshape(count, chains, P)
        out = np.zeros((count, chains, nt, P))
        cs = np.zeros((count, chains, N_STAT))
        check(self.lib.epx_nuts_transitions(self.ctx, k0, count, seeds.ctypes.data_as(_lib.c_int64_p),
                                            chains, nt, t_offset, layout, dptr(q0), dptr(eps),
                                            dptr(inv_e), dptr(out), dptr(cs)))
        return out, cs

    def last_layout(self):
        return int(self.lib.epx_last_layout(self.ctx))

    def set_site_order(self, order=None):
        """Scheduling hint: workgroup i of the next full-batch sampling calls takes site order[i]."""
        if order is None:
            check(self.lib.epx_set_site_order(self.ctx, None, 0))
            return
        order = np.ascontiguousarray(order, dtype=np.int32)
        check(self.lib.epx_set_site_order(self.ctx, order.ctypes.data, int(order.shape[0])))

    def set_site_split(self, n_lead):
        """The first `n_lead` sites of the site order run one workgroup per chain, concurrently
        with the others (see include/epx.h: epx_set_site_split)."""
        check(self.lib.epx_set_site_split(self.ctx, int(n_lead)))

    def last_split(self):
        return int(self.lib.epx_last_split(self.ctx))

    def set_piece_queue(self, piece_len=0, rate=None):
        """Piece queue of the resident sampler (include/epx.h: epx_set_piece_queue): the sites' runs in pieces of
        `piece_len` transitions, looping workgroups each claiming the site with the largest remaining predicted work; `rate`: predicted
        leapfrogs per transition of every site (None: all equal); piece_len 0 clears."""
        if rate is not None:
            rate = np.ascontiguousarray(rate, dtype=np.float64)
            if rate.shape != (self.K,):
                raise ValueError('rate: one value per site')
        check(self.lib.epx_set_piece_queue(self.ctx, int(piece_len), None if rate is None else dptr(rate)))

    def set_trace(self, sites=0):
        """Test hook: record every transition (warm-up included) of the first `sites` sites of the next sampling calls."""
        self._trace_sites = int(sites)
        check(self.lib.epx_set_trace(self.ctx, int(sites)))

    def get_trace(self, chains, iter, count=None):
        """(sites, chains, iter, 8 + P): [eps used, leapfrogs, accept, depth, divergent, eps after adaptation,
        sum of the metric, log density, sample] of every transition of the last sampling call's traced sites
        (`count`: the sites that call covered, when it was not all of them)."""
        out = np.zeros((min(self._trace_sites, self.K if count is None else int(count)), int(chains), int(iter), 8 + self.P))
        check(self.lib.epx_get_trace(self.ctx, dptr(out), out.size))
        return out

    def last_segments(self):
        return int(self.lib.epx_last_segments(self.ctx))

    def cu_count(self):
        return int(self.lib.epx_cu_count(self.ctx))

    def row_passes(self, chains, k0=0, count=None):
        """Passes over the site rows made by the last sampling call: in the lock-step layouts (3, 4: streaming / resident
        rows; 7: the row team) the (up to 4) chains of a workgroup share one pass per leapfrog; otherwise every
        gradient is its own pass (over LDS-resident rows)."""
        cs = self.get_chain_stats(chains, k0, count)[:, :, 3]
        if self.last_layout() in (1, 2, 5, 6):
            return cs.sum(axis=1)
        nb = (chains + 3) // 4
        pad = np.zeros((cs.shape[0], nb * 4)); pad[:, :chains] = cs
        return pad.reshape(cs.shape[0], nb, 4).max(axis=2).sum(axis=1)

    def team_passes(self, k0=0, count=None):
        """Layout 7 only: the passes the row team actually made per site, yielded passes included (row_passes() counts the
        gradients of the site's longest chain: a lower bound of this)."""
        count = self.K - k0 if count is None else count
        out = np.zeros(count)
        check(self.lib.epx_get_team_passes(self.ctx, int(k0), int(count), dptr(out)))
        return out

    def get_chain_stats(self, chains, k0=0, count=None):
        count = self.K - k0 if count is None else count
        out = np.zeros((count, chains, N_STAT))
        check(self.lib.epx_get_chain_stats(self.ctx, k0, count, dptr(out)))
        return out

    def logdensity_grad(self, k, theta, layout=2):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        lp = ctypes.c_double()
        g = np.zeros(self.P)
        check(self.lib.epx_logdensity_grad_layout(self.ctx, int(k), dptr(theta), int(layout), ctypes.byref(lp), dptr(g)))
        return lp.value, g

    def invert_normal_params(self, A, b):
        """util.invert_normal_params on this engine's device (new arrays)."""
        from . import util
        return util.invert_normal_params(np.asfortranarray(A, dtype=np.float64),
                                         np.asarray(b, dtype=np.float64), device=self.device)

    # ---- global update
    def site_sums(self):
        """Packed [sum Qi, sum ri, sum dQi, sum dri] of the local sites (NumPy array)."""
        out = np.zeros(self.packed_len)
        check(self.lib.epx_site_sums(self.ctx, dptr(out), None))
        return out

    def damped_trial(self, df, packed=None):
        """One damping trial from host-side packed sums (None: the sums the device holds)."""
        g, c, fb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if packed is not None:
            packed = np.ascontiguousarray(packed, dtype=np.float64)
        check(self.lib.epx_damped_trial(self.ctx, float(df), dptr(packed), None,
                                        ctypes.byref(g), ctypes.byref(c), ctypes.byref(fb)))
        return bool(g.value), bool(c.value), fb.value

    def update_trial(self, df, reduce_sums, site_base=0, stat_sum=(), stat_max=(), want_moments=False):
        """One damping trial as one stream-ordered batch (epx_update_trial, include/epx.h): site
        sums + all-reduce over the engine's communicator (with `reduce_sums`), global check,
        cavities, flags reduced over the ranks, one synchronisation.  Returns
        (global_pd, cav_pd, first_bad, stat_sum, stat_max, S, m); S, m are None without
        `want_moments`."""
        ss = np.array(stat_sum, dtype=np.float64).reshape(-1)
        sm = np.array(stat_max, dtype=np.float64).reshape(-1)
        g, c, fb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        S = m = None
        if want_moments:
            S = np.empty((self.d, self.d), order='F')
            m = np.empty(self.d)
        check(self.lib.epx_update_trial(
            self.ctx, float(df), 1 if reduce_sums else 0, int(site_base),
            dptr(ss) if ss.size else None, int(ss.size), dptr(sm) if sm.size else None, int(sm.size),
            1 if want_moments else 0, ctypes.byref(g), ctypes.byref(c), ctypes.byref(fb), dptr(S), dptr(m)))
        return bool(g.value), bool(c.value), int(fb.value), ss, sm, S, m

    # ---- the engine's communicator (dist.EpxComm binds it)
    def comm_size(self):
        r, n = ctypes.c_int(), ctypes.c_int()
        check(self.lib.epx_comm_size(self.ctx, ctypes.byref(r), ctypes.byref(n)))
        return r.value, n.value

    def comm_allreduce(self, buf, op):
        assert buf.dtype == np.float64 and buf.flags['C_CONTIGUOUS']
        check(self.lib.epx_comm_allreduce(self.ctx, dptr(buf), int(buf.size), int(op)))
        return buf

    def comm_allgather(self, buf, world):
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        out = np.empty((world, buf.size))
        check(self.lib.epx_comm_allgather(self.ctx, dptr(buf), int(buf.size), dptr(out)))
        return out

    def damp_sweep(self, damps, packed, m_target, S_target, samp_target=None):
        """Score every damping factor of `damps` (find_damp.py:146-173) -> (ndf, 5) array
        [global_pd, cav_pd (this rank's sites), mse, kl, ll]; criteria NaN unless both flags hold.
        packed: the all-reduced site sums (host array), or None for the ones the device holds."""
        damps = np.ascontiguousarray(damps, dtype=np.float64)
        m_t = np.ascontiguousarray(m_target, dtype=np.float64)
        S_t = np.asfortranarray(S_target, dtype=np.float64)
        half_logdet = float(np.sum(np.log(np.diag(np.linalg.cholesky(S_t)))))
        xm = xs = None
        ns = 0
        if samp_target is not None:
            samp = np.asarray(samp_target, dtype=np.float64)
            ns = samp.shape[0]
            xm = np.ascontiguousarray(samp.mean(axis=0))
            c = samp - xm
            xs = np.asfortranarray(c.T.dot(c))
        out = np.zeros((damps.shape[0], 5))
        host = None if packed is None else dptr(np.ascontiguousarray(packed, dtype=np.float64))
        check(self.lib.epx_damp_sweep(self.ctx, int(damps.shape[0]), dptr(damps), host, None, dptr(m_t), dptr(S_t),
                                      half_logdet, dptr(xm) if xm is not None else None,
                                      dptr(xs) if xs is not None else None, int(ns), dptr(out)))
        return out

    def mix_sums(self):
        """[sum_k scatter_k, sum_k mean_k, sum_k mean_k mean_k'] of the local sites' tilted moments."""
        out = np.zeros(2 * self.d * self.d + self.d)
        check(self.lib.epx_mix_sums(self.ctx, dptr(out)))
        return out

    def accept(self, df):
        check(self.lib.epx_accept(self.ctx, float(df)))

    def global_moments(self):
        S = np.empty((self.d, self.d), order='F')
        m = np.empty(self.d)
        check(self.lib.epx_global_moments(self.ctx, dptr(S), dptr(m)))
        return S, m

    def force_pd(self, df, thresh, target):
        forced = np.zeros(self.K, dtype=np.uint8)
        check(self.lib.epx_force_pd(self.ctx, float(df), float(thresh), float(target),
                                    forced.ctypes.data_as(_lib.c_uint8_p)))
        return forced.astype(bool)
