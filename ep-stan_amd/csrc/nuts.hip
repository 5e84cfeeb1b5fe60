// On-GPU NUTS for the per-site tilted distributions of ep-stan's EP loop.
//
// Replaces the Stan subprocess of /root/reference/epstan/method.py:43-118,
// 349-363 (PyStan 2.17 adapt_diag_e_nuts) for the hierarchical logistic
// regression family experiment/models/m{1..5}b_sg.stan.  The algorithm is the
// one restated in oracle/nuts_oracle.c (multinomial NUTS, diagonal metric,
// Stan 2.17 adaptation); the decision sequence and the Philox stream are the
// same so both can be compared draw by draw.
//
// Mapping to gfx950:
//   * one site = one workgroup (layout 1): the site's rows of X are copied
//     ONCE per site update from HBM into LDS (coalesced 16-B loads, XOR-swizzled
//     16-B slots so the row-per-lane ds_read_b128 pattern is conflict free);
//     every leapfrog gradient then runs out of LDS.  One wave = one chain.
//   * small K (fewer sites than CUs, configs C1/C2): one workgroup per
//     (site, chain) (layout 2); the 4 waves of the group split the rows and the
//     cavity-precision mat-vec and exchange partial sums through LDS with one
//     s_barrier per leapfrog.  All waves carry the chain state redundantly in
//     registers and take identical decisions.
//   * the whole NUTS state (position, momentum, gradient, tree ends, rho, ...)
//     lives in VGPRs, element e of a length-P vector in lane e%64, register
//     e/64; dot products are wave64 butterflies, no LDS round trips.
//   * beta is broadcast through SGPRs (v_readlane), so the row loop is
//     ds_read_b128 + v_fma_f64 only; the D partial sums of X'g are reduced
//     with a transposing butterfly (DP-1 exchanges instead of 6*DP).
#include "nuts_common.h"

namespace epx {



// In-kernel cycle stamps exist only in the diagnostic build (-DEPX_STAMPS); its
// run time is never quoted, only the shares of the segments.
#ifdef EPX_STAMPS
#define STAMP(i)                                                                   \
    do {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                        \
        tacc[i] += t_ - tprev; tprev = t_;                                         \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int NV, int DP, int WPC, bool OML, bool STL, bool GAUSS>
__global__ void __launch_bounds__(256)
k_nuts(NutsArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    using V = Vec<NV>;
    constexpr int LOG = Log2<DP>::v;
    constexpr int SPR = DP / 2;                       // 16-B slots per row
    constexpr int RPL = DP >= 32 ? 1 : 32 / DP;       // rows per 256-B bank line
    constexpr int XREC = 64 * (1 + NV) + 2;           // per-wave exchange record (doubles)
    constexpr int SREC = nuts_stack_record(NV);       // per-level stack record (doubles)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int team = wave / WPC, wt = wave % WPC;
    const int bps = (a.chains + a.cpb - 1) / a.cpb;
    const int sb = a.order ? a.order[blockIdx.x / bps] : (int)(blockIdx.x / bps), cb = blockIdx.x % bps;
    const int k = a.k0 + sb;
    const int chain = cb * a.cpb + team;
    const int D = a.D, d = a.d, P = a.P, model = a.model;
    const int64_t row0 = a.k_lim[k];
    const int n = (int)(a.k_lim[k + 1] - row0);

    double *Xs = reinterpret_cast<double *>(smem);
    uint8_t *ys = smem + a.off_y;                                   // 0/1 responses (logistic family) ...
    double *ysd = reinterpret_cast<double *>(smem + a.off_y);       // ... or real ones (Gaussian family)
    auto y_at = [&](int r) -> double { if constexpr (GAUSS) return ysd[r]; else return (double)ys[r]; };
    (void)ys; (void)ysd;
    double *xch = reinterpret_cast<double *>(smem + a.off_xch);

    // ---- stage the site's rows: HBM -> LDS, once per site update
    {
        const double *Xg = a.X + (size_t)row0 * D;
        const int nslot = n * SPR;
        for (int s = tid; s < nslot; s += blockDim.x) {
            const int r = s / SPR, jp = s % SPR, c0 = 2 * jp;
            double2 v;
            if ((D & 1) == 0 && c0 + 1 < D) v = *reinterpret_cast<const double2 *>(Xg + (size_t)r * D + c0);
            else {
                v.x = c0 < D ? Xg[(size_t)r * D + c0] : 0.0;
                v.y = c0 + 1 < D ? Xg[(size_t)r * D + c0 + 1] : 0.0;
            }
            const int sw = (r / RPL) & (SPR - 1);
            *reinterpret_cast<double2 *>(Xs + (size_t)r * DP + 2 * (jp ^ sw)) = v;
        }
        if constexpr (GAUSS) { for (int r = tid; r < n; r += blockDim.x) ysd[r] = a.yd[row0 + r]; }
        else { for (int r = tid; r < n; r += blockDim.x) ys[r] = a.y[row0 + r]; }
    }
    // Omega and the tree stack: LDS-typed pointers when resident (template flags keep
    // the address space static, so the compiler emits ds_read/ds_write, not flat_*)
    const double *Om_g = a.cav_Om + (size_t)k * d * d;
    double *Oms = reinterpret_cast<double *>(smem + a.off_Om);
    if constexpr (OML) {
        for (int idx = tid; idx < d * d; idx += blockDim.x) Oms[idx] = Om_g[idx];
    }
    __syncthreads();
    if (chain >= a.chains) return;        // only possible when WPC == 1 (no later barriers)
    auto om_at = [&](int idx) -> double { if constexpr (OML) return Oms[idx]; else return Om_g[idx]; };

    double *stk_l = reinterpret_cast<double *>(smem + a.off_stack) + (size_t)team * a.max_depth * SREC;
    double *stk_g = STL ? nullptr : a.stack + ((size_t)sb * a.chains + chain) * a.stack_stride;
    auto ld_stk = [&](int l, int v, int i) -> double { const int off = l * SREC + (v * NV + i) * 64 + lane; if constexpr (STL) return stk_l[off]; else return stk_g[off]; };
    auto st_stk = [&](int l, int v, int i, double x) { const int off = l * SREC + (v * NV + i) * 64 + lane; if constexpr (STL) stk_l[off] = x; else stk_g[off] = x; };

    const RngKey key = make_key((uint64_t)a.seeds[sb], chain);
    const bool laplace = (model == 4);

    // ------------------------------------------------------------- state
    V mu, inv_e, qs, gs, zq, zp, zg, pq, pp, pg, mq, mp, mg, rho, psp, psm;
    V n_rho, n_psl, bq, bg, psr, wmean, wm2;
    double lps = 0, zlp = 0, plp = 0, mlp = 0, b_key = 0, b_plp = 0;
    FORV {
        const int e = lane + 64 * i;
        mu.v[i] = e < d ? a.cav_mu[(size_t)k * d + e] : 0.0;
        inv_e.v[i] = 1.0;
        wmean.v[i] = 0.0; wm2.v[i] = 0.0;
        gs.v[i] = 0; zq.v[i] = 0; zp.v[i] = 0; zg.v[i] = 0; pq.v[i] = 0; pp.v[i] = 0; pg.v[i] = 0;
        mq.v[i] = 0; mp.v[i] = 0; mg.v[i] = 0; rho.v[i] = 0; psp.v[i] = 0; psm.v[i] = 0;
        n_rho.v[i] = 0; n_psl.v[i] = 0; bq.v[i] = 0; bg.v[i] = 0; psr.v[i] = 0;
    }
    // initial position (method.py:159 init / :404-406 init_prev)
    {
        const double *lastp = a.last + ((size_t)k * a.chains + chain) * P;
        FORV {
            const int e = lane + 64 * i;
            double q0 = 0.0;
            if (e < P) {
                if (a.init_mode == 2) q0 = lastp[e];
                else if (a.init_mode == 0) {
                    double u1, u2;
                    rng_u2(key, 0, K_INIT, (uint32_t)(e >> 1), 0, u1, u2);
                    q0 = -2.0 + 4.0 * ((e & 1) ? u2 : u1);
                }
            }
            qs.v[i] = q0;
        }
    }
    // adaptation state (stepsize_adaptation.hpp / windowed_adaptation.hpp @ Stan 2.17)
    const double DELTA = 0.8, GAMMA = 0.05, T0 = 10.0, KAPPA = 0.75, LOG08 = -0.2231435513142097558;
    double eps = 1.0, da_mu = log(10.0), s_bar = 0, x_bar = 0, da_count = 0;
    int va_init_buf = 75, va_term = 50, va_base = 25;
    if (va_init_buf + va_base + va_term > a.warmup && a.warmup >= 20) {
        va_init_buf = (int)(0.15 * a.warmup);
        va_term = (int)(0.1 * a.warmup);
        va_base = a.warmup - (va_init_buf + va_term);
    }
    int va_counter = 0, va_wsize = va_base, va_next = va_init_buf + va_base - 1;
    double va_n = 0;
    // statistics
    double eps_sum = 0, acc_sum = 0, depth_sum = 0, nleap_tot = 0, ngrad = 0;
    int ndiv = 0, npost = 0, kept = 0, failed = 0;
    // transition state
    int t = 0, mode = MODE_INIT, depth = 0, leaf = 0, nleaf = 1, fwd = 1, nleap = 0, divergent = 0, init_try = 0;
    int ss_trial = 0, ss_dir = 0, ss_after_update = 0;
    uint32_t ss_t = 0;
    double H0 = 0, lsw = 0, sum_metro = 0, eps_l = 0;
    int parity = 0;
    // Batched random numbers: lane x of u_dir holds DIR(depth x) for x < 16 and TOP(depth x-16)
    // for 16 <= x < 32 of the current transition; lane x of gum holds the Gumbel variate
    // -log(-log u) of leaf (leaf & ~63) + x of the current doubling.
    double u_dir = 0.0, gum = 0.0;
    // Leaf energy errors dH of the current doubling, lane (leaf & 63); reduced 64 at a time into
    // the running log-sum-weight (lw_m + log lw_s) and the accept statistic.
    double dhb = 0.0, lw_m = -INFINITY, lw_s = 0.0;

    FORV { zq.v[i] = qs.v[i]; }
    const bool teacher = a.eps_in != nullptr;       // fixed step size / metric (test hook)
    if (teacher) {
        eps = a.eps_in[(size_t)sb * a.chains + chain];
        if (a.inv_e_in) {
            const double *ie = a.inv_e_in + ((size_t)sb * a.chains + chain) * P;
            FORV { const int e = lane + 64 * i; if (e < P) inv_e.v[i] = ie[e]; }
        }
    }
    // opt-in carried adaptation: last call's step size of the chain, the site's pooled sample variances
    const bool carry = !teacher && a.carry_eps != nullptr && a.carry_eps[(size_t)k * a.chains + chain] > 0.0;
    if (carry) {
        eps = a.carry_eps[(size_t)k * a.chains + chain];
        da_mu = log(10.0 * eps);
        const double *cm = a.carry_metric + (size_t)k * P;
        FORV { const int e = lane + 64 * i; if (e < P) inv_e.v[i] = cm[e]; }
    }
    const uint32_t toff = (uint32_t)a.t_offset + 1u;

    // reduce the buffered leaf energy errors (lanes 0..cnt-1 of dhb)
    auto flush_dh = [&](int cnt) {
        const bool ok = lane < cnt;
        const double dh = ok ? dhb : -INFINITY;
        const double mb = wave_max(dh);
        const double m_new = fmax(lw_m, mb);
        double w = 0.0, me = 0.0;
        if (ok) {
            w = (m_new == -INFINITY) ? 0.0 : exp(dh - m_new);
            me = dh > 0 ? 1.0 : exp(dh);
        }
        wave_sum2(w, me);
        const double scale = (lw_m == -INFINITY) ? 0.0 : exp(lw_m - m_new);
        lw_s = lw_s * scale + w;
        lw_m = m_new;
        sum_metro += me;
    };

#ifdef EPX_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (;;) {
        STAMP(6);
#define EPX_AFTER_EXCHANGE_BARRIER do { } while (0)
#define EPX_DEFER_ENERGY 0
#include "nuts_gradient.inc"
#undef EPX_DEFER_ENERGY
#undef EPX_AFTER_EXCHANGE_BARRIER
        ngrad += 1.0;
        STAMP(5);

#define EPX_CHAIN_EXIT break
#define EPX_DBG_EXIT return
#define STAMP_LEAF do { } while (0)
#include "nuts_state_machine.inc"
#undef STAMP_LEAF
#undef EPX_CHAIN_EXIT
#undef EPX_DBG_EXIT
    }

    // ------------------------------------------------------------- epilogue
#ifdef EPX_STAMPS
    if (a.stamps && wt == 0 && team == 0 && lane == 0) {
        for (int i = 0; i < 7; ++i) a.stamps[(size_t)blockIdx.x * 8 + i] = tacc[i];
        a.stamps[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)ngrad;
    }
#endif
    if (wt == 0) {
        double *lastp = a.last + ((size_t)k * a.chains + chain) * P;
        FORV { const int e = lane + 64 * i; if (e < P) lastp[e] = qs.v[i]; }
        if (failed) {
            for (int kk = 0; kk < a.nkeep; ++kk) {
                double *dst = a.draws + (((size_t)k * a.chains + chain) * a.nkeep + kk) * P;
                FORV { const int e = lane + 64 * i; if (e < P) dst[e] = qs.v[i]; }
            }
        }
        if (lane == 0) {
            double *st = a.chain_stats + ((size_t)k * a.chains + chain) * ST_COUNT;
            st[ST_STEPSIZE_MEAN] = a.iter > 0 && !failed ? eps_sum / a.iter : 0.0;
            st[ST_STEPSIZE_FINAL] = eps;
            st[ST_NLEAP] = nleap_tot;
            st[ST_NGRAD] = ngrad;
            st[ST_NDIV] = ndiv;
            st[ST_ACCEPT_MEAN] = npost ? acc_sum / npost : 0.0;
            st[ST_DEPTH_MEAN] = npost ? depth_sum / npost : 0.0;
            st[ST_FAIL] = failed;
        }
    }
}

// ---------------------------------------------------------------------------
// k_nuts_spec: layout 2 (one workgroup = one chain, four cooperating gradient waves) with the
// tree bookkeeping on a FIFTH wave, one leapfrog behind.
//
// In k_nuts every wave runs the bookkeeping / adaptation state machine after each gradient
// (28 % of a leapfrog at C2) although, along a trajectory, the next leapfrog does not depend on
// it.  Here the four gradient waves (GW) integrate ahead speculatively -- leapfrog after
// leapfrog from their own copy of (q, p, grad), publishing every finished state in an LDS
// mailbox -- and the bookkeeping wave (BK) consumes those states in order with the SAME state
// machine (nuts_state_machine.inc).  When BK's decision changes the integration state (a new
// doubling from the other tree end, a new transition, a step-size search trial, a metric
// update) it posts a restart record; the GWs pick it up two leapfrogs later and drop what they
// integrated in between.  Accepted states are exactly those of the sequential algorithm, so the
// draws are bit-identical to k_nuts; the price is two wasted gradients per change of direction
// (1 % of a depth-10 transition; short trees pay more, but they are not the chains a launch
// waits for).  Synchronisation: the ONE workgroup barrier per leapfrog that the gradient's
// exchange needs anyway; mailbox and control records are double buffered by leapfrog parity.
//   tick s of a GW : gradient (barrier s inside) -> publish state s -> read BK's record of
//                    interval s-1 -> maybe restart
//   interval s of BK (between barriers s and s+1): take state s-1 -> state machine -> maybe
//                    post record s
enum { SPEC_NONE = 0, SPEC_RESTART = 1, SPEC_EXIT = 2 };

template <int NV, int DP, bool RES, bool GAUSS, bool GRP>
__global__ void __launch_bounds__(320)
k_nuts_spec(NutsArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    using V = Vec<NV>;
    constexpr int WPC = 4;
    constexpr bool OML = RES, STL = RES;              // Omega and the tree stack: both in LDS, or L2 / HBM
    constexpr int LOG = Log2<DP>::v;
    constexpr int SPR = DP / 2;
    constexpr int RPL = DP >= 32 ? 1 : 32 / DP;
    constexpr int XREC = 64 * (1 + NV) + 2;
    constexpr int SREC = nuts_stack_record(NV);
    constexpr int MREC = 3 * NV * 64 + 4 + 64;        // mailbox: q, p, grad, ll, -, generation, -, per-lane lp terms
    constexpr int CREC = 4 * NV * 64 + 4;             // control: q, p, grad, metric, eps_l, command, stamp

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool is_bk = wave == WPC;
    const int team = 0, wt = is_bk ? 0 : wave;
    (void)team;
    const int sb = a.order ? a.order[blockIdx.x / a.chains] : (int)(blockIdx.x / a.chains);
    const int chain = blockIdx.x % a.chains;
    const int k = a.k0 + sb;
    const int D = a.D, d = a.d, model = a.model;
    const int64_t row0 = a.k_lim[k];
    const int n = (int)(a.k_lim[k + 1] - row0);
    // several groups per site (GRP): theta = [phi | eta (ng) | etb (ng x D)]; records stay a.P wide
    constexpr int GREC = 66, WREC = 64 * NV + 2;        // per-group / per-wave exchange records (nuts_gradient_groups.inc)
    int ng = 1;
    int *gl_s = reinterpret_cast<int *>(smem + a.off_gl);
    double *qcopy = reinterpret_cast<double *>(smem + a.off_gl) + ((a.ngmax + 1 + 3) / 4) * 2;     // behind the row limits, 16-B aligned
    (void)qcopy;
    if constexpr (GRP) {
        const int g0 = a.site_g0[k];
        ng = a.site_g0[k + 1] - g0;
        for (int g = tid; g <= ng; g += blockDim.x) gl_s[g] = (int)(a.g_lim[g0 + g] - row0);
    }
    const int P = GRP ? d + ng * (model == 0 ? 1 : 1 + D) : a.P;
    (void)gl_s; (void)GREC; (void)WREC;

    double *Xs = reinterpret_cast<double *>(smem);
    uint8_t *ys = smem + a.off_y;                                   // 0/1 responses (logistic family) ...
    double *ysd = reinterpret_cast<double *>(smem + a.off_y);       // ... or real ones (Gaussian family)
    auto y_at = [&](int r) -> double { if constexpr (GAUSS) return ysd[r]; else return (double)ys[r]; };
    (void)ys; (void)ysd;
    double *xch = reinterpret_cast<double *>(smem + a.off_xch);
    double *mbox = reinterpret_cast<double *>(smem + a.off_spec);            // 2 x MREC
    double *ctrl = mbox + 2 * MREC;                                         // 2 x CREC
    {
        const double *Xg = a.X + (size_t)row0 * D;
        const int nslot = n * SPR;
        for (int s = tid; s < nslot; s += blockDim.x) {
            const int r = s / SPR, jp = s % SPR, c0 = 2 * jp;
            double2 v;
            if ((D & 1) == 0 && c0 + 1 < D) v = *reinterpret_cast<const double2 *>(Xg + (size_t)r * D + c0);
            else {
                v.x = c0 < D ? Xg[(size_t)r * D + c0] : 0.0;
                v.y = c0 + 1 < D ? Xg[(size_t)r * D + c0 + 1] : 0.0;
            }
            const int sw = (r / RPL) & (SPR - 1);
            *reinterpret_cast<double2 *>(Xs + (size_t)r * DP + 2 * (jp ^ sw)) = v;
        }
        if constexpr (GAUSS) { for (int r = tid; r < n; r += blockDim.x) ysd[r] = a.yd[row0 + r]; }
        else { for (int r = tid; r < n; r += blockDim.x) ys[r] = a.y[row0 + r]; }
    }
    const double *Om_g = a.cav_Om + (size_t)k * d * d;
    double *Oms = reinterpret_cast<double *>(smem + a.off_Om);
    if constexpr (OML) {
        for (int idx = tid; idx < d * d; idx += blockDim.x) Oms[idx] = Om_g[idx];
    }
    if (tid < 2) { ctrl[tid * CREC + 4 * NV * 64 + 2] = -5.0; mbox[tid * MREC + 3 * NV * 64 + 2] = -5.0; }
    auto om_at = [&](int idx) -> double { if constexpr (OML) return Oms[idx]; else return Om_g[idx]; };
    (void)om_at;
    const bool laplace = (model == 4);
    V mu, inv_e, zq, zp, zg;
    FORV {
        const int e = lane + 64 * i;
        mu.v[i] = e < d ? a.cav_mu[(size_t)k * d + e] : 0.0;
        inv_e.v[i] = 1.0; zq.v[i] = 0; zp.v[i] = 0; zg.v[i] = 0;
    }
    double eps_l = 0.0, zlp = 0.0;

    if (!is_bk) {
        // =========================================================== gradient waves
        __syncthreads();                        // rows, Omega and BK's first record are in place
        int gen = 0, parity = 0;
#ifdef EPX_STAMPS
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nticks = 0;
        unsigned long long tprev = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
        {
            const double *c = ctrl + 1 * CREC;  // initial record (stamp -1)
            FORV { zq.v[i] = c[(0 * NV + i) * 64 + lane]; zp.v[i] = c[(1 * NV + i) * 64 + lane];
                   zg.v[i] = c[(2 * NV + i) * 64 + lane]; inv_e.v[i] = c[(3 * NV + i) * 64 + lane]; }
            eps_l = c[4 * NV * 64];
        }
        const int lane0 = lane;
        for (int s = 0;; ++s) {
            // NV > 1: `lane` is re-derived through an opaque move every leapfrog, otherwise the
            // per-element index arithmetic of the gradient is hoisted out of the loop and, next to
            // the bookkeeping wave's register needs, spilled (25 scratch reloads per leapfrog at
            // NV = 2, DP = 32)
            int lane_v = lane0;
            if (NV > 1) asm volatile("" : "+v"(lane_v));
            const int lane = lane_v;
            STAMP(6);
            // BK's word of interval s-1 is complete once barrier s has passed: fetch stamp and
            // command right there, their latency hides behind the chain rule
            const double *c = ctrl + ((s + 1) & 1) * CREC;
            double c_stamp = -9.0, c_cmd = 0.0, lp_lane = 0.0, ll_u = 0.0;
#define EPX_AFTER_EXCHANGE_BARRIER do { c_stamp = c[4 * NV * 64 + 2]; c_cmd = c[4 * NV * 64 + 1]; } while (0)
#define EPX_DEFER_ENERGY 1
            if constexpr (GRP) {
#include "nuts_gradient_groups.inc"
                (void)kin;
            } else {
#include "nuts_gradient.inc"
                (void)kin;
            }
#undef EPX_DEFER_ENERGY
#undef EPX_AFTER_EXCHANGE_BARRIER
            STAMP(5);
#ifdef EPX_STAMPS
            ++nticks;
#endif
            if (wt == 0) {
                double *m = mbox + (s & 1) * MREC;
                FORV { m[(0 * NV + i) * 64 + lane] = zq.v[i]; m[(1 * NV + i) * 64 + lane] = zp.v[i];
                       m[(2 * NV + i) * 64 + lane] = zg.v[i]; }
                m[3 * NV * 64 + 4 + lane] = lp_lane;
                if (lane == 0) { m[3 * NV * 64] = ll_u; m[3 * NV * 64 + 2] = (double)gen; }
            }
            if (s >= 1 && c_stamp == (double)(s - 1)) {
                const int cmd = (int)c_cmd;
                if (cmd == SPEC_EXIT) {
#ifdef EPX_STAMPS
                    if (a.stamps && wt == 0 && lane == 0) {
                        // slot 6 belongs to the bookkeeping wave: its busy time between barriers
                        for (int i = 0; i < 6; ++i) a.stamps[(size_t)blockIdx.x * 8 + i] = tacc[i];
                        a.stamps[(size_t)blockIdx.x * 8 + 7] = nticks;
                    }
#endif
                    break;
                }
                if (cmd == SPEC_RESTART) {
                    FORV { zq.v[i] = c[(0 * NV + i) * 64 + lane]; zp.v[i] = c[(1 * NV + i) * 64 + lane];
                           zg.v[i] = c[(2 * NV + i) * 64 + lane]; inv_e.v[i] = c[(3 * NV + i) * 64 + lane]; }
                    eps_l = c[4 * NV * 64];
                    ++gen;
                }
            }
        }
        return;
    }

    // =============================================================== bookkeeping wave
    double *stk_l = reinterpret_cast<double *>(smem + a.off_stack);
    double *stk_g = STL ? nullptr : a.stack + ((size_t)sb * a.chains + chain) * a.stack_stride;
    auto ld_stk = [&](int l, int v, int i) -> double { const int off = l * SREC + (v * NV + i) * 64 + lane; if constexpr (STL) return stk_l[off]; else return stk_g[off]; };
    auto st_stk = [&](int l, int v, int i, double x) { const int off = l * SREC + (v * NV + i) * 64 + lane; if constexpr (STL) stk_l[off] = x; else stk_g[off] = x; };
    const RngKey key = make_key((uint64_t)a.seeds[sb], chain);
    V qs, gs, pq, pp, pg, mq, mp, mg, rho, psp, psm, wmean, wm2, bq, bg, sent_e, in_q, in_p, in_g;
    double lps = 0, plp = 0, mlp = 0, b_key = 0, b_plp = 0;
    FORV {
        bq.v[i] = 0; bg.v[i] = 0; wmean.v[i] = 0.0; wm2.v[i] = 0.0; gs.v[i] = 0; pq.v[i] = 0; pp.v[i] = 0; pg.v[i] = 0;
        mq.v[i] = 0; mp.v[i] = 0; mg.v[i] = 0; rho.v[i] = 0; psp.v[i] = 0; psm.v[i] = 0;
    }
    {
        const double *lastp = a.last + ((size_t)k * a.chains + chain) * a.P;
        FORV {
            const int e = lane + 64 * i;
            double q0 = 0.0;
            if (e < P) {
                if (a.init_mode == 2) q0 = lastp[e];
                else if (a.init_mode == 0) {
                    double u1, u2;
                    rng_u2(key, 0, K_INIT, (uint32_t)(e >> 1), 0, u1, u2);
                    q0 = -2.0 + 4.0 * ((e & 1) ? u2 : u1);
                }
            }
            qs.v[i] = q0;
        }
    }
    const double DELTA = 0.8, GAMMA = 0.05, T0 = 10.0, KAPPA = 0.75, LOG08 = -0.2231435513142097558;
    double eps = 1.0, da_mu = log(10.0), s_bar = 0, x_bar = 0, da_count = 0;
    int va_init_buf = 75, va_term = 50, va_base = 25;
    if (va_init_buf + va_base + va_term > a.warmup && a.warmup >= 20) {
        va_init_buf = (int)(0.15 * a.warmup);
        va_term = (int)(0.1 * a.warmup);
        va_base = a.warmup - (va_init_buf + va_term);
    }
    int va_counter = 0, va_wsize = va_base, va_next = va_init_buf + va_base - 1;
    double va_n = 0;
    double eps_sum = 0, acc_sum = 0, depth_sum = 0, nleap_tot = 0, ngrad = 0;
    int ndiv = 0, npost = 0, kept = 0, failed = 0;
    int t = 0, mode = MODE_INIT, depth = 0, leaf = 0, nleaf = 1, fwd = 1, nleap = 0, divergent = 0, init_try = 0;
    int ss_trial = 0, ss_dir = 0, ss_after_update = 0;
    uint32_t ss_t = 0;
    double H0 = 0, lsw = 0, sum_metro = 0;
    double u_dir = 0.0, gum = 0.0;
    double dhb = 0.0, lw_m = -INFINITY, lw_s = 0.0;
    FORV { zq.v[i] = qs.v[i]; }
    const bool teacher = a.eps_in != nullptr;
    if (teacher) {
        eps = a.eps_in[(size_t)sb * a.chains + chain];
        if (a.inv_e_in) {
            const double *ie = a.inv_e_in + ((size_t)sb * a.chains + chain) * a.P;
            FORV { const int e = lane + 64 * i; if (e < P) inv_e.v[i] = ie[e]; }
        }
    }
    // opt-in carried adaptation: last call's step size of the chain, the site's pooled sample variances
    const bool carry = !teacher && a.carry_eps != nullptr && a.carry_eps[(size_t)k * a.chains + chain] > 0.0;
    if (carry) {
        eps = a.carry_eps[(size_t)k * a.chains + chain];
        da_mu = log(10.0 * eps);
        const double *cm = a.carry_metric + (size_t)k * a.P;
        FORV { const int e = lane + 64 * i; if (e < P) inv_e.v[i] = cm[e]; }
    }
    const uint32_t toff = (uint32_t)a.t_offset + 1u;
    auto flush_dh = [&](int cnt) {
        const bool ok = lane < cnt;
        const double dh = ok ? dhb : -INFINITY;
        const double mb = wave_max(dh);
        const double m_new = fmax(lw_m, mb);
        double w = 0.0, me = 0.0;
        if (ok) {
            w = (m_new == -INFINITY) ? 0.0 : exp(dh - m_new);
            me = dh > 0 ? 1.0 : exp(dh);
        }
        wave_sum2(w, me);
        const double scale = (lw_m == -INFINITY) ? 0.0 : exp(lw_m - m_new);
        lw_s = lw_s * scale + w;
        lw_m = m_new;
        sum_metro += me;
    };
    // post the integration state the GWs have to continue from (stamp = interval)
    double sent_eps = 0.0;
    auto post = [&](int stamp, int cmd) {
        double *c = ctrl + (stamp & 1) * CREC;
        FORV { c[(0 * NV + i) * 64 + lane] = zq.v[i]; c[(1 * NV + i) * 64 + lane] = zp.v[i];
               c[(2 * NV + i) * 64 + lane] = zg.v[i]; c[(3 * NV + i) * 64 + lane] = inv_e.v[i]; sent_e.v[i] = inv_e.v[i]; }
        if (lane == 0) { c[4 * NV * 64] = eps_l; c[4 * NV * 64 + 1] = (double)cmd; c[4 * NV * 64 + 2] = (double)stamp; }
        sent_eps = eps_l;
    };
    int gen = 0;
    post(-1, SPEC_RESTART);                     // the initial point, eps_l = 0: the first "leapfrog" is its gradient
    __syncthreads();
#ifdef EPX_STAMPS
    unsigned long long bk_busy = 0, bk_in = 0;
#endif
    for (int s = 0;; ++s) {
#ifdef EPX_STAMPS
        if (s > 0) { __builtin_amdgcn_s_waitcnt(0xC07F); bk_busy += __builtin_amdgcn_s_memtime() - bk_in; }
#endif
        __syncthreads();                        // barrier s
#ifdef EPX_STAMPS
        bk_in = __builtin_amdgcn_s_memtime();
#endif
        if (s == 0) continue;
        const double *m = mbox + ((s - 1) & 1) * MREC;
        if ((int)m[3 * NV * 64 + 2] != gen) continue;           // integrated past a change of state: dropped
        FORV { in_q.v[i] = m[(0 * NV + i) * 64 + lane]; in_p.v[i] = m[(1 * NV + i) * 64 + lane];
               in_g.v[i] = m[(2 * NV + i) * 64 + lane];
               zq.v[i] = in_q.v[i]; zp.v[i] = in_p.v[i]; zg.v[i] = in_g.v[i]; }
        // lp and the kinetic energy: the reductions the gradient waves left to this wave
        double lpt = m[3 * NV * 64 + 4 + lane], ks = 0.0;
        FORV ks += inv_e.v[i] * zp.v[i] * zp.v[i];
        wave_sum2(lpt, ks);
        zlp = lpt + m[3 * NV * 64];
        const double kin = 0.5 * ks;
        ngrad += 1.0;
        V n_rho, n_psl, psr;
#define EPX_CHAIN_EXIT { post(s, SPEC_EXIT); __syncthreads(); break; }
#define EPX_DBG_EXIT { post(s, SPEC_EXIT); __syncthreads(); return; }
#define STAMP_LEAF do { } while (0)
#include "nuts_state_machine.inc"
#undef STAMP_LEAF
#undef EPX_CHAIN_EXIT
#undef EPX_DBG_EXIT
        // the GWs keep integrating from the state they published; tell them only if that is no
        // longer where (or how) the trajectory continues
        int moved = (eps_l != sent_eps) ? 1 : 0;
        FORV {
            moved |= (zq.v[i] != in_q.v[i]) | (zp.v[i] != in_p.v[i]) | (zg.v[i] != in_g.v[i]) | (inv_e.v[i] != sent_e.v[i]);
        }
        if (__any(moved)) { post(s, SPEC_RESTART); ++gen; }
    }

    // ------------------------------------------------------------- epilogue (BK owns the chain)
#ifdef EPX_STAMPS
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] = bk_busy;
#endif
    {
        double *lastp = a.last + ((size_t)k * a.chains + chain) * a.P;
        FORV { const int e = lane + 64 * i; if (e < P) lastp[e] = qs.v[i]; }
        if (failed) {
            for (int kk = 0; kk < a.nkeep; ++kk) {
                double *dst = a.draws + (((size_t)k * a.chains + chain) * a.nkeep + kk) * a.P;
                FORV { const int e = lane + 64 * i; if (e < P) dst[e] = qs.v[i]; }
            }
        }
        if (lane == 0) {
            double *st = a.chain_stats + ((size_t)k * a.chains + chain) * ST_COUNT;
            st[ST_STEPSIZE_MEAN] = a.iter > 0 && !failed ? eps_sum / a.iter : 0.0;
            st[ST_STEPSIZE_FINAL] = eps;
            st[ST_NLEAP] = nleap_tot;
            st[ST_NGRAD] = ngrad;
            st[ST_NDIV] = ndiv;
            st[ST_ACCEPT_MEAN] = npost ? acc_sum / npost : 0.0;
            st[ST_DEPTH_MEAN] = npost ? depth_sum / npost : 0.0;
            st[ST_FAIL] = failed;
        }
    }
}

// ---------------------------------------------------------------------------
// host side: LDS layout + dispatch over the instantiated shapes
size_t nuts_lds_layout(NutsArgs &a, int wpc, int dp, int n_max) {
    const int nv = (a.P + 63) / 64;
    size_t off = (size_t)n_max * dp * 8;
    a.n_max = n_max;
    a.off_y = (int)off; off += (((size_t)n_max * (a.gauss ? 8 : 1)) + 15) & ~(size_t)15;
    a.off_xch = (int)off;
    a.off_gl = 0;
    if (a.grp) {
        // several groups per site: per-wave (Omega partials, ll) and per-group (dbeta, dalpha) records, both parities,
        // then the group row limits
        off += (size_t)2 * (wpc * (64 * nv + 2) + (size_t)a.ngmax * 66) * 8;
        a.off_gl = (int)off;
        off += ((size_t)(a.ngmax + 1) * 4 + 15) & ~(size_t)15;
        off += (size_t)wpc * 2 * 64 * nv * 8;              // per-wave copies of q and exp(q)
    } else if (wpc > 1) off += (size_t)2 * wpc * (64 * (1 + nv) + 2) * 8;
    off = (off + 15) & ~(size_t)15;
    const size_t cap = 160 * 1024;
    const size_t om = (size_t)a.d * a.d * 8;
    a.om_in_lds = 0; a.off_Om = (int)off;
    if (off + om <= cap) { a.om_in_lds = 1; off += om; off = (off + 15) & ~(size_t)15; }
    const size_t stack = (size_t)a.cpb * a.max_depth * nuts_stack_record(nv) * 8;
    a.stack_in_lds = 0; a.off_stack = (int)off;
    if (a.om_in_lds && off + stack <= cap) { a.stack_in_lds = 1; off += stack; }
    if (wpc > 1 && !a.stack_in_lds && a.om_in_lds) {      // layout 2 is built for "both" or "neither"
        a.om_in_lds = 0;
        off = (size_t)a.off_Om;
    }
    // layout 2: room for the speculative kernel's mailbox / control records?
    a.off_spec = 0;
    if (wpc == 4 && a.cpb == 1 && a.om_in_lds == a.stack_in_lds) {
        const size_t rec = (size_t)2 * ((3 * nv * 64 + 4 + 64) + (4 * nv * 64 + 4)) * 8;
        off = (off + 15) & ~(size_t)15;
        if (off + rec <= cap) { a.off_spec = (int)off; off += rec; }
    }
    a.lds_bytes = (int)off;
    return off;
}

template <int NV, int DP, bool RES, bool GAUSS = false, bool GRP = false>
static int launch_spec(const NutsArgs &a, int nblocks, hipStream_t stream) {
    auto kern = k_nuts_spec<NV, DP, RES, GAUSS, GRP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(320), a.lds_bytes, stream, a);
    return (int)hipGetLastError();
}

template <int NV, int DP, int WPC, bool OML, bool STL, bool GAUSS = false>
static int launch_one(const NutsArgs &a, int nblocks, hipStream_t stream) {
    auto kern = k_nuts<NV, DP, WPC, OML, STL, GAUSS>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
    if (e != hipSuccess) return (int)e;
    const int threads = 64 * WPC * a.cpb;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(threads), a.lds_bytes, stream, a);
    return (int)hipGetLastError();
}

template <int NV, int DP>
static int launch_wpc(const NutsArgs &a, int nblocks, int wpc, hipStream_t stream) {
    // (Omega in LDS, stack in LDS): layout 2 has both or neither; layout 1 may have Omega only
    if (a.grp) {
        // several groups per site: the everything-resident bookkeeping-wave kernel only (the host checks)
        if (wpc == 4 && a.off_spec > 0 && a.om_in_lds && a.stack_in_lds && !a.no_spec)
            return a.gauss ? launch_spec<NV, DP, true, true, true>(a, nblocks, stream)
                           : launch_spec<NV, DP, true, false, true>(a, nblocks, stream);
        return -1;
    }
    if (a.gauss) {
        // Gaussian-likelihood family: the everything-resident kernels only (the host refuses other shapes)
        if (wpc == 4 && a.off_spec > 0 && a.om_in_lds && !a.no_spec) return launch_spec<NV, DP, true, true>(a, nblocks, stream);
        if (wpc == 1 && a.om_in_lds && a.stack_in_lds) return launch_one<NV, DP, 1, true, true, true>(a, nblocks, stream);
        if (wpc == 1 && a.om_in_lds) return launch_one<NV, DP, 1, true, false, true>(a, nblocks, stream);
        return -1;
    }
    if (wpc == 4) {
        if (a.off_spec > 0 && !a.no_spec)
            return a.om_in_lds ? launch_spec<NV, DP, true>(a, nblocks, stream) : launch_spec<NV, DP, false>(a, nblocks, stream);
        if (a.om_in_lds && a.stack_in_lds) return launch_one<NV, DP, 4, true, true>(a, nblocks, stream);
        return launch_one<NV, DP, 4, false, false>(a, nblocks, stream);
    }
    if (wpc == 1) {
        if (a.om_in_lds && a.stack_in_lds) return launch_one<NV, DP, 1, true, true>(a, nblocks, stream);
        if (a.om_in_lds) return launch_one<NV, DP, 1, true, false>(a, nblocks, stream);
        return launch_one<NV, DP, 1, false, false>(a, nblocks, stream);
    }
    return -1;
}

template <int NV>
static int launch_dp(const NutsArgs &a, int nblocks, int wpc, int dp, hipStream_t stream) {
    switch (dp) {
    case 4: return launch_wpc<NV, 4>(a, nblocks, wpc, stream);
    case 8: return launch_wpc<NV, 8>(a, nblocks, wpc, stream);
    case 16: return launch_wpc<NV, 16>(a, nblocks, wpc, stream);
    case 32: return launch_wpc<NV, 32>(a, nblocks, wpc, stream);
    }
    return -1;
}

int launch_nuts(const NutsArgs &a, int count, int wpc, int dp, int nv, hipStream_t stream) {
    const int bps = (a.chains + a.cpb - 1) / a.cpb;
    const int nblocks = count * bps;
    if (nv == 1) return launch_dp<1>(a, nblocks, wpc, dp, stream);
    if (nv == 2) return launch_dp<2>(a, nblocks, wpc, dp, stream);
    return -1;
}

// ---------------------------------------------------------------------------
// per-site statistics: mean step size over chains (method.py:99-102) and the
// max split-Rhat over the sampled coordinates (PyStan 2.17 _chains.pyx form,
// method.py:104), one block per site.
__global__ void __launch_bounds__(256)
k_site_stats(RhatArgs a) {
    __shared__ double red[16];
    const int sbk = blockIdx.x, k = a.k0 + sbk, tid = threadIdx.x;
    const int C = a.chains, P = a.P;
    const int Pk = a.site_g0 ? a.d + (a.site_g0[k + 1] - a.site_g0[k]) * a.pg : P;   // this site's coordinates
    const int n = a.nkeep - (a.nkeep % 2), hlen = n / 2;
    double rmax = 0.0;
    for (int e = tid; e < Pk; e += blockDim.x) {
        if (hlen < 2) break;
        double mean_of_means = 0.0, var_within = 0.0;
        // two passes per half chain (numerically plain, like NumPy var)
        double hm[32];
        const int H = 2 * C;
        for (int hc = 0; hc < H && hc < 32; ++hc) {
            const int c = hc >> 1, second = hc & 1;
            const double *base = a.draws + (((size_t)k * C + c) * a.nkeep + (second ? a.nkeep - hlen : 0)) * P + e;
            double s = 0.0;
            for (int t = 0; t < hlen; ++t) s += base[(size_t)t * P];
            const double m = s / hlen;
            double v = 0.0;
            for (int t = 0; t < hlen; ++t) { const double dlt = base[(size_t)t * P] - m; v += dlt * dlt; }
            hm[hc] = m;
            var_within += v / (hlen - 1);
            mean_of_means += m;
        }
        mean_of_means /= H; var_within /= H;
        double vb = 0.0;
        for (int hc = 0; hc < H && hc < 32; ++hc) vb += (hm[hc] - mean_of_means) * (hm[hc] - mean_of_means);
        const double var_between = hlen * vb / (H - 1);
        const double rh = sqrt((var_between / var_within + hlen - 1) / hlen);
        rmax = fmax(rmax, rh);
        if (isnan(rh)) rmax = NAN;
    }
    // block max (NaN propagates like np.max)
    double v = rmax;
    int isn = isnan(v) ? 1 : 0;
    if (isn) v = 0.0;
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    if ((tid & 63) == 0) red[tid >> 6] = v;
    isn = __syncthreads_or(isn);
    if (tid == 0) {
        double r = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = fmax(r, red[w]);
        double *out = a.site_stats + (size_t)sbk * 8;
        double step = 0, nleap = 0, ngrad = 0, ndiv = 0, acc = 0, dep = 0, fail = 0;
        for (int c = 0; c < C; ++c) {
            const double *st = a.chain_stats + ((size_t)k * C + c) * ST_COUNT;
            step += st[ST_STEPSIZE_MEAN]; nleap += st[ST_NLEAP]; ngrad += st[ST_NGRAD];
            ndiv += st[ST_NDIV]; acc += st[ST_ACCEPT_MEAN]; dep += st[ST_DEPTH_MEAN]; fail += st[ST_FAIL];
        }
        out[0] = step / C; out[1] = isn ? NAN : r; out[2] = nleap; out[3] = ngrad;
        out[4] = ndiv; out[5] = acc / C; out[6] = dep / C; out[7] = fail;
    }
}

// History for `adapt = carry`: per chain the step size its warm-up ended with, per site the pooled
// variance of the kept draws of ALL its chains, regularised like Stan's windowed estimate
// ((n / (n + 5)) var + 1e-3 (5 / (n + 5)), var_adaptation.hpp @ 2.17).  A site with a failed chain
// keeps no history (step size -1): its next update adapts from scratch.
__global__ void __launch_bounds__(128)
k_carry_update(CarryArgs a) {
    const int k = a.k0 + blockIdx.x, tid = threadIdx.x;
    const int C = a.chains, P = a.P, n = C * a.nkeep;
    double fail = 0.0;
    for (int c = 0; c < C; ++c) fail += a.chain_stats[((size_t)k * C + c) * ST_COUNT + ST_FAIL];
    const bool ok = fail == 0.0 && n >= 2;
    if (tid < C) a.carry_eps[(size_t)k * C + tid] = ok ? a.chain_stats[((size_t)k * C + tid) * ST_COUNT + ST_STEPSIZE_FINAL] : -1.0;
    if (!ok) return;
    const double *base = a.draws + (size_t)k * n * P;               // draw s of the site at base + s * P (chain-major)
    for (int e = tid; e < P; e += blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < n; ++t) s += base[(size_t)t * P + e];
        const double m = s / n;
        double v = 0.0;
        for (int t = 0; t < n; ++t) { const double dlt = base[(size_t)t * P + e] - m; v += dlt * dlt; }
        const double s2 = v / (n - 1.0), nn = (double)n;
        a.carry_metric[(size_t)k * P + e] = (nn / (nn + 5.0)) * s2 + 1e-3 * (5.0 / (nn + 5.0));
    }
}

__global__ void k_rng_probe(uint64_t seed, int chain, uint32_t t, uint32_t kind, uint32_t a,
                            uint32_t b, double *out4) {
    if (threadIdx.x == 0) {
        RngKey key = make_key(seed, chain);
        double u1, u2;
        rng_u2(key, t, kind, a, b, u1, u2);
        out4[0] = u1; out4[1] = u2;
        out4[2] = rng_normal(key, t, kind, (int)(2 * a), b);
        out4[3] = rng_normal(key, t, kind, (int)(2 * a + 1), b);
    }
}

}  // namespace epx
