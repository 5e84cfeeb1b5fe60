// Streaming variant of the on-GPU NUTS sampler for sites whose rows do NOT fit LDS
// (BASELINE config C5: D = 128, n_j = 2000 -> 2 MB of X per site, d = 258, P = 387).
//
// Same algorithm as k_nuts (the tree bookkeeping / adaptation code is the shared include
// nuts_state_machine.inc); what changes is the gradient, which is HBM-bound here
// (0.5 flop per byte of X per chain):
//   * one workgroup = one site; waves 0..3 = chains, advancing in LOCK STEP: every loop
//     iteration is one leapfrog of every chain, so the site's rows are streamed from HBM ONCE
//     per leapfrog for all chains (n_j*D*8 bytes) instead of once per chain; waves 4 and 5 are
//     the loader and the logistic wave of the row streaming engine (epx_stream_tile.h: LDS-DMA
//     ring, MFMA f64 skinny products, one barrier per 16-row tile);
//   * the cavity term Omega (phi - mu) is one pass over Omega (d*d*8 bytes from HBM/L2) for
//     all chains (thread = row of Omega);
//   * per-chain vectors have up to 448 coordinates (element e in lane e%64, register e/64).
//     Only the nine vectors touched on every leaf stay in registers; the thirteen that change
//     once per subtree / transition (current sample, tree ends, rho, p-sharp of the ends,
//     adaptation sums) live in a per-chain "cold store" in global memory (L2-resident,
//     512-byte coalesced rows) -- 23 x 14 VGPRs would otherwise spill into scratch and pay a
//     memory round trip in the middle of every bookkeeping step;
//   * the parameter transforms gather through LDS copies of q and exp(q).
// Algorithmic HBM bytes per leapfrog of one site: n_j*D*8 + n_j*4 + d*d*8 (X, y, Omega), for
// min(chains, 4) gradients.
#include "epx_device.h"
#include "epx_kernels.h"
#include "epx_stream_tile.h"
#include "epx_pieces.h"
#include <type_traits>

namespace epx {

template <int NV> struct VecS { double v[NV]; };
#define FORV _Pragma("unroll") for (int i = 0; i < NV; ++i)
// In-kernel cycle stamps exist only in the diagnostic build (-DEPX_STAMPS); its run time is
// never quoted, only the shares of the segments (scripts/stamps_stream.py).
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

// cold-store vector: same `.v[i]` syntax as VecS, backed by global memory
struct ColdRef {
    gdouble *p;
    __device__ operator double() const { return *p; }
    __device__ const ColdRef &operator=(double x) const { *p = x; return *this; }
    __device__ const ColdRef &operator=(const ColdRef &o) const { const double x = *o.p; *p = x; return *this; }
    __device__ const ColdRef &operator+=(double x) const { *p = *p + x; return *this; }
};
// b is wave-uniform (scalar registers), so every access is `saddr + lane*8 + immediate`
struct ColdIdx { gdouble *b; int lane; __device__ ColdRef operator[](int i) const { return ColdRef{b + (lane + 64 * i)}; } };
struct ColdV { ColdIdx v; };
enum { CV_QS, CV_GS, CV_PQ, CV_PP, CV_PG, CV_MQ, CV_MP, CV_MG, CV_RHO, CV_PSP, CV_PSM, CV_WMEAN, CV_WM2, CV_BQ, CV_BG, CV_COUNT };

enum { SMODE_INIT = 0, SMODE_SS = 1, SMODE_TREE = 2 };
#define MODE_INIT SMODE_INIT
#define MODE_SS SMODE_SS
#define MODE_TREE SMODE_TREE

#ifndef EPX_STREAM_WAVE_SCALAR
#define EPX_STREAM_WAVE_SCALAR 2
#endif
#ifndef EPX_OM_UNROLL
#define EPX_OM_UNROLL 16
#endif
constexpr int OM_UNROLL = EPX_OM_UNROLL;   // columns of Omega in flight per thread
static_assert(OM_UNROLL <= EPX_OM_PAD_COLS, "the cavity-column ring reads OM_UNROLL columns past a site's Omega: epx_api.hip pads EPX_OM_PAD_COLS");

typedef const __attribute__((address_space(4))) NutsArgs StreamArgsK;    // the kernel arguments where they are: kernarg segment

// One piece of a site's run on the streaming layout: transitions [q_t0, q_t0 + piece length) of site q_site by the waves of
// one workgroup (PIECED), or the whole run of the site blockIdx.x names.
// RES: the resident variant (rows in LDS for the whole site update, 4 chain waves only, D <= 32)
template <int NV, int DPB, bool RES, bool PIECED>
__device__ __forceinline__ void stream_piece(StreamArgsK *kargs_p, int q_site, int q_t0, unsigned long long tl_entry, unsigned long long tl_claim) {
    extern __shared__ __align__(16) unsigned char smem[];
    StreamArgsK &a = *kargs_p;
    (void)tl_entry; (void)tl_claim;
    const bool queued = PIECED;
    const int t_begin = queued ? q_t0 : 0;
    const int q_len = queued ? piece_len_at(a, q_site, q_t0) : 0;
    const int t_end = queued ? (q_t0 + q_len < a.iter ? q_t0 + q_len : a.iter) : a.iter;
    const bool resume = t_begin > 0;
    using V = VecS<NV>;
    constexpr int NT = RES ? 256 : STREAM_THREADS;
    // register vectors when they fit (NV <= 2: 23 x 4 VGPRs), cold store otherwise
    using CV = typename std::conditional<(RES && NV <= 2), VecS<NV>, ColdV>::type;
    constexpr int SREC = nuts_stack_record(NV); // per-level stack record (doubles)
    constexpr int PMAX = 64 * NV;

    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    const int wave0 = wave;          // (the leapfrog loop re-derives `wave` as a SCALAR per iteration: see there)
    const int wt = 0;                           // one wave per chain
    const bool is_chain = wave < NCH;
    const int bps = (a.chains + NCH - 1) / NCH;
    const int sb = queued ? q_site : (a.order ? a.order[blockIdx.x / bps] : (int)(blockIdx.x / bps));
    const int cb = queued ? 0 : blockIdx.x % bps;
    const int k = a.k0 + sb;
    const int chain = cb * NCH + (is_chain ? wave : 0);
    const bool active = is_chain && chain < a.chains;
    const int D = a.D, d = a.d, model = a.model;
    const int64_t row0 = a.k_lim[k];
    // groups of the site (K < J): ng blocks of rows, each with its own eta (and etb); the site
    // samples P = d + ng * pg coordinates, records (draws, last, stack) use the stride a.P
    const int g0 = a.site_g0 ? a.site_g0[k] : 0;
    const int ng = a.site_g0 ? a.site_g0[k + 1] - g0 : 1;
    const int P = d + ng * (model == 0 ? 1 : 1 + D);
    // Gaussian-likelihood family (experiment/models/m*a.stan): phi = [log sigma | the b-model's phi]: the b-model's
    // hyper-parameters sit O = 1 further on; responses are real (a.yd)
    const int O = a.gauss ? 1 : 0;

    // ---- LDS carve-up
    StreamLds L;
    ResMap RM;
    unsigned eng_end;
    if constexpr (RES) {
        RM = res_map<DPB>(a.n_max, a.ngmax, a.ntmax);
        L.beta_s = reinterpret_cast<double *>(smem + RM.beta); L.Gs = reinterpret_cast<double *>(smem + RM.gsum);
        L.alpha_s = reinterpret_cast<double *>(smem + RM.alpha); L.da_s = reinterpret_cast<double *>(smem + RM.da);
        L.tdesc = reinterpret_cast<int *>(smem + RM.tdesc);
        eng_end = RM.end;
    } else {
        L.template carve<DPB>(smem, a.ngmax, a.ntmax, a.gauss);
        eng_end = stream_map<DPB>(a.ngmax, a.ntmax, a.gauss).end;
    }
    double *mu_s = reinterpret_cast<double *>(smem + eng_end);     // d (padded to even)
    double *vs4 = mu_s + ((d + 1) & ~1);                           // d x 4: phi - mu, [e][chain]
    double *Ovs = vs4 + d * NCH;                                   // d x 4: Omega (phi - mu)   (streaming variant)
    double *q_s = Ovs + (RES ? 0 : d * NCH);                       // 4 x PMAX
    double *eq_s = q_s + NCH * PMAX;                               // 4 x PMAX
    double *opart = eq_s + NCH * PMAX;                             // RES: 4 waves x d x 4 partial Omega products
    double *Om_s = opart + (RES ? NCH * d * NCH : 0);              // RES: Omega itself when it fits (a.om_in_lds)
    int *sh_done = reinterpret_cast<int *>(Om_s + ((RES && a.om_in_lds) ? d * d : 0));
    if (tid == 0) *sh_done = 0;
    for (int e = tid; e < d; e += NT) mu_s[e] = a.cav_mu[(size_t)k * d + e];

    PassArgs<DPB> site;
    site.Xg = a.X + (size_t)row0 * D;
    site.gauss = a.gauss;
    site.yg = a.gauss ? reinterpret_cast<const int *>(a.yd + row0) : a.y32 + row0;
    site.n = (int)(a.k_lim[k + 1] - row0); site.D = D;
    site.ngmax = a.ngmax; site.ntmax = a.ntmax;
    site.lds0 = (unsigned)(size_t)smem; site.slot_f = 0; site.slot_i = 0; site.t_i = 0;
    site.wave = wave; site.lane = lane0;
    {
        // tile table: every group's rows are tiled on their own
        int *nt_s = sh_done + 1;
        if (tid == 0) {
            int t = 0;
            for (int g = 0; g < ng; ++g) {
                const long long lo = a.g_lim ? a.g_lim[g0 + g] - row0 : 0;
                const long long hi = a.g_lim ? a.g_lim[g0 + g + 1] - row0 : site.n;
                for (long long r = lo; r < hi; r += TR, ++t) {
                    L.tdesc[2 * t] = (int)r;
                    L.tdesc[2 * t + 1] = (int)(hi - r < TR ? hi - r : TR) | (g << 8);
                }
            }
            *nt_s = t;
        }
        __syncthreads();
        site.ntile = *nt_s;
    }
    if constexpr (RES) {
        res_load_site<DPB>(site.Xg, site.yg, site.n, D, a.ngmax, site.lds0, RM, tid, NT);
        __syncthreads();
    } else {
        if (wave == NCH) { loader_init<DPB>(site, lane0); ring_prime<DPB>(site, lane0); }
    }

    const double *Om_g = a.cav_Om + (size_t)k * d * d;
    if constexpr (RES) {
        if (a.om_in_lds) { for (int i = tid; i < d * d; i += NT) Om_s[i] = Om_g[i]; }
    }
    // (pieced launches: tree stack and cold store belong to the WORKGROUP, so that no line of them is ever cached by the
    // L2s of two XCDs; what a chain carries from piece to piece goes through the checkpoint record)
    const size_t chain_slot = (size_t)(queued ? (int)blockIdx.x : sb) * a.chains + (active ? chain : 0);
    double *ckp = queued ? piece_record(a, sb, t_begin, active ? chain : 0, NV) : nullptr;            // the record this piece starts from
    double *ckp_out = queued ? piece_record(a, sb, t_end, active ? chain : 0, NV) : nullptr;        // ... and the one it leaves
    // wave-uniform base pointers (held in scalar registers; lanes add lane*8)
    auto uniform_ptr = [](double *p) -> gdouble * {
        const unsigned long long u = (unsigned long long)p;
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return reinterpret_cast<gdouble *>((uintptr_t)(((unsigned long long)hi32 << 32) | lo32));
    };
    gdouble *stk_g = uniform_ptr(a.stack + chain_slot * ((size_t)a.max_depth * SREC + (size_t)CV_COUNT * PMAX));
    gdouble *cold = stk_g + (size_t)a.max_depth * SREC;
    int stk_lane = lane0;               // (the loop's opaque copy of the lane index: set at the top of every iteration)
    auto ld_stk = [&](int l, int v, int i) -> double { return stk_g[l * SREC + (v * NV + i) * 64 + stk_lane]; };
    auto st_stk = [&](int l, int v, int i, double x) { stk_g[l * SREC + (v * NV + i) * 64 + stk_lane] = x; };

    const RngKey key = make_key((uint64_t)a.seeds[sb], chain);
    const bool laplace = (model == 4);

    // ------------------------------------------------------------- state (as in k_nuts)
    V inv_e, zq, zp, zg;                                                // registers, live across leapfrogs
    CV qs, gs, pq, pp, pg, mq, mp, mg, rho, psp, psm, wmean, wm2, bq, bg;   // cold store (or registers, see CV)
    auto bind = [&](CV &x, int which, int ln) {
        if constexpr (std::is_same<CV, ColdV>::value) { x.v.b = cold + which * PMAX; x.v.lane = ln; }
    };
#define EPX_BIND_COLD(ln)                                                                              \
    bind(qs, CV_QS, ln); bind(gs, CV_GS, ln); bind(pq, CV_PQ, ln); bind(pp, CV_PP, ln); bind(pg, CV_PG, ln); \
    bind(mq, CV_MQ, ln); bind(mp, CV_MP, ln); bind(mg, CV_MG, ln); bind(rho, CV_RHO, ln);                 \
    bind(psp, CV_PSP, ln); bind(psm, CV_PSM, ln); bind(wmean, CV_WMEAN, ln); bind(wm2, CV_WM2, ln); \
    bind(bq, CV_BQ, ln); bind(bg, CV_BG, ln)
    EPX_BIND_COLD(lane0);
    double lps = 0, zlp = 0, plp = 0, mlp = 0, b_key = 0, b_plp = 0;
    FORV {
        inv_e.v[i] = 1.0;
        zq.v[i] = 0; zp.v[i] = 0; zg.v[i] = 0;
    }
    if (active) {
        const double *lastp = a.last + ((size_t)k * a.chains + chain) * a.P;
        FORV {
            const int e = lane0 + 64 * i;
            double q0 = 0.0;
            if (e < P) {
                if (a.init_mode == 2) q0 = lastp[e];
                else if (a.init_mode == 0) {
                    double u1, u2;
                    rng_u2(key, 0, K_INIT, (uint32_t)(e >> 1), 0, u1, u2);
                    q0 = -2.0 + 4.0 * ((e & 1) ? u2 : u1);
                }
            }
            if (resume) {
                // the sample and the Welford sums of the piece before this one
                q0 = ck_load(ckp + (0 * NV + i) * 64 + lane0);
                wmean.v[i] = ck_load(ckp + (1 * NV + i) * 64 + lane0);
                wm2.v[i] = ck_load(ckp + (2 * NV + i) * 64 + lane0);
            } else {
                wmean.v[i] = 0.0; wm2.v[i] = 0.0;
            }
            zq.v[i] = q0;
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
    double H0 = 0, lsw = 0, sum_metro = 0, eps_l = 0;
    double u_dir = 0.0, gum = 0.0;
    double dhb = 0.0, lw_m = -INFINITY, lw_s = 0.0;

    const bool teacher = a.eps_in != nullptr;
    if (teacher && active) {
        eps = a.eps_in[(size_t)sb * a.chains + chain];
        if (a.inv_e_in) {
            const double *ie = a.inv_e_in + ((size_t)sb * a.chains + chain) * a.P;
            FORV { const int e = lane0 + 64 * i; if (e < P) inv_e.v[i] = ie[e]; }
        }
    }
    // opt-in carried adaptation: last call's step size of the chain, the site's pooled sample variances
    const bool carry = active && !teacher && a.carry_eps != nullptr && a.carry_eps[(size_t)k * a.chains + chain] > 0.0;
    if (carry) {
        eps = a.carry_eps[(size_t)k * a.chains + chain];
        da_mu = log(10.0 * eps);
        const double *cm = a.carry_metric + (size_t)k * a.P;
        FORV { const int e = lane0 + 64 * i; if (e < P) inv_e.v[i] = cm[e]; }
    }
    int finished = active ? 0 : 1;
    double ck_mark = 0.0;                              // (a failed chain's scalars, handed on from boundary to boundary)
    if (resume && active) {
        FORV inv_e.v[i] = ck_load(ckp + (3 * NV + i) * 64 + lane0);
        const double ckv = ck_load(ckp + 4 * NV * 64 + lane0);
        ck_mark = ckv;
#define EPX_CK_GET(idx, x) ck_assign(x, readlane_d(ckv, idx));
        EPX_CK_LIST(EPX_CK_GET)
#undef EPX_CK_GET
        ngrad -= 1.0;                                 // the gradient at the restored sample is evaluated once more
        if (failed) finished = 1;                     // it failed in its first piece, where everything was written
    }
    const bool was_failed = resume && failed != 0;
    const uint32_t toff = (uint32_t)a.t_offset + 1u;

    auto flush_dh = [&](int cnt) {
        const bool ok = lane0 < cnt;
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

    int counted = is_chain ? 0 : 1;
#ifdef EPX_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        STAMP(4);
        // `lane` is re-derived through an opaque move every iteration: otherwise the compiler
        // hoists all per-element index arithmetic of the loop body (7 elements x dozens of
        // indices and predicates) out of the loop and spills it
        // (and again after the row pass, so that nothing index-like stays live across it)
        double kin = 0.0, sa = 0, lpt = 0.0, ll = 0.0;
        {
        int lane_v = lane0;
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v;
        stk_lane = lane;
#if EPX_STREAM_WAVE_SCALAR
        // (and the wave index as a scalar: `tid >> 6` is a vector value for the compiler, so every per-wave LDS base
        // (q_s + wave * PMAX, ...) was a vector register computed once, spilled, and reloaded from scratch in front
        // of its use -- one vector-memory round trip each, five in a row in step A)
        const int wave = __builtin_amdgcn_readfirstlane(wave0);
#endif
        // ---- lock step: leave only when every chain of the workgroup is done
        if (finished && !counted) { if (lane == 0) atomicAdd(sh_done, 1); counted = 1; }
        lds_barrier();
        if (*sh_done >= NCH) break;
        STAMP(5);

        // =================================================== leapfrog, all chains together
        if (is_chain) {
            FORV zp.v[i] += 0.5 * eps_l * zg.v[i];
            FORV zq.v[i] += eps_l * inv_e.v[i] * zp.v[i];
            // ---- step A (wave = chain): publish q, exp(q); alpha, beta, phi - mu
            double *qc = q_s + wave * PMAX, *eqc = eq_s + wave * PMAX;
            FORV { qc[lane + 64 * i] = zq.v[i]; eqc[lane + 64 * i] = exp_d(zq.v[i]); }
            // theta = [phi (d) | eta (ng) | etb (ng x D)]; per group j: alpha_j, beta_j (Appendix A of
            // SURVEY.md, m*b.stan); a0 + eta_j * sa and b0[c] + etb_j[c] * sbv[c]
            sa = eqc[O + (model >= 3 ? 1 : 0)];
            const double a0 = model >= 3 ? qc[O] : 0.0;
            if (O && lane == 0) L.is2_s[wave] = exp_d(-2.0 * qc[0]);          // 1 / sigma^2
            for (int g = 0; g < ng; ++g) {
#pragma unroll
                for (int b = 0; b < (DPB + 63) / 64; ++b) {
                    const int c = lane + 64 * b;
                    if (DPB < 64 && c >= DPB) continue;
                    double bj = 0.0;
                    if (c < D) {
                        const double eb = model == 0 ? 0.0 : qc[d + ng + g * D + c];
                        if (model == 0) bj = qc[O + 1 + c];
                        else if (model == 1) bj = eb * eqc[O + 1];
                        else if (model == 2) bj = eb * eqc[O + 1 + c];
                        else bj = qc[O + 2 + c] + eb * eqc[O + 2 + D + c];
                    }
                    L.beta_s[(g * DPB + c) * NCH + wave] = bj;
                }
                if (lane == 0) L.alpha_s[g * NCH + wave] = a0 + qc[d + g] * sa;
            }
            FORV { const int e = lane + 64 * i; if (e < d) vs4[e * NCH + wave] = zq.v[i] - mu_s[e]; }
        }
        STAMP(0);
        lds_barrier();
        {
            // ---- Omega (phi - mu) for all chains in one pass over Omega
            if constexpr (RES) {
                // d <= 66 here: wave = every 4th column, lane = row (two row groups), partial sums per wave
                const int r0 = lane < d ? lane : d - 1, r1 = lane + 64 < d ? lane + 64 : d - 1;
                double o[2][NCH] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                constexpr int CUO = 17;             // all of a wave's columns in flight at d <= 68
                auto omega_cols = [&](auto load) {
                    for (int j0 = wave; j0 < d; j0 += NCH * CUO) {
                        double om0[CUO], om1[CUO];
#pragma unroll
                        for (int u = 0; u < CUO; ++u) {
                            const int j = j0 + NCH * u < d ? j0 + NCH * u : d - 1;
                            om0[u] = load(j * d + r0);
                            om1[u] = load(j * d + r1);
                        }
#pragma unroll
                        for (int u = 0; u < CUO; ++u) {
                            const bool ok = j0 + NCH * u < d;
                            const int j = ok ? j0 + NCH * u : d - 1;
                            double2 v01 = *reinterpret_cast<const double2 *>(vs4 + j * NCH);
                            double2 v23 = *reinterpret_cast<const double2 *>(vs4 + j * NCH + 2);
                            if (!ok) { v01 = make_double2(0.0, 0.0); v23 = v01; }
                            o[0][0] = fma(om0[u], v01.x, o[0][0]); o[0][1] = fma(om0[u], v01.y, o[0][1]);
                            o[0][2] = fma(om0[u], v23.x, o[0][2]); o[0][3] = fma(om0[u], v23.y, o[0][3]);
                            o[1][0] = fma(om1[u], v01.x, o[1][0]); o[1][1] = fma(om1[u], v01.y, o[1][1]);
                            o[1][2] = fma(om1[u], v23.x, o[1][2]); o[1][3] = fma(om1[u], v23.y, o[1][3]);
                        }
                    }
                };
                // Omega from LDS when the site leaves room for it (small sites), else from L2
                if (a.om_in_lds) omega_cols([&](int idx) { return Om_s[idx]; });
                else omega_cols([&](int idx) { return Om_g[idx]; });
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = lane + 64 * h;
                    if (e < d) {
                        double *dst = opart + ((size_t)wave * d + e) * NCH;
                        *reinterpret_cast<double2 *>(dst) = make_double2(o[h][0], o[h][1]);
                        *reinterpret_cast<double2 *>(dst + 2) = make_double2(o[h][2], o[h][3]);
                    }
                }
            } else if (tid < d) {
                // thread = row.  The columns arrive through a RING of OM_UNROLL registers: a column is multiplied in and its
                // register is asked for the column OM_UNROLL further on at once, so OM_UNROLL loads per thread are in flight
                // all the time (round 3 asked for a batch, waited and multiplied: the batch drained before the next one was
                // requested, ~9 in flight on average, and the d % OM_UNROLL columns of the remainder loop each paid a whole
                // memory latency).  One running pointer; the requests of the last full round reach up to OM_UNROLL columns
                // beyond the matrix (the next site's, or the padding behind the last one: epx_api.hip) and are never used.
                // (the thread index goes through an opaque move: the addresses of the first OM_UNROLL requests do not change from
                // leapfrog to leapfrog, LLVM computes them once, spills them, and a scratch reload in front of the loop is a
                // vector-memory operation the loop's first wait then has to cover -- with every request behind it)
                double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
                int tid_o = tid;
                asm volatile("" : "+v"(tid_o));
                const double *omp = Om_g + tid_o;
                double om[OM_UNROLL];
#pragma unroll
                for (int u = 0; u < OM_UNROLL; ++u) { om[u] = *omp; omp += d; }
                int j = 0;
                for (; j + OM_UNROLL <= d; j += OM_UNROLL) {
#pragma unroll
                    for (int u = 0; u < OM_UNROLL; ++u) {
                        const double2 v01 = *reinterpret_cast<const double2 *>(vs4 + (j + u) * NCH);
                        const double2 v23 = *reinterpret_cast<const double2 *>(vs4 + (j + u) * NCH + 2);
                        o0 = fma(om[u], v01.x, o0); o1 = fma(om[u], v01.y, o1);
                        o2 = fma(om[u], v23.x, o2); o3 = fma(om[u], v23.y, o3);
                        om[u] = *omp; omp += d;
                    }
                }
#pragma unroll
                for (int u = 0; u < OM_UNROLL; ++u) {
                    if (j + u < d) {
                        const double2 v01 = *reinterpret_cast<const double2 *>(vs4 + (j + u) * NCH);
                        const double2 v23 = *reinterpret_cast<const double2 *>(vs4 + (j + u) * NCH + 2);
                        o0 = fma(om[u], v01.x, o0); o1 = fma(om[u], v01.y, o1);
                        o2 = fma(om[u], v23.x, o2); o3 = fma(om[u], v23.y, o3);
                    }
                }
                *reinterpret_cast<double2 *>(Ovs + tid * NCH) = make_double2(o0, o1);
                *reinterpret_cast<double2 *>(Ovs + tid * NCH + 2) = make_double2(o2, o3);
            }
        }
        lds_barrier();
        STAMP(1);
        if (is_chain) {
            // cavity part of the gradient and of lp
            FORV {
                const int e = lane + 64 * i;
                double g = 0.0;
                if (e < d) {
                    double ov;
                    if constexpr (RES)
                        ov = (opart[(0 * d + e) * NCH + wave] + opart[(1 * d + e) * NCH + wave])
                             + (opart[(2 * d + e) * NCH + wave] + opart[(3 * d + e) * NCH + wave]);
                    else
                        ov = Ovs[e * NCH + wave];
                    g = -ov; lpt += -0.5 * vs4[e * NCH + wave] * ov;
                }
                zg.v[i] = g;
            }
        }
        if constexpr (RES) {
            ll = resident_pass<DPB>(site.lds0, RM, site.ntile, a.ngmax, wave, lane, NT, tid);
        } else {
            const PassOut po = stream_pass<DPB>(site);
            site.slot_f = po.slot_f; site.slot_i = po.slot_i; site.t_i = po.t_i;
            ll = po.ll;
        }
        }
        STAMP(2);
        if (!is_chain) continue;
        int lane_w = lane0;
        asm volatile("" : "+v"(lane_w));
        const int lane = lane_w;
        stk_lane = lane;
#if EPX_STREAM_WAVE_SCALAR > 1
        const int wave = __builtin_amdgcn_readfirstlane(wave0);          // (as at the loop top: a scalar for step D and the books)
#endif
        EPX_BIND_COLD(lane);
        {
            // ---- step D (wave = chain): lp and the chain rule back to (phi, eta, etb)
            const double *qc = q_s + wave * PMAX, *eqc = eq_s + wave * PMAX;
            // G_g[c] = sum over the rows of group g of x[c] * residual, da_g = sum of the residuals
            auto Gg = [&](int g, int c) { return L.Gs[(g * DPB + c) * NCH + wave]; };
            auto dag = [&](int g) { return L.da_s[g * NCH + wave]; };
            double dot = 0.0;
            if (model == 1) {
                // m2b: d log sigma_b = sb * sum_g sum_c G_g[c] etb_g[c]
                double tsum = 0.0;
                FORV {
                    const int e = lane + 64 * i;
                    if (e >= d + ng && e < P) { const int idx = e - d - ng; tsum += Gg(idx / D, idx % D) * zq.v[i]; }
                }
                dot = wave_sum(tsum);
            }
            // Gaussian likelihood: the pass returned sum -(y - f)^2 / (2 sigma^2); d/d log sigma = sum (y - f)^2 / sigma^2 - n
            // and lp gets -n log sigma (as in the resident kernels, nuts_gradient.inc)
            double c_ls = 0.0;
            if (O) { c_ls = -2.0 * ll - (double)site.n; ll -= (double)site.n * qc[0]; }
            // sums over the groups that the shared coordinates need
            double s_da = 0.0, s_daeta = 0.0;
            for (int g = 0; g < ng; ++g) { const double t = dag(g); s_da += t; s_daeta += t * qc[d + g]; }
            FORV {
                const int e = lane + 64 * i;
                const double q = zq.v[i];
                double g = zg.v[i];                 // cavity part (elements < d), 0 beyond
                if (e >= d && e < P) lpt -= laplace ? fabs(q) : 0.5 * q * q;
                const double pr = laplace ? (double)((q > 0) - (q < 0)) : q;
                if (e < d) {
                    // hyper-parameters: every group contributes
                    const int isa = model >= 3 ? 1 : 0;
                    const int eb = e - O;                                   // index in the b-model's phi
                    if (O && e == 0) g += c_ls;                             // log sigma
                    else if (model >= 3 && eb == 0) g += s_da;
                    else if (eb == isa) g += s_daeta * sa;
                    else if (model == 1) { if (eb == 1) g += dot * eqc[O + 1]; }
                    else {
                        // slope block(s): m1b beta (1..D); m3b log sigma_b (1..D); m4b/m5b mu_b (2..1+D), log sigma_b (2+D..)
                        const int c = model >= 3 ? (eb < 2 + D ? eb - 2 : eb - 2 - D) : eb - 1;
                        const bool scale = model == 2 || (model >= 3 && eb >= 2 + D);
                        double acc = 0.0;
                        for (int gg = 0; gg < ng; ++gg)
                            acc += scale ? Gg(gg, c) * qc[d + ng + gg * D + c] : Gg(gg, c);
                        g += scale ? acc * eqc[e] : acc;
                    }
                } else if (e < d + ng) {
                    g = dag(e - d) * sa - pr;                               // eta_g
                } else if (e < P) {
                    const int idx = e - d - ng, gg = idx / D, c = idx - gg * D;   // etb_g[c]
                    const double sbv = model == 1 ? eqc[O + 1] : (model == 2 ? eqc[O + 1 + c] : eqc[O + 2 + D + c]);
                    g = Gg(gg, c) * sbv - pr;
                }
                zg.v[i] = e < P ? g : 0.0;
            }
            double ks = 0.0;
            FORV { zp.v[i] += 0.5 * eps_l * zg.v[i]; ks += inv_e.v[i] * zp.v[i] * zp.v[i]; }
            wave_sum2(lpt, ks);
            zlp = lpt + ll;
            kin = 0.5 * ks;
        }
        STAMP(3);
        if (finished) continue;          // idle chain slots only take part in the shared work
        ngrad += 1.0;

        // the new-subtree vectors never outlive one iteration (a leaf is merged, then parked on
        // the stack or consumed), so they are iteration-local: nothing to keep across the row pass
        V n_rho, n_psl, psr;
#define EPX_CHAIN_EXIT { finished = 1; eps_l = 0.0; continue; }
#define EPX_DBG_EXIT { finished = 1; eps_l = 0.0; continue; }
#define STAMP_LEAF STAMP(6)
#define EPX_RESUME resume
#define EPX_T_END t_end
#include "nuts_state_machine.inc"
#undef EPX_RESUME
#undef EPX_T_END
#undef STAMP_LEAF
#undef EPX_CHAIN_EXIT
#undef EPX_DBG_EXIT
    }
    if constexpr (!RES) { if (wave == NCH) wait_vm<0>(); }      // drain the prefetched tiles before the LDS goes away

#ifdef EPX_STAMPS
    if (a.stamps && wave == 0 && lane0 == 0) {
        // (looping workgroups: a record per piece, numbered by a counter behind the records)
        const size_t nrec = PIECED ? (size_t)a.seg_nwg : (size_t)gridDim.x;
        const size_t rec = (PIECED && a.persist) ? (size_t)atomicAdd(a.stamps + 3 * nrec * 8 + 15, 1ull) : (size_t)blockIdx.x;
        for (int i = 0; i < 7; ++i) a.stamps[rec * 8 + i] = tacc[i];
        a.stamps[rec * 8 + 7] = (unsigned long long)ngrad;
        // second record: the piece's timeline (entry, claim, end of sampling; site, first transition, leapfrogs)
        unsigned long long *tl = a.stamps + (nrec + rec) * 8;
        tl[0] = tl_entry; tl[1] = tl_claim; tl[2] = __builtin_amdgcn_s_memrealtime();
        tl[3] = (unsigned long long)sb; tl[4] = (unsigned long long)t_begin; tl[5] = (unsigned long long)ngrad;
        // which CU ran it: HW_ID (wave / simd / pipe / cu / sh / se fields) and XCC_ID, as the hardware registers read
        tl[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        tl[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    }
#endif
    // ------------------------------------------------------------- epilogue
    if constexpr (PIECED) {
        if (active && !a.dbg && !was_failed) {
            // checkpoint at the transition boundary: the sample, the Welford sums, the metric and the scalars of
            // EPX_CK_LIST (the gradient at the sample is re-evaluated by the piece that continues)
            FORV {
                ck_store(ckp_out + (0 * NV + i) * 64 + lane0, qs.v[i]);
                ck_store(ckp_out + (1 * NV + i) * 64 + lane0, wmean.v[i]);
                ck_store(ckp_out + (2 * NV + i) * 64 + lane0, wm2.v[i]);
                ck_store(ckp_out + (3 * NV + i) * 64 + lane0, inv_e.v[i]);
            }
            double ckv = 0.0;
#define EPX_CK_PUT(idx, x) ckv = lane0 == (idx) ? (double)(x) : ckv;
            EPX_CK_LIST(EPX_CK_PUT)
#undef EPX_CK_PUT
            ck_store(ckp_out + 4 * NV * 64 + lane0, ckv);
            piece_checkpoint_out();                                 // the record is out before the site is put back
        }
        if (active && !a.dbg && was_failed) { ck_store(ckp_out + 4 * NV * 64 + lane0, ck_mark); piece_checkpoint_out(); }
        __syncthreads();
        if (threadIdx.x == 0) piece_release(a, smem);
    }
    if (active && !a.dbg && !was_failed && (failed || t >= a.iter)) {
        double *lastp = a.last + ((size_t)k * a.chains + chain) * a.P;
        FORV { const int e = lane0 + 64 * i; if (e < a.P) lastp[e] = qs.v[i]; }
        if (failed) {
            for (int kk = 0; kk < a.nkeep; ++kk) {
                double *dst = a.draws + (((size_t)k * a.chains + chain) * a.nkeep + kk) * a.P;
                FORV { const int e = lane0 + 64 * i; if (e < a.P) dst[e] = qs.v[i]; }
            }
        }
        if (lane0 == 0) {
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

#ifdef EPX_STAMPS
#define EPX_TL_NOW() __builtin_amdgcn_s_memrealtime()       /* (100 MHz, the same clock on every CU) */
#else
#define EPX_TL_NOW() 0ull
#endif

// PIECED: the launch has one workgroup per piece of a site's transitions (epx_pieces.h); a template parameter so that
// the plain kernel stays what it was
template <int NV, int DPB, bool RES, bool PIECED>
__global__ void __launch_bounds__(RES ? 256 : STREAM_THREADS)
k_nuts_stream(NutsArgs a_by_value) {
    extern __shared__ __align__(16) unsigned char smem[];
    (void)a_by_value;
    StreamArgsK *kargs_p = (StreamArgsK *)__builtin_amdgcn_kernarg_segment_ptr();
    int q_site = -1, q_t0 = 0;
    const unsigned long long tl_entry = EPX_TL_NOW();
    if constexpr (PIECED) {
        if (piece_claim(*kargs_p, smem, (int)threadIdx.x, q_site, q_t0) <= 0) {
            if (threadIdx.x == 0) atomicOr(kargs_p->err, 4);
            return;
        }
    }
    stream_piece<NV, DPB, RES, PIECED>(kargs_p, q_site, q_t0, tl_entry, EPX_TL_NOW());
}

// The pieced launch with LOOPING workgroups (NutsArgs::persist): as many workgroups as the device holds at a time, each
// claiming pieces until no site has anything left.  With one workgroup per piece the in-order dispatcher (workgroup i goes
// to XCD i % 8, and the next one waits for a CU of ITS XCD while CUs of the others are free) left ~10 % of the CU-time
// of a C5-shard launch unused (profiles/r03_stream_piece_timeline.json; scripts/probe/dispatch_gaps.hip shows the effect
// with no sampler in it): 15.0 -> 15.8 site-updates/s on one box.  The piece's body is a real call: inlined into the loop
// it would be compiled with everything live around it.
template <int NV, int DPB>
__device__ __attribute__((noinline)) void stream_piece_call(unsigned long long kargs_u, int q_site, int q_t0, unsigned long long tl_entry, unsigned long long tl_claim) {
    // (arguments of a call travel in VECTOR registers and count as divergent: the pointer to the kernel arguments is made
    // scalar again -- through a vector pointer every a.field would be a vector load -- and so is the site.
    // __builtin_amdgcn_kernarg_segment_ptr() is null inside a called function)
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)kargs_u), hi = __builtin_amdgcn_readfirstlane((unsigned)(kargs_u >> 32));
    StreamArgsK *kargs_p = (StreamArgsK *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    stream_piece<NV, DPB, false, true>(kargs_p, __builtin_amdgcn_readfirstlane(q_site), __builtin_amdgcn_readfirstlane(q_t0), tl_entry, tl_claim);
}

template <int NV, int DPB>
__global__ void __launch_bounds__(STREAM_THREADS)
k_nuts_stream_loop(NutsArgs a_by_value) {
    extern __shared__ __align__(16) unsigned char smem[];
    (void)a_by_value;
    StreamArgsK *kargs_p = (StreamArgsK *)__builtin_amdgcn_kernarg_segment_ptr();
    for (;;) {
        int q_site = -1, q_t0 = 0;
        const unsigned long long tl_entry = EPX_TL_NOW();
        const int got = piece_claim(*kargs_p, smem, (int)threadIdx.x, q_site, q_t0);
        if (got <= 0) {
            if (got < 0 && threadIdx.x == 0) atomicOr(kargs_p->err, 4);
            return;
        }
        stream_piece_call<NV, DPB>((unsigned long long)(uintptr_t)kargs_p, q_site, q_t0, tl_entry, EPX_TL_NOW());
        __syncthreads();                                 // (the next claim's scratch is the LDS this piece used)
    }
}

// LDS bytes of the streaming kernel
size_t nuts_stream_lds_bytes(int nv, int dpb, int d, int ngmax, int ntmax, int nmax_res, int gauss) {
    const size_t pmax = 64 * (size_t)nv;
    size_t eng;
    size_t dbl = (size_t)((d + 1) & ~1) + 2 * (size_t)d * NCH + 2 * NCH * pmax + 2;
    if (nmax_res > 0) {             // resident variant: dpb in {16, 32}; per-wave Omega partials instead of Ovs
        eng = dpb == 16 ? res_map<16>(nmax_res, ngmax, ntmax).end : res_map<32>(nmax_res, ngmax, ntmax).end;
        dbl += (size_t)NCH * d * NCH - (size_t)d * NCH;
    } else
        eng = dpb == 64 ? stream_map<64>(ngmax, ntmax, gauss).end : stream_map<128>(ngmax, ntmax, gauss).end;
    return eng + dbl * 8;
}
// doubles of global memory per chain: tree stack + cold store
size_t nuts_stream_chain_doubles(int nv, int max_depth) {
    return (size_t)max_depth * nuts_stack_record(nv) + (size_t)CV_COUNT * 64 * nv;
}

template <int NV, int DPB, bool RES>
static int launch_stream_one(const NutsArgs &a, int nblocks, size_t lds, hipStream_t stream) {
    auto kern = k_nuts_stream<NV, DPB, RES, false>;
    if constexpr (!RES) {
        if (a.dyn_prog) {
            kern = k_nuts_stream<NV, DPB, RES, true>; nblocks = a.seg_nwg;
            if (a.persist) {
                // looping workgroups: as many as the device holds at a time (never more than there are pieces)
                kern = k_nuts_stream_loop<NV, DPB>;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return (int)e;
                int per_cu = 0, dev = 0, ncu = 0;
                e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), STREAM_THREADS, lds);
                if (e != hipSuccess) return (int)e;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
                const int hold = (per_cu > 0 ? (per_cu < 8 ? per_cu : 8) : 1) * (ncu > 0 ? ncu : 1);      // (the host sized the workgroups' private memory for at most 8 per CU)
                if (nblocks > hold) nblocks = hold;
            }
        }
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(RES ? 256 : STREAM_THREADS), lds, stream, a);
    return (int)hipGetLastError();
}

template <int DPB, bool RES>
static int launch_stream_nv(const NutsArgs &a, int nblocks, int nv, size_t lds, hipStream_t stream) {
    switch (nv) {
#ifndef EPX_STREAM_MIN      // (scripts/build_stream_variant.sh: A/B builds of the C5 shape only -- NV = 7, DPB = 128 -- in seconds)
    case 1: return launch_stream_one<1, DPB, RES>(a, nblocks, lds, stream);
    case 2: return launch_stream_one<2, DPB, RES>(a, nblocks, lds, stream);
    case 3: return launch_stream_one<3, DPB, RES>(a, nblocks, lds, stream);
    case 4: return launch_stream_one<4, DPB, RES>(a, nblocks, lds, stream);
    case 5: return launch_stream_one<5, DPB, RES>(a, nblocks, lds, stream);
    case 6: return launch_stream_one<6, DPB, RES>(a, nblocks, lds, stream);
#endif
    case 7: return launch_stream_one<7, DPB, RES>(a, nblocks, lds, stream);
    }
    return -1;
}

// count sites; streaming: dpb in {64, 128}; resident (a.n_max rows in LDS): dpb in {16, 32};
// nv = ceil(P / 64) <= 7; tree stack and cold store live in a.stack
int launch_nuts_stream(const NutsArgs &a, int count, int dpb, int nv, hipStream_t stream) {
    const int bps = (a.chains + NCH - 1) / NCH;
    const int nblocks = count * bps;
    const size_t lds = (size_t)a.lds_bytes;          // (the host laid it out: nuts_stream_lds_bytes + the piece words)
#ifndef EPX_STREAM_MIN
    if (dpb == 16) return launch_stream_nv<16, true>(a, nblocks, nv, lds, stream);
    if (dpb == 32) return launch_stream_nv<32, true>(a, nblocks, nv, lds, stream);
    if (dpb == 64) return launch_stream_nv<64, false>(a, nblocks, nv, lds, stream);
#endif
    if (dpb == 128) return launch_stream_nv<128, false>(a, nblocks, nv, lds, stream);
    return -1;
}

}  // namespace epx
