// Streaming variant of the on-GPU NUTS sampler for sites whose rows do NOT fit LDS
// (BASELINE config C5: D = 128, n_j = 2000 -> 2 MB of X per site, d = 258, P = 387).
//
// Same algorithm as k_nuts (the tree bookkeeping / adaptation code is the shared include
// nuts_state_machine.inc); what changes is the gradient, which is HBM-bound here
// (0.5 flop per byte of X):
//   * one workgroup = one site, wave c = chain c, and the (up to 4) chains advance in
//     LOCK STEP: every loop iteration is one leapfrog of every chain, so the site's rows
//     are streamed from HBM ONCE per leapfrog for all chains (n_j*D*8 bytes) instead of
//     once per chain;
//   * X goes through a 64-row LDS tile (register-staged prefetch of the next tile while the
//     current one is consumed); per tile the 4 waves split the COLUMNS for the forward
//     product F = X B (lane = row, 4 chains per lane) and for the backward product
//     G += X' (y - sigmoid F) (lane = column, rows split over lane groups), and split the
//     CHAINS for the logistic terms;
//   * the cavity term Omega (phi - mu) is one pass over Omega (d*d*8 bytes from HBM/L2) for
//     all chains (thread = row of Omega);
//   * per-chain vectors (P up to 448) stay in registers, element e in lane e%64, register e/64;
//     the parameter transforms gather through LDS copies of q and exp(q).
// Algorithmic HBM bytes per leapfrog of one site: n_j*D*8 + n_j + d*d*8 (X, y, Omega), for
// min(chains, 4) gradients.
#include "epx_device.h"
#include "epx_kernels.h"

namespace epx {

template <int NV> struct VecS { double v[NV]; };
#define FORV _Pragma("unroll") for (int i = 0; i < NV; ++i)
#define STAMP(i) do { } while (0)

enum { SMODE_INIT = 0, SMODE_SS = 1, SMODE_TREE = 2 };
#define MODE_INIT SMODE_INIT
#define MODE_SS SMODE_SS
#define MODE_TREE SMODE_TREE

constexpr int TR = 64;          // rows per LDS tile
constexpr int NCH = 4;          // chain slots (waves) per workgroup

template <int NV, int DPB>
__global__ void __launch_bounds__(256)
k_nuts_stream(NutsArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    using V = VecS<NV>;
    constexpr int DW = DPB / 4;                 // columns per wave
    constexpr int NH = 64 / DW;                 // lane groups over the rows in the backward pass
    constexpr int XS = DPB + 1;                 // padded row stride of the tile (odd: conflict free)
    constexpr int SREC = 4 * NV * 64 + 2;       // per-level stack record (doubles)
    constexpr int PMAX = 64 * NV;
    constexpr int NPRE = TR * DPB / 256;        // doubles of the next tile staged per thread

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = 0;                           // one wave per chain
    const int bps = (a.chains + NCH - 1) / NCH;
    const int sb = blockIdx.x / bps, cb = blockIdx.x % bps;
    const int k = a.k0 + sb;
    const int chain = cb * NCH + wave;
    const bool active = chain < a.chains;
    const int D = a.D, d = a.d, P = a.P, model = a.model;
    const int64_t row0 = a.k_lim[k];
    const int n = (int)(a.k_lim[k + 1] - row0);
    const int ntile = (n + TR - 1) / TR;

    // ---- LDS carve-up (doubles)
    double *Xt = reinterpret_cast<double *>(smem);                 // TR x XS
    double *beta_s = Xt + TR * XS;                                 // DPB x 4
    double *part = beta_s + DPB * NCH;                             // 4 waves x 4 chains x TR
    double *gs4 = part + 4 * NCH * TR;                             // TR x 4
    double *Gs = gs4 + TR * NCH;                                   // DPB x 4
    double *vs4 = Gs + DPB * NCH;                                  // d x 4 (padded to PMAX)
    double *Ovs = vs4 + PMAX * NCH;                                // d x 4
    double *q_s = Ovs + PMAX * NCH;                                // 4 x PMAX
    double *eq_s = q_s + NCH * PMAX;                               // 4 x PMAX
    double *alpha_s = eq_s + NCH * PMAX;                           // 4 (+ pad)
    int *sh_done = reinterpret_cast<int *>(alpha_s + 8);
    if (tid == 0) *sh_done = 0;

    const double *Xg = a.X + (size_t)row0 * D;
    const uint8_t *yg = a.y + row0;
    const double *Om_g = a.cav_Om + (size_t)k * d * d;
    double *stk_g = a.stack + ((size_t)sb * a.chains + (active ? chain : 0)) * a.max_depth * SREC;
    auto ld_stk = [&](int off) -> double { return stk_g[off]; };
    auto st_stk = [&](int off, double v) { stk_g[off] = v; };

    const RngKey key = make_key((uint64_t)a.seeds[sb], chain);
    const bool laplace = (model == 4);

    // ------------------------------------------------------------- state (as in k_nuts)
    V mu, inv_e, qs, gs, zq, zp, zg, pq, pp, pg, mq, mp, mg, rho, psp, psm;
    V n_rho, n_psl, n_pq, n_pg, psr, wmean, wm2;
    double lps = 0, zlp = 0, plp = 0, mlp = 0, n_key = 0, n_plp = 0;
    FORV {
        const int e = lane + 64 * i;
        mu.v[i] = e < d ? a.cav_mu[(size_t)k * d + e] : 0.0;
        inv_e.v[i] = 1.0;
        wmean.v[i] = 0.0; wm2.v[i] = 0.0;
        gs.v[i] = 0; zq.v[i] = 0; zp.v[i] = 0; zg.v[i] = 0; pq.v[i] = 0; pp.v[i] = 0; pg.v[i] = 0;
        mq.v[i] = 0; mp.v[i] = 0; mg.v[i] = 0; rho.v[i] = 0; psp.v[i] = 0; psm.v[i] = 0;
        n_rho.v[i] = 0; n_psl.v[i] = 0; n_pq.v[i] = 0; n_pg.v[i] = 0; psr.v[i] = 0;
    }
    if (active) {
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
    } else { FORV qs.v[i] = 0.0; }
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
    int t = 0, mode = MODE_INIT, depth = 0, leaf = 0, nleaf = 1, fwd = 1, nleap = 0, divergent = 0;
    int ss_trial = 0, ss_dir = 0, ss_after_update = 0;
    uint32_t ss_t = 0;
    double H0 = 0, lsw = 0, sum_metro = 0, eps_l = 0;
    double u_dir = 0.0, gum = 0.0;
    double dhb = 0.0, lw_m = -INFINITY, lw_s = 0.0;

    FORV { zq.v[i] = qs.v[i]; }
    const bool teacher = a.eps_in != nullptr;
    if (teacher && active) {
        eps = a.eps_in[(size_t)sb * a.chains + chain];
        if (a.inv_e_in) {
            const double *ie = a.inv_e_in + ((size_t)sb * a.chains + chain) * P;
            FORV { const int e = lane + 64 * i; if (e < P) inv_e.v[i] = ie[e]; }
        }
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

    // stage tile `tt` of X from HBM into registers (zero padded rows / columns)
    auto load_tile = [&](int tt, double *pre) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = u * 256 + tid;                  // element of the TR x DPB tile
            const int r = idx / DPB, c = idx - r * DPB;
            const int row = tt * TR + r;
            pre[u] = (row < n && c < D) ? Xg[(size_t)row * D + c] : 0.0;
        }
    };
    auto store_tile = [&](const double *pre) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / DPB, c = idx - r * DPB;
            Xt[r * XS + c] = pre[u];
        }
    };

    int finished = active ? 0 : 1, counted = 0;
    for (;;) {
        // ---- lock step: leave only when every chain of the workgroup is done
        if (finished && !counted) { if (lane == 0) atomicAdd(sh_done, 1); counted = 1; }
        __syncthreads();
        if (*sh_done >= NCH) break;

        // =================================================== leapfrog, all chains together
        double kin = 0.0;
        FORV zp.v[i] += 0.5 * eps_l * zg.v[i];
        FORV zq.v[i] += eps_l * inv_e.v[i] * zp.v[i];
        V eq;
        FORV eq.v[i] = exp_d(zq.v[i]);
        double sa = 0, eta = 0, sb2 = 0;
        {
            // ---- step A (wave = chain): publish q, exp(q); alpha, beta, phi - mu
            double *qc = q_s + wave * PMAX, *eqc = eq_s + wave * PMAX;
            FORV { qc[lane + 64 * i] = zq.v[i]; eqc[lane + 64 * i] = eq.v[i]; }
            double alpha;
            if (model == 0) { sa = eqc[0]; eta = qc[d]; alpha = eta * sa; }
            else if (model == 1) { sa = eqc[0]; sb2 = eqc[1]; eta = qc[2]; alpha = eta * sa; }
            else if (model == 2) { sa = eqc[0]; eta = qc[d]; alpha = eta * sa; }
            else { sa = eqc[1]; eta = qc[d]; alpha = qc[0] + eta * sa; }
#pragma unroll
            for (int b = 0; b < DPB / 64; ++b) {
                const int j = lane + 64 * b;
                double bj = 0.0;
                if (j < D) {
                    if (model == 0) bj = qc[1 + j];
                    else if (model == 1) bj = qc[3 + j] * sb2;
                    else if (model == 2) bj = qc[d + 1 + j] * eqc[1 + j];
                    else bj = qc[2 + j] + qc[d + 1 + j] * eqc[2 + D + j];
                }
                beta_s[j * NCH + wave] = bj;
            }
            FORV { const int e = lane + 64 * i; if (e < d) vs4[e * NCH + wave] = zq.v[i] - mu.v[i]; }
            if (lane == 0) alpha_s[wave] = alpha;
        }
        double ll = 0.0, da = 0.0;
        {
            // ---- step B: stream the rows once for all chains
            double acc[NCH] = {0.0, 0.0, 0.0, 0.0};
            double pre[NPRE];
            load_tile(0, pre);
            store_tile(pre);
            __syncthreads();
            const double alpha_c = alpha_s[wave];
            for (int tt = 0; tt < ntile; ++tt) {
                if (tt + 1 < ntile) load_tile(tt + 1, pre);            // HBM loads in flight during the tile
                // forward partials: this wave's columns, lane = row, 4 chains per lane
                {
                    double pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0;
                    const double *xr = Xt + lane * XS + wave * DW;
                    const double *bp = beta_s + (wave * DW) * NCH;
#pragma unroll 8
                    for (int dd = 0; dd < DW; ++dd) {
                        const double x = xr[dd];
                        const double2 b01 = *reinterpret_cast<const double2 *>(bp + dd * NCH);
                        const double2 b23 = *reinterpret_cast<const double2 *>(bp + dd * NCH + 2);
                        pf0 = fma(x, b01.x, pf0); pf1 = fma(x, b01.y, pf1);
                        pf2 = fma(x, b23.x, pf2); pf3 = fma(x, b23.y, pf3);
                    }
                    double *pw = part + (wave * NCH) * TR + lane;
                    pw[0 * TR] = pf0; pw[1 * TR] = pf1; pw[2 * TR] = pf2; pw[3 * TR] = pf3;
                }
                __syncthreads();
                // logistic terms: this wave's chain, lane = row
                {
                    const int row = tt * TR + lane;
                    double f = alpha_c;
#pragma unroll
                    for (int w = 0; w < 4; ++w) f += part[(w * NCH + wave) * TR + lane];
                    double l = 0.0, g = 0.0;
                    if (row < n) logistic_terms(f, (double)yg[row], l, g);
                    ll += l; da += g;
                    gs4[lane * NCH + wave] = g;
                }
                __syncthreads();
                // backward: this wave's columns, lane = (column, row group), 4 chains per lane
                {
                    const int dl = lane % DW, h = lane / DW;
                    const double *xc = Xt + wave * DW + dl;
#pragma unroll 4
                    for (int r = h; r < TR; r += NH) {
                        const double x = xc[r * XS];
                        const double2 g01 = *reinterpret_cast<const double2 *>(gs4 + r * NCH);
                        const double2 g23 = *reinterpret_cast<const double2 *>(gs4 + r * NCH + 2);
                        acc[0] = fma(x, g01.x, acc[0]); acc[1] = fma(x, g01.y, acc[1]);
                        acc[2] = fma(x, g23.x, acc[2]); acc[3] = fma(x, g23.y, acc[3]);
                    }
                }
                __syncthreads();
                if (tt + 1 < ntile) { store_tile(pre); __syncthreads(); }
            }
            // fold the row groups, publish G[column][chain]
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if constexpr (DW <= 16) acc[c] += partner_d<4>(acc[c], lane);
                acc[c] += partner_d<5>(acc[c], lane);
            }
            if (lane < DW) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) Gs[(wave * DW + lane) * NCH + c] = acc[c];
            }
            wave_sum2(da, ll);
        }
        {
            // ---- step C: Omega (phi - mu) for all chains in one pass over Omega (thread = row)
#pragma unroll
            for (int rr = 0; rr < (PMAX + 255) / 256; ++rr) {
                const int irow = tid + 256 * rr;
                if (irow < d) {
                    double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
                    int j = 0;
                    for (; j + 4 <= d; j += 4) {
                        double om[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) om[u] = Om_g[(size_t)(j + u) * d + irow];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double2 v01 = *reinterpret_cast<const double2 *>(vs4 + (j + u) * NCH);
                            const double2 v23 = *reinterpret_cast<const double2 *>(vs4 + (j + u) * NCH + 2);
                            o0 = fma(om[u], v01.x, o0); o1 = fma(om[u], v01.y, o1);
                            o2 = fma(om[u], v23.x, o2); o3 = fma(om[u], v23.y, o3);
                        }
                    }
                    for (; j < d; ++j) {
                        const double om = Om_g[(size_t)j * d + irow];
                        o0 = fma(om, vs4[j * NCH], o0); o1 = fma(om, vs4[j * NCH + 1], o1);
                        o2 = fma(om, vs4[j * NCH + 2], o2); o3 = fma(om, vs4[j * NCH + 3], o3);
                    }
                    Ovs[irow * NCH + 0] = o0; Ovs[irow * NCH + 1] = o1;
                    Ovs[irow * NCH + 2] = o2; Ovs[irow * NCH + 3] = o3;
                }
            }
        }
        __syncthreads();
        {
            // ---- step D (wave = chain): lp and the chain rule back to (phi, eta, etb)
            const double *qc = q_s + wave * PMAX, *eqc = eq_s + wave * PMAX;
            auto dbat = [&](int j) { return (j >= 0 && j < D) ? Gs[j * NCH + wave] : 0.0; };
            auto gq = [&](int e) { return (e >= 0 && e < P) ? qc[e] : 0.0; };
            auto geq = [&](int e) { return (e >= 0 && e < P) ? eqc[e] : 0.0; };
            double dot = 0.0;
            if (model == 1) {
                double tsum = 0.0;
                FORV { const int e = lane + 64 * i; if (e >= 3 && e < P) tsum += dbat(e - 3) * zq.v[i]; }
                dot = wave_sum(tsum);
            }
            double lpt = 0.0;
            FORV {
                const int e = lane + 64 * i;
                const double q = zq.v[i];
                double g = 0.0;
                if (e < d) {
                    const double ov = Ovs[e * NCH + wave];
                    g = -ov; lpt += -0.5 * (q - mu.v[i]) * ov;
                } else if (e < P) lpt -= laplace ? fabs(q) : 0.5 * q * q;
                const double pr = laplace ? (double)((q > 0) - (q < 0)) : q;
                if (model == 0) {
                    if (e == 0) g += da * eta * sa;
                    else if (e <= D) g += dbat(e - 1);
                    else if (e == d) g = da * sa - pr;
                } else if (model == 1) {
                    if (e == 0) g += da * eta * sa;
                    else if (e == 1) g += dot * sb2;
                    else if (e == 2) g = da * sa - pr;
                    else if (e < P) g = dbat(e - 3) * sb2 - pr;
                } else if (model == 2) {
                    const int j = e <= D ? e - 1 : e - d - 1;
                    const double db = dbat(j);
                    if (e == 0) g += da * eta * sa;
                    else if (e <= D) g += db * gq(d + 1 + j) * eq.v[i];
                    else if (e == d) g = da * sa - pr;
                    else if (e < P) g = db * geq(1 + j) - pr;
                } else {
                    const int j = e < 2 + D ? e - 2 : (e < d ? e - 2 - D : e - d - 1);
                    const double db = dbat(j);
                    if (e == 0) g += da;
                    else if (e == 1) g += da * eta * sa;
                    else if (e < 2 + D) g += db;
                    else if (e < d) g += db * gq(d + 1 + j) * eq.v[i];
                    else if (e == d) g = da * sa - pr;
                    else if (e < P) g = db * geq(2 + D + j) - pr;
                }
                zg.v[i] = e < P ? g : 0.0;
            }
            double ks = 0.0;
            FORV { zp.v[i] += 0.5 * eps_l * zg.v[i]; ks += inv_e.v[i] * zp.v[i] * zp.v[i]; }
            wave_sum2(lpt, ks);
            zlp = lpt + ll;
            kin = 0.5 * ks;
        }
        if (finished) continue;          // idle chain slots only take part in the shared work
        ngrad += 1.0;

#define EPX_CHAIN_EXIT { finished = 1; eps_l = 0.0; continue; }
#define EPX_DBG_EXIT { finished = 1; eps_l = 0.0; continue; }
#include "nuts_state_machine.inc"
#undef EPX_CHAIN_EXIT
#undef EPX_DBG_EXIT
    }

    // ------------------------------------------------------------- epilogue
    if (active && !a.dbg) {
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

// LDS bytes of the streaming kernel
size_t nuts_stream_lds_bytes(int nv, int dpb) {
    const size_t pmax = 64 * (size_t)nv;
    size_t dbl = (size_t)TR * (dpb + 1) + (size_t)dpb * NCH + 4 * NCH * TR + (size_t)TR * NCH + (size_t)dpb * NCH
                 + 2 * pmax * NCH + 2 * NCH * pmax + 8 + 2;
    return dbl * 8;
}

template <int NV, int DPB>
static int launch_stream_one(const NutsArgs &a, int nblocks, size_t lds, hipStream_t stream) {
    auto kern = k_nuts_stream<NV, DPB>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, stream, a);
    return (int)hipGetLastError();
}

template <int DPB>
static int launch_stream_nv(const NutsArgs &a, int nblocks, int nv, size_t lds, hipStream_t stream) {
    switch (nv) {
    case 1: return launch_stream_one<1, DPB>(a, nblocks, lds, stream);
    case 2: return launch_stream_one<2, DPB>(a, nblocks, lds, stream);
    case 3: return launch_stream_one<3, DPB>(a, nblocks, lds, stream);
    case 4: return launch_stream_one<4, DPB>(a, nblocks, lds, stream);
    case 5: return launch_stream_one<5, DPB>(a, nblocks, lds, stream);
    case 6: return launch_stream_one<6, DPB>(a, nblocks, lds, stream);
    case 7: return launch_stream_one<7, DPB>(a, nblocks, lds, stream);
    }
    return -1;
}

// count sites; dpb in {64, 128}; nv = ceil(P / 64) <= 7; the tree stack lives in a.stack
int launch_nuts_stream(const NutsArgs &a, int count, int dpb, int nv, hipStream_t stream) {
    const int bps = (a.chains + NCH - 1) / NCH;
    const int nblocks = count * bps;
    const size_t lds = nuts_stream_lds_bytes(nv, dpb);
    if (dpb == 64) return launch_stream_nv<64>(a, nblocks, nv, lds, stream);
    if (dpb == 128) return launch_stream_nv<128>(a, nblocks, nv, lds, stream);
    return -1;
}

}  // namespace epx
