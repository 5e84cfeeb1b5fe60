// Layout 8 ("quad"): four waves per workgroup, ONE per SIMD; wave c is a quarter of the site's row team AND the integrator
// AND the bookkeeper of chain c (nuts_quad.hip; the probe scripts/probe/quad_pass.hip times the same pass).
//
// This header holds what the kernel and the probe share: the row phase of a pass -- layout 7's (nuts_duo.hip, TEAM form),
// operation by operation: every wave takes its quarter of the site's 16-row tiles through  F = alpha + X B
// (v_mfma_f64_4x4x4: 16 rows x 4 chains per instruction), the logistic terms on the products' own lanes and  G += X' g,
// plus its 16-row group of the cavity term  Omega V  -- and the layout of a chain's LDS slot.
// Replaces the log-density evaluations inside the Stan subprocess of /root/reference/epstan/method.py:349-363
// (model: /root/reference/experiment/models/m4b_sg.stan:19-43).
#pragma once
#include "nuts_common.h"

namespace epx {

typedef double q8_v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) q8_v2 *q8_v2p;
typedef __attribute__((address_space(3))) double q8_lds;
typedef volatile __attribute__((address_space(3))) int q8_word;

// rows kept in LDS for a site of n rows: every wave takes the same EVEN number of 16-row tiles (rounds are tile pairs)
__host__ __device__ inline int q8_tiles_per_wave(int n) { const int t = ((n + 15) / 16 + 3) / 4; return (t + 1) & ~1; }
__host__ __device__ inline int q8_rows(int n) { return 4 * q8_tiles_per_wave(n) * 16; }
// a chain's slot (doubles): [job: alpha, -, beta (DP)] [V = phi - mu (VN)] [4 x partial results: X'g (DP), sum g, log-lik] [Omega V (VN)]
template <int DP> struct Q8Slot {
    static constexpr int RES = DP + 2, VN = 2 * DP + 8, BOFF = 2, VOFF = RES, RREC = RES, RESO = RES + VN, OVOFF = RESO + 4 * RREC,
                         DOUBLES = OVOFF + VN;
};
__device__ inline double q8_mfma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ inline void q8_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The row role of wave `wr`.  v_mfma_f64_4x4x4f64 operand layout (scripts/probe/mfma_layout.hip): A[b][i][k] in lane
// 16 k + 4 b + i, B[b][k][j] in lane 16 k + 4 b + j, D[b][i][j] in lane 16 i + 4 b + j; with lane = (hi, bb, lo):
//   forward   A = X[row 4 bb + lo of the tile][column 8 r + 2 hi (+1)],  B = beta of chain lo,  D = f[row 4 bb + hi][chain lo]
//   backward  B = g[row 4 bb + hi][chain lo] (the forward product's own lanes),  A = X[row 4 bb + hi][column 8 r + 2 lo (+1)],
//             D = (X' g)[column 8 r + 2 hi (+1)][chain lo], one partial sum per row block bb.
template <int DP>
struct QuadRows {
    using S = Q8Slot<DP>;
    static constexpr int SPR = DP / 2, RPL = DP >= 32 ? 1 : 32 / DP;
    static constexpr int KS = DP / 4, NRD = DP / 8, ROWB = DP * 8, TILEB = 16 * ROWB, TILEV = TILEB / 16;
    static constexpr int DMAX = 2 * DP + 2, NJ = (DMAX + 3) / 4, NGF = DMAX / 16, NJT = (NJ + 3) / 4;
    static_assert(NGF <= 4, "one 16-row group of the cavity term per wave");
    double om[NJ], omt[NJT];        // cavity precision as A operands: group wr (rows 16 wr + 4 bb + lo, columns 4 J + hi); wave 3 also the rows beyond the groups
    unsigned af[NRD], ab[NRD];      // LDS byte addresses of the wave's first tile pair: forward / backward operand
    unsigned ybits;                 // responses of this lane's product rows, one bit per tile
    int n, t0, t1, wr, lo, bb, hi, rb;
    bool g_on, t_on;
    q8_lds *sl;                     // the slot of chain lo: this lane's column of the products

    __device__ __forceinline__ void init(unsigned xbase, int n_, const uint8_t *y, const double *Om_g, int d, int wr_, int lane, q8_lds *slot0) {
        n = n_; wr = wr_;
        lo = lane & 3; bb = (lane >> 2) & 3; hi = lane >> 4;
        const int tpw = q8_tiles_per_wave(n);
        t0 = wr * tpw; t1 = t0 + tpw;
        {
            const int e = 16 * wr + 4 * bb + lo;
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
                const int c = 4 * J + hi;
                om[J] = (wr < NGF && e < d && c < d) ? Om_g[(size_t)c * d + e] : 0.0;
            }
            const int et = 16 * NGF + lo;
#pragma unroll
            for (int tt = 0; tt < NJT; ++tt) {
                const int c = 4 * (4 * tt + bb) + hi;
                omt[tt] = (wr == 3 && et < d && c < d) ? Om_g[(size_t)c * d + et] : 0.0;
            }
        }
        g_on = wr < NGF && 16 * wr < d; t_on = wr == 3 && d > 16 * NGF;
        const int rf = lane & 15;
        rb = 4 * bb + hi;
        ybits = 0;
        for (int t = t0; t < t1; ++t) {
            const int r = 16 * t + rb;
            if (r < n && y[r]) ybits |= 1u << (t - t0);
        }
        const unsigned swf = (unsigned)((rf / RPL) & (SPR - 1)), swb = (unsigned)((rb / RPL) & (SPR - 1));
#pragma unroll
        for (int r = 0; r < NRD; ++r) {
            af[r] = xbase + (unsigned)t0 * TILEB + (unsigned)rf * ROWB + ((((unsigned)(4 * r + hi)) ^ swf) << 4);
            ab[r] = xbase + (unsigned)t0 * TILEB + (unsigned)rb * ROWB + ((((unsigned)(4 * r + lo)) ^ swb) << 4);
        }
        sl = slot0 + lo * S::DOUBLES;
    }

    // One pass: the jobs of the four chains are in the slots; on return this wave's partial sums (X'g, sum g, log-lik) and
    // its rows of Omega V are stored (not yet waited for: the caller's barrier drains the LDS).
    __device__ __forceinline__ void pass() {
        // ---- operands (all requested before the first product)
        double bop[KS];
#pragma unroll
        for (int r = 0; r < NRD; ++r) {
            const q8_v2 v = *(q8_v2p)(sl + S::BOFF + 8 * r + 2 * hi);
            bop[2 * r] = v.x; bop[2 * r + 1] = v.y;
        }
        const double alpha_c = sl[0];
        // ---- cavity term Omega V of the four chains
        if (g_on) {
            double vb[NJ];
#pragma unroll
            for (int J = 0; J < NJ; ++J) vb[J] = sl[S::VOFF + 4 * J + hi];
            double acc = 0.0, acc1 = 0.0;
#pragma unroll
            for (int J = 0; J < NJ; J += 2) {
                acc = q8_mfma(om[J], vb[J], acc);
                if (J + 1 < NJ) acc1 = q8_mfma(om[J + 1 < NJ ? J + 1 : J], vb[J + 1 < NJ ? J + 1 : J], acc1);
            }
            sl[S::OVOFF + 16 * wr + rb] = acc + acc1;
        }
        if (t_on) {
            double vt[NJT];
#pragma unroll
            for (int tt = 0; tt < NJT; ++tt) vt[tt] = sl[S::VOFF + 4 * (4 * tt + bb) + hi];
            double acc = 0.0;
#pragma unroll
            for (int tt = 0; tt < NJT; ++tt) acc = q8_mfma(omt[tt], vt[tt], acc);
            acc += dpp_d<0x124>(acc); acc += dpp_d<0x128>(acc);          // the four blocks' k-shares (row_ror 4, 8)
            if (bb == 0) sl[S::OVOFF + 16 * NGF + hi] = acc;
        }
        // ---- the rows: two tiles per round (their logistic terms overlap); the LDS reads run ahead of their use
        double gacc[KS];
#pragma unroll
        for (int c = 0; c < KS; ++c) gacc[c] = 0.0;
        double dsum = 0.0, lsum = 0.0, wprod = 1.0;
        q8_v2 xf0[NRD], xf1[NRD];
#pragma unroll
        for (int r = 0; r < NRD; ++r) {
            xf0[r] = *reinterpret_cast<const q8_v2p>((uintptr_t)af[r]);
            xf1[r] = *reinterpret_cast<const q8_v2p>((uintptr_t)(af[r] + TILEB));
        }
        q8_v2p pf[NRD], pb[NRD];
#pragma unroll
        for (int r = 0; r < NRD; ++r) { pf[r] = reinterpret_cast<q8_v2p>((uintptr_t)af[r]); pb[r] = reinterpret_cast<q8_v2p>((uintptr_t)ab[r]); }
        unsigned yb = ybits;
        for (int t = t0; t < t1; t += 2, yb >>= 2) {
            const double y0 = (double)(yb & 1u), y1 = (double)((yb >> 1) & 1u);
            double f0 = alpha_c, f1 = alpha_c;
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                // (the two tiles' chains alternate: a product that waits for its own accumulator issues 4 cycles late)
                f0 = q8_mfma(xf0[r].x, bop[2 * r], f0); f1 = q8_mfma(xf1[r].x, bop[2 * r], f1);
                __builtin_amdgcn_sched_barrier(0);
                f0 = q8_mfma(xf0[r].y, bop[2 * r + 1], f0); f1 = q8_mfma(xf1[r].y, bop[2 * r + 1], f1);
                __builtin_amdgcn_sched_barrier(0);
            }
            q8_v2 xb0[NRD], xb1[NRD];
#pragma unroll
            for (int r = 0; r < NRD; ++r) { xb0[r] = pb[r][0]; xb1[r] = pb[r][TILEV]; }
            // (the round after the last one reads what lies behind the wave's tiles: values unused)
#pragma unroll
            for (int r = 0; r < NRD; ++r) { xf0[r] = pf[r][2 * TILEV]; xf1[r] = pf[r][3 * TILEV]; }
            double l0, l1, w0, w1, g0, g1;
            logistic_pair_lean(f0, f1, y0, y1, l0, l1, w0, w1, g0, g1);
            if (16 * (t + 2) > n) {                // the site's last tile(s): rows beyond n add nothing
                const bool v0 = 16 * t + rb < n, v1 = 16 * (t + 1) + rb < n;
                l0 = v0 ? l0 : 0.0; w0 = v0 ? w0 : 1.0; g0 = v0 ? g0 : 0.0;
                l1 = v1 ? l1 : 0.0; w1 = v1 ? w1 : 1.0; g1 = v1 ? g1 : 0.0;
            }
            lsum += l0; wprod *= w0; dsum += g0;
            lsum += l1; wprod *= w1; dsum += g1;
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                gacc[2 * r] = q8_mfma(xb0[r].x, g0, gacc[2 * r]); gacc[2 * r + 1] = q8_mfma(xb0[r].y, g0, gacc[2 * r + 1]);
            }
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                gacc[2 * r] = q8_mfma(xb1[r].x, g1, gacc[2 * r]); gacc[2 * r + 1] = q8_mfma(xb1[r].y, g1, gacc[2 * r + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < NRD; ++r) { pf[r] += 2 * TILEV; pb[r] += 2 * TILEV; }
        }
        // ---- sums over the row blocks (lanes ^ 4, ^ 8), then over the rows hi of a block for the two scalars
        // (a product with ones: D[.][j] = sum over k of B[k][j])
#pragma unroll
        for (int c = 0; c < KS; ++c) { gacc[c] += dpp_d<0x124>(gacc[c]); gacc[c] += dpp_d<0x128>(gacc[c]); }
        q8_lds *res = sl + S::RESO + wr * S::RREC;
        double dz = q8_mfma(1.0, dsum, 0.0), lz = q8_mfma(1.0, lsum - log_ge1_d_vc(wprod), 0.0);
        dz += dpp_d<0x124>(dz); lz += dpp_d<0x124>(lz);
        dz += dpp_d<0x128>(dz); lz += dpp_d<0x128>(lz);
        if (bb == 0) {
#pragma unroll
            for (int c = 0; c < KS; ++c) res[8 * (c >> 1) + 2 * hi + (c & 1)] = gacc[c];
            if (hi == 0) { res[DP] = dz; res[DP + 1] = lz; }
        }
    }
};

}  // namespace epx
