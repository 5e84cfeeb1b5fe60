// Device-side helpers shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EPX_WAVE 64

// A pointer the compiler KNOWS to be global memory: through a generic pointer whose origin it cannot see (a base
// rebuilt from scalar registers, a member of a struct) it emits flat_load / flat_store, which are counted in vmcnt AND
// lgkmcnt and complete out of order -- every wait for an LDS read then also waits for the global stores in flight, and
// a wait for a flat load is a vmcnt(0) that drains an LDS-DMA ring.
typedef __attribute__((address_space(1))) double gdouble;
__device__ inline gdouble *as_global(const double *p) { return reinterpret_cast<gdouble *>((uintptr_t)p); }

namespace epx {

// ----------------------------------------------------------------------------
// Philox4x32-10 counter-based stream; identical to oracle/nuts_oracle.c so the
// device sampler and the CPU restatement take the same decisions.
struct RngKey { uint32_t k0, k1; };

__device__ __host__ inline RngKey make_key(uint64_t seed, int chain) {
    RngKey k;
    k.k0 = (uint32_t)seed;
    k.k1 = (uint32_t)(seed >> 32) ^ (0x85EBCA6Bu * (uint32_t)(chain + 1));
    return k;
}

__device__ inline void philox4x32(RngKey key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                  uint32_t out[4]) {
    uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform on the OPEN interval (0, 1): (v + 1/2) 2^-53 for a 53-bit integer v, so 2^-54 <= u <= 1 - 2^-54 (the Gumbel
// keys -log(-log u) and the Box-Muller radius of the samplers never see 0 or 1)
__device__ inline double u01(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ inline void rng_u2(RngKey k, uint32_t t, uint32_t kind, uint32_t a, uint32_t b,
                              double &u1, double &u2) {
    uint32_t o[4];
    philox4x32(k, t, kind, a, b, o);
    u1 = u01(o[0], o[1]);
    u2 = u01(o[2], o[3]);
}

__device__ inline double rng_uniform(RngKey k, uint32_t t, uint32_t kind, uint32_t a, uint32_t b) {
    double u1, u2;
    rng_u2(k, t, kind, a, b, u1, u2);
    return u1;
}

// standard normal of vector element e (Box-Muller pair shared by e and e^1)
__device__ inline double rng_normal(RngKey k, uint32_t t, uint32_t kind, int e, uint32_t b) {
    double u1, u2;
    rng_u2(k, t, kind, (uint32_t)(e >> 1), b, u1, u2);
    double r = sqrt(-2.0 * log(u1));
    double a = 6.283185307179586476925286766559 * u2;
    double s, c;
    sincos(a, &s, &c);
    return (e & 1) ? r * s : r * c;
}

// ----------------------------------------------------------------------------
// Cross-lane movement without LDS round trips.  DPP moves have VALU latency
// (the ds_bpermute path behind __shfl costs an LDS-crossbar round trip per
// dependent step, ~10x more on a serial reduction chain).
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK = 0xF>
__device__ inline double dpp_d(double x) {
    union { double d; int u[2]; } a, r;
    a.d = x;
    r.u[0] = __builtin_amdgcn_update_dpp(0, a.u[0], CTRL, ROW_MASK, 0xF, true);
    r.u[1] = __builtin_amdgcn_update_dpp(0, a.u[1], CTRL, ROW_MASK, 0xF, true);
    return r.d;
}
enum { DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_QUAD_REV = 0x1B, DPP_ROW_MIRROR = 0x140,
       DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143 };

// value of a partner lane whose lane-index bit B differs from this lane's:
//   B=5: lane^32 (v_permlane32_swap)   B=4: lane^16 (v_permlane16_swap)
//   B=3: lane^15 (row_mirror)          B=2: lane^7  (row_half_mirror)
//   B=1: lane^3  (quad reverse)        B=0: lane^1
template <int B>
__device__ inline double partner_d(double x, int lane) {
    if constexpr (B == 5 || B == 4) {
        union { double d; unsigned u[2]; } a, r;
        a.d = x;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v2u_t s;
            if constexpr (B == 5) s = __builtin_amdgcn_permlane32_swap(a.u[h], a.u[h], false, false);
            else s = __builtin_amdgcn_permlane16_swap(a.u[h], a.u[h], false, false);
            // s.x = this lane's old "vdst" view, s.y = "src0" view (see ISA: the halves / odd-even
            // rows are swapped between the two registers)
            r.u[h] = (lane & (1 << B)) ? s.x : s.y;
        }
        return r.d;
    } else if constexpr (B == 3) return dpp_d<DPP_ROW_MIRROR>(x);
    else if constexpr (B == 2) return dpp_d<DPP_ROW_HALF_MIRROR>(x);
    else if constexpr (B == 1) return dpp_d<DPP_QUAD_REV>(x);
    else return dpp_d<DPP_QUAD_XOR1>(x);
}

// One butterfly exchange for selector bit 5 or 4 without selects: v_permlane{32,16}_swap
// swaps the upper half (odd rows) of `a` with the lower half (even rows) of `b`, after which
// lanes whose bit B is 0 hold (own a, partner's a) and the others (partner's b, own b):
// a' + b' is "keep + received" for every lane.
template <int B>
__device__ inline double swap_add_d(double a, double b) {
    union { double d; unsigned u[2]; } x, y, p, q;
    x.d = a; y.d = b;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        v2u_t s;
        if constexpr (B == 5) s = __builtin_amdgcn_permlane32_swap(x.u[h], y.u[h], false, false);
        else s = __builtin_amdgcn_permlane16_swap(x.u[h], y.u[h], false, false);
        p.u[h] = s.x; q.u[h] = s.y;
    }
    return p.d + q.d;
}

// value of lane `lane` (wave-uniform index) broadcast as a scalar
__device__ inline double readlane_d(double v, int lane) {
    union { double d; uint32_t u[2]; } x;
    x.d = v;
    x.u[0] = __builtin_amdgcn_readlane(x.u[0], lane);
    x.u[1] = __builtin_amdgcn_readlane(x.u[1], lane);
    return x.d;
}

// value of lane addr4 / 4 in every lane through the LDS crossbar (all lanes active; addr4 = 4 x lane, uniform)
__device__ inline double bcast_lds(double v, int addr4) {
    union { double d; int u[2]; } x;
    x.d = v;
    x.u[0] = __builtin_amdgcn_ds_bpermute(addr4, x.u[0]);
    x.u[1] = __builtin_amdgcn_ds_bpermute(addr4, x.u[1]);
    return x.d;
}

// value of the lane whose index differs from this lane's ONLY in bit B (B >= 2):
// a true xor partner, needed by all-reduce stages that must not mix the low bits.
//   B=2: row_ror:4 and B=3: row_ror:8 rotate within a row of 16 and keep the low bits
//   (summing v + ror4(v), then + ror8, covers the 4 lanes that share (lane & 3)).
template <int B>
__device__ inline double plain_partner_d(double x, int lane) {
    if constexpr (B == 2) return dpp_d<0x124>(x);
    else if constexpr (B == 3) return dpp_d<0x128>(x);
    else return partner_d<B>(x, lane);
}

// wave64 sums: DPP row reduction, result broadcast from lane 63 through SGPRs.
__device__ inline double wave_sum(double v) {
    v += dpp_d<DPP_QUAD_XOR1>(v);
    v += dpp_d<DPP_QUAD_XOR2>(v);
    v += dpp_d<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_d<DPP_ROW_MIRROR>(v);
    v += dpp_d<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_d<DPP_ROW_BCAST31, 0xC>(v);
    return readlane_d(v, 63);
}
__device__ inline double wave_max(double v) {
    // old = -inf keeps rows that a row_bcast does not write neutral
    auto mx = [](double x, double y) { return fmax(x, y); };
    v = mx(v, dpp_d<DPP_QUAD_XOR1>(v));
    v = mx(v, dpp_d<DPP_QUAD_XOR2>(v));
    v = mx(v, dpp_d<DPP_ROW_HALF_MIRROR>(v));
    v = mx(v, dpp_d<DPP_ROW_MIRROR>(v));
    // every lane of a row now holds the row maximum: combine the 4 rows through SGPRs
    const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
    return mx(mx(r0, r1), mx(r2, r3));
}
__device__ inline void wave_sum2(double &a, double &b) {
    a += dpp_d<DPP_QUAD_XOR1>(a); b += dpp_d<DPP_QUAD_XOR1>(b);
    a += dpp_d<DPP_QUAD_XOR2>(a); b += dpp_d<DPP_QUAD_XOR2>(b);
    a += dpp_d<DPP_ROW_HALF_MIRROR>(a); b += dpp_d<DPP_ROW_HALF_MIRROR>(b);
    a += dpp_d<DPP_ROW_MIRROR>(a); b += dpp_d<DPP_ROW_MIRROR>(b);
    a += dpp_d<DPP_ROW_BCAST15, 0xA>(a); b += dpp_d<DPP_ROW_BCAST15, 0xA>(b);
    a += dpp_d<DPP_ROW_BCAST31, 0xC>(a); b += dpp_d<DPP_ROW_BCAST31, 0xC>(b);
    a = readlane_d(a, 63); b = readlane_d(b, 63);
}
// The same two sums in 22 instead of 40 vector instructions: one v_permlane32_swap exchange leaves a's partial sums in
// lanes 0..31 and b's in lanes 32..63 of ONE register, which the row stages then reduce together (a different order of
// additions than wave_sum2: only for kernels that do not promise bit-equality with it)
__device__ inline void wave_sum2_packed(double &a, double &b) {
    double v = swap_add_d<5>(a, b);
    v += dpp_d<DPP_QUAD_XOR1>(v);
    v += dpp_d<DPP_QUAD_XOR2>(v);
    v += dpp_d<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_d<DPP_ROW_MIRROR>(v);
    v += dpp_d<DPP_ROW_BCAST15, 0xA>(v);
    a = readlane_d(v, 31); b = readlane_d(v, 63);
}
__device__ inline void wave_sum3(double &a, double &b, double &c) {
    wave_sum2(a, b);
    c = wave_sum(c);
}

// make a value the compiler can keep in SGPRs (it is identical in all lanes)
__device__ inline double uniform_d(double v) {
    union { double d; uint32_t u[2]; } x;
    x.d = v;
    x.u[0] = __builtin_amdgcn_readfirstlane(x.u[0]);
    x.u[1] = __builtin_amdgcn_readfirstlane(x.u[1]);
    return x.d;
}
__device__ inline int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ inline double log_sum_exp2(double a, double b) {
    if (a == -INFINITY) return b;
    if (a == INFINITY && b == INFINITY) return INFINITY;
    if (a > b) return a + log1p(exp(b - a));
    return b + log1p(exp(a - b));
}

// log_sum_exp2 with the lean exponential / log1p of this file (the row team's bookkeeping: the other waves of the workgroup
// wait for the wave that completes a subtree)
__device__ inline double exp_d(double x);
__device__ inline double log1p_unit_d(double e);
__device__ inline double log_sum_exp2_lean(double a, double b) {
    if (a == -INFINITY) return b;
    if (a == INFINITY && b == INFINITY) return INFINITY;
    if (a > b) return a + log1p_unit_d(exp_d(b - a));
    return b + log1p_unit_d(exp_d(a - b));
}

// Merging two log-weights a (left) and b (right): lse = log(e^a + e^b) and the
// multinomial probability of the right one, e^b / (e^a + e^b), from ONE exp
// (base_nuts.hpp computes log_sum_exp and exp(b - lse) separately).
__device__ inline void merge_weights(double a, double b, double &lse, double &p_right) {
    if (a == -INFINITY) { lse = b; p_right = (b == -INFINITY) ? NAN : 1.0; return; }
    if (b >= a) {
        const double e = exp(a - b);
        lse = b + log1p(e);
        p_right = 1.0 / (1.0 + e);
    } else {
        const double e = exp(b - a);
        lse = a + log1p(e);
        p_right = e / (1.0 + e);
    }
}

// ----------------------------------------------------------------------------
// Lean double-precision elementary functions for the per-leapfrog hot path.
// The kernel is bound by instruction issue, and the library exp/log1p cost ~250
// instructions per logistic term (extended-precision internals); these are
// ~1e-16 relative (checked against the oracle's libm in the gradient tests) in ~60.
// v_rcp_f64 is good to 2^-24.4 (scripts/probe/rcp_accuracy.hip, profiles/r06_rcp_f64_accuracy.txt).  One THIRD-order
// step behind it -- t = 1 - x r, r (1 + t + t^2): truncation t^3 ~ 1e-22 -- takes three FMAs where two Newton steps took
// four, and rounds as well (round 6; the logistic pair is the row phase's instruction count).
__device__ inline double rcp_d(double x) {            // 1/x, x finite and normal
    const double r = __builtin_amdgcn_rcp(x);
    const double t = fma(-x, r, 1.0);
    return fma(r, fma(t, t, t), r);
}

// exp(r) on |r| <= ln2 / 2 (+ the slack of the two-constant range reduction): Chebyshev interpolant of degree 11 on
// [-0.3466, 0.3466] in the monomial basis, computed in 60-digit arithmetic (scripts/exp_minimax.py: approximation error
// 4.2e-18 relative; the double-precision Horner form errs by 2.2e-16, exactly as the degree-13 Taylor form did).
// c1 = c0 = 1 to the last bit.  ONE list: exp_d, exp_d_vc and the logistic pairs below evaluate the same polynomial in
// the same order, so they agree bit for bit with one another.
#define EPX_EXP_MAGIC 6755399441055744.0       /* 1.5 * 2^52 */
#define EPX_EXP_C11 2.5110046444457486e-08
#define EPX_EXP_C10 2.7632651132874099e-07
#define EPX_EXP_C9 2.7557240894691074e-06
#define EPX_EXP_C8 2.4801485451264055e-05
#define EPX_EXP_C7 0.00019841269890069431
#define EPX_EXP_C6 0.0013888888952343797
#define EPX_EXP_C5 0.0083333333333195925
#define EPX_EXP_C4 0.041666666666487988
#define EPX_EXP_C3 0.1666666666666668
#define EPX_EXP_C2 0.50000000000000189

__device__ inline double exp_d(double x) {
    const double xc = fmin(fmax(x, -800.0), 800.0);   // keeps k*ln2 finite; ldexp saturates to 0 / inf
    // k = round(x / ln2) by the magic-number form: x log2(e) + 1.5 * 2^52 (one FMA, rounded once) has k in its low mantissa
    // bits -- as an integer for ldexp WITHOUT a conversion instruction -- and kf = that minus the constant (round 6)
    const double kt = fma(xc, 1.4426950408889634074, EPX_EXP_MAGIC);
    const double kf = kt - EPX_EXP_MAGIC;
    double r = fma(kf, -6.93147180369123816490e-01, xc);
    r = fma(kf, -1.90821492927058770002e-10, r);
    // |r| <= 0.3466: the degree-11 near-minimax polynomial (EPX_EXP_C*, above: 4.2e-18, the accuracy of the Taylor form
    // to r^13 it replaced in round 6, two steps shorter)
    double p = EPX_EXP_C11;
    p = fma(p, r, EPX_EXP_C10);
    p = fma(p, r, EPX_EXP_C9);
    p = fma(p, r, EPX_EXP_C8);
    p = fma(p, r, EPX_EXP_C7);
    p = fma(p, r, EPX_EXP_C6);
    p = fma(p, r, EPX_EXP_C5);
    p = fma(p, r, EPX_EXP_C4);
    p = fma(p, r, EPX_EXP_C3);
    p = fma(p, r, EPX_EXP_C2);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double res = ldexp(p, __double2loint(kt));
    return (x != x) ? x : res;                        // NaN in, NaN out
}

// log(1 + e) for e in [0, 1]: 2 atanh(s), s = (m-1)/(m+1) after folding m = 1+e into [1/sqrt2, sqrt2]
__device__ inline double log1p_unit_d(double e) {
    const double m = 1.0 + e;
    const bool big = m > 1.4142135623730951;
    const double a = big ? fma(0.5, m, -1.0) : e;     // mm - 1 (exact for the folded branch)
    const double b = big ? fma(0.5, m, 1.0) : 2.0 + e;
    const double s = a * rcp_d(b);
    const double z = s * s;                           // |s| <= 0.1716
    double p = 4.7619047619047616e-02;                // 1/21
    p = fma(p, z, 5.2631578947368418e-02);            // 1/19
    p = fma(p, z, 5.8823529411764705e-02);            // 1/17
    p = fma(p, z, 6.6666666666666666e-02);            // 1/15
    p = fma(p, z, 7.6923076923076927e-02);            // 1/13
    p = fma(p, z, 9.0909090909090912e-02);            // 1/11
    p = fma(p, z, 1.1111111111111110e-01);            // 1/9
    p = fma(p, z, 1.4285714285714285e-01);            // 1/7
    p = fma(p, z, 0.2);                               // 1/5
    p = fma(p, z, 3.3333333333333331e-01);            // 1/3
    p = fma(p, z, 1.0);
    const double r = 2.0 * s * p;
    return big ? r + 6.931471805599453094e-01 : r;
}

// y f - log(1+e^f) and y - sigmoid(f) sharing one exp (bernoulli_logit)
__device__ inline void logistic_terms(double f, double y, double &ll, double &g) {
    const double e = exp_d(-fabs(f));
    const double l1p = log1p_unit_d(e);
    const double inv = rcp_d(1.0 + e);
    const double s = (f >= 0) ? inv : e * inv;
    ll = y * f - (fmax(f, 0.0) + l1p);
    g = y - s;
}

// log(x) for any positive, finite, NORMAL x: x = 2^k m with m in [1/sqrt2, sqrt2) (k of either sign), log m by the same
// atanh series as log1p_unit_d.  (Written for a product of (1 + e) factors, hence the name; the row team's state machine
// also takes it on uniforms in (0, 1) and on -log(u) in (0, inf): see log_pos_d.)  x == 0 does NOT give -inf here.
__device__ inline double log_ge1_d(double x) {
    int k = 0;
    double m = frexp(x, &k);                          // m in [0.5, 1)
    const bool lo = m < 0.70710678118654752;
    m = lo ? 2.0 * m : m; k = lo ? k - 1 : k;         // m in [1/sqrt2, sqrt2)
    const double s = (m - 1.0) * rcp_d(m + 1.0);
    const double z = s * s;                           // |s| <= 0.1716
    double p = 4.7619047619047616e-02;                // 1/21
    p = fma(p, z, 5.2631578947368418e-02);
    p = fma(p, z, 5.8823529411764705e-02);
    p = fma(p, z, 6.6666666666666666e-02);
    p = fma(p, z, 7.6923076923076927e-02);
    p = fma(p, z, 9.0909090909090912e-02);
    p = fma(p, z, 1.1111111111111110e-01);
    p = fma(p, z, 1.4285714285714285e-01);
    p = fma(p, z, 0.2);
    p = fma(p, z, 3.3333333333333331e-01);
    p = fma(p, z, 1.0);
    return fma((double)k, 6.931471805599453094e-01, 2.0 * s * p);
}

// log(x) of the bookkeeping's arguments -- a uniform of rng_u2 (strictly inside (0, 1): u01() adds one half to a 53-bit
// integer), minus its logarithm (positive), a sum of weights (0 only when every leaf of a subtree diverged: libm's -inf)
__device__ inline double log_pos_d(double x) { return x == 0.0 ? -INFINITY : log_ge1_d(x); }

// The same terms with the logarithm left to the caller: log-likelihood term = lin - log(w),
// w = 1 + exp(-|f|) in [1, 2].  A caller that sums many terms multiplies the w's and takes ONE
// logarithm per few hundred of them (the streaming sampler's logistic wave: the ~35 dependent
// instructions of log1p per tile were on the critical path of every tile phase).
__device__ inline void logistic_split(double f, double y, double &lin, double &w, double &g) {
    const double e = exp_d(-fabs(f));
    w = 1.0 + e;
    const double inv = rcp_d(w);
    const double s = (f >= 0) ? inv : e * inv;
    lin = y * f - fmax(f, 0.0);
    g = y - s;
}

// The same for TWO independent arguments, operation by operation in lock step: a single wave then has two
// dependent FP64 chains in flight instead of one (the values are those of logistic_split, bit for bit).
__device__ inline void logistic_split2(double fa, double fb, double ya, double yb, double &lina, double &linb,
                                       double &wa, double &wb, double &ga, double &gb) {
    const double xa = -fabs(fa), xb = -fabs(fb);
    const double ca = fmin(fmax(xa, -800.0), 800.0), cb = fmin(fmax(xb, -800.0), 800.0);
    const double kta = fma(ca, 1.4426950408889634074, EPX_EXP_MAGIC), ktb = fma(cb, 1.4426950408889634074, EPX_EXP_MAGIC);
    const double ka = kta - EPX_EXP_MAGIC, kb = ktb - EPX_EXP_MAGIC;          // (exp_d's magic-number rounding)
    double ra = fma(ka, -6.93147180369123816490e-01, ca), rb = fma(kb, -6.93147180369123816490e-01, cb);
    ra = fma(ka, -1.90821492927058770002e-10, ra); rb = fma(kb, -1.90821492927058770002e-10, rb);
    double pa = EPX_EXP_C11, pb = EPX_EXP_C11;
#define EPX_STEP2(c) pa = fma(pa, ra, c); pb = fma(pb, rb, c); __builtin_amdgcn_sched_barrier(0)
    EPX_STEP2(EPX_EXP_C10); EPX_STEP2(EPX_EXP_C9); EPX_STEP2(EPX_EXP_C8); EPX_STEP2(EPX_EXP_C7); EPX_STEP2(EPX_EXP_C6);
    EPX_STEP2(EPX_EXP_C5); EPX_STEP2(EPX_EXP_C4); EPX_STEP2(EPX_EXP_C3); EPX_STEP2(EPX_EXP_C2); EPX_STEP2(1.0); EPX_STEP2(1.0);
#undef EPX_STEP2
    double ea = ldexp(pa, __double2loint(kta)), eb = ldexp(pb, __double2loint(ktb));
    ea = (xa != xa) ? xa : ea; eb = (xb != xb) ? xb : eb;
    wa = 1.0 + ea; wb = 1.0 + eb;
    double qa = __builtin_amdgcn_rcp(wa), qb = __builtin_amdgcn_rcp(wb);
    const double ta = fma(-wa, qa, 1.0), tb = fma(-wb, qb, 1.0);          // (rcp_d's third-order step, two at a time)
    qa = fma(qa, fma(ta, ta, ta), qa); qb = fma(qb, fma(tb, tb, tb), qb);
    const double sa = (fa >= 0) ? qa : ea * qa, sb = (fb >= 0) ? qb : eb * qb;
    lina = ya * fa - fmax(fa, 0.0); linb = yb * fb - fmax(fb, 0.0);
    ga = ya - sa; gb = yb - sb;
}

// The same two logistic terms without the not-a-number select behind the exponential: a non-finite argument still
// shows in `lin` (y f - max(f, 0)), which is how the row team's caller learns of it -- through the log density.
__device__ inline void logistic_pair_lean(double fa, double fb, double nha, double nhb, double &lina, double &linb,
                                          double &wa, double &wb, double &ga, double &gb) {
    // -|f| clamped at -800 by ONE instruction (same value as fmin(fmax(-|f|, -800), 800): the compiler's form of it
    // canonicalises -|f| with a v_max of its own first and keeps the idle upper clamp: three instructions)
    // (fa, fb may come STRAIGHT from v_mfma_f64_4x4x4 -- the row team adds alpha through the accumulator -- and the
    // compiler's hazard recogniser does not look into inline assembly: a VALU read of a DGEMM 4x4x4 result needs 6 wait
    // states, which only the instructions that happen to be scheduled in between provided.  Under another machine
    // scheduler (-amdgpu-sched-strategy=iterative-minreg, round 5) they were not there and every layout-7 test failed; the
    // s_nop makes the distance part of the statement: 6 cycles per two tiles)
    double ca, cb;
    const double lim = -800.0;
    asm("s_nop 5\n\tv_max_f64 %0, -|%2|, %4\n\tv_max_f64 %1, -|%3|, %4" : "=&v"(ca), "=&v"(cb) : "v"(fa), "v"(fb), "s"(lim));
    const double kta = fma(ca, 1.4426950408889634074, EPX_EXP_MAGIC), ktb = fma(cb, 1.4426950408889634074, EPX_EXP_MAGIC);
    const double ka = kta - EPX_EXP_MAGIC, kb = ktb - EPX_EXP_MAGIC;          // (exp_d's magic-number rounding)
    double ra = fma(ka, -6.93147180369123816490e-01, ca), rb = fma(kb, -6.93147180369123816490e-01, cb);
    ra = fma(ka, -1.90821492927058770002e-10, ra); rb = fma(kb, -1.90821492927058770002e-10, rb);
    double pa = EPX_EXP_C11, pb = EPX_EXP_C11;
#define EPX_STEP2(c) pa = fma(pa, ra, c); pb = fma(pb, rb, c); __builtin_amdgcn_sched_barrier(0)
    EPX_STEP2(EPX_EXP_C10); EPX_STEP2(EPX_EXP_C9); EPX_STEP2(EPX_EXP_C8); EPX_STEP2(EPX_EXP_C7); EPX_STEP2(EPX_EXP_C6);
    EPX_STEP2(EPX_EXP_C5); EPX_STEP2(EPX_EXP_C4); EPX_STEP2(EPX_EXP_C3); EPX_STEP2(EPX_EXP_C2); EPX_STEP2(1.0); EPX_STEP2(1.0);
#undef EPX_STEP2
    const double ea = ldexp(pa, __double2loint(kta)), eb = ldexp(pb, __double2loint(ktb));
    wa = 1.0 + ea; wb = 1.0 + eb;
    double qa = __builtin_amdgcn_rcp(wa), qb = __builtin_amdgcn_rcp(wb);
    const double ta = fma(-wa, qa, 1.0), tb = fma(-wb, qb, 1.0);          // (rcp_d's third-order step, two at a time)
    qa = fma(qa, fma(ta, ta, ta), qa); qb = fma(qb, fma(tb, tb, tb), qb);
    // The responses arrive as nh = 1/2 - y (+1/2 for y = 0, -1/2 for y = 1: two bit operations from the row's bit, as
    // many as the conversion to 0.0 / 1.0 took).  sigmoid(f) - 1/2 is odd in f and q = sigmoid(|f|), so
    //     y - sigmoid(f) = -nh - copysign(q - 1/2, f)          (q - 1/2 is exact: q in [1/2, 1])
    // is a subtraction, a sign transfer and an addition where the selected form (f >= 0 ? q : e q, then y - s) took a
    // product, a comparison, two conditional moves and a subtraction; and  y f - max(f, 0) = -nh f - |f| / 2  exactly
    // (round 6).  The residual of a row with f << 0 now carries q's ABSOLUTE rounding error (1e-16) instead of a relative
    // one -- the size of the rounding of the sums it enters.
    const double ha = qa - 0.5, hb = qb - 0.5;
    lina = fma(-nha, fa, -0.5 * fabs(fa)); linb = fma(-nhb, fb, -0.5 * fabs(fb));
    ga = -nha - __builtin_copysign(ha, fa); gb = -nhb - __builtin_copysign(hb, fb);
    (void)ea; (void)eb;
}

// p * x + c as the three-address instruction, whatever register class the compiler found for the constant: where the
// scalar registers are used up (the row team's kernel) LLVM keeps polynomial coefficients in vector registers and then
// selects the two-address v_fmac_f64 -- which overwrites its addend, so every Horner step came with a v_mov_b64 of the
// coefficient in front of it.  Same value as fma().
__device__ inline double fma_vc(double p, double x, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(x), "v"(c));
    return r;
}
// exp_d, bit for bit, with those steps (the state wave's view update: the exponential sits on the critical stretch)
__device__ inline double exp_d_vc(double x) {
    const double xc = fmin(fmax(x, -800.0), 800.0);
    const double kt = fma(xc, 1.4426950408889634074, EPX_EXP_MAGIC);
    const double kf = kt - EPX_EXP_MAGIC;
    double r = fma(kf, -6.93147180369123816490e-01, xc);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = EPX_EXP_C11;
    p = fma_vc(p, r, EPX_EXP_C10); p = fma_vc(p, r, EPX_EXP_C9); p = fma_vc(p, r, EPX_EXP_C8); p = fma_vc(p, r, EPX_EXP_C7);
    p = fma_vc(p, r, EPX_EXP_C6); p = fma_vc(p, r, EPX_EXP_C5); p = fma_vc(p, r, EPX_EXP_C4); p = fma_vc(p, r, EPX_EXP_C3);
    p = fma_vc(p, r, EPX_EXP_C2);
    p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    const double res = ldexp(p, __double2loint(kt));
    return (x != x) ? x : res;
}
// log_ge1_d, bit for bit, with those steps (the row team's one logarithm per pass)
__device__ inline double log_ge1_d_vc(double x) {
    int k = 0;
    double m = frexp(x, &k);
    const bool lo = m < 0.70710678118654752;
    m = lo ? 2.0 * m : m; k = lo ? k - 1 : k;
    const double s = (m - 1.0) * rcp_d(m + 1.0);
    const double z = s * s;
    double p = 4.7619047619047616e-02;
    p = fma_vc(p, z, 5.2631578947368418e-02); p = fma_vc(p, z, 5.8823529411764705e-02); p = fma_vc(p, z, 6.6666666666666666e-02);
    p = fma_vc(p, z, 7.6923076923076927e-02); p = fma_vc(p, z, 9.0909090909090912e-02); p = fma_vc(p, z, 1.1111111111111110e-01);
    p = fma_vc(p, z, 1.4285714285714285e-01); p = fma_vc(p, z, 0.2); p = fma_vc(p, z, 3.3333333333333331e-01);
    p = fma(p, z, 1.0);
    return fma((double)k, 6.931471805599453094e-01, 2.0 * s * p);
}

}  // namespace epx
