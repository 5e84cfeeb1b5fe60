// Pieces shared by the LDS-resident sampler kernels (nuts.hip, nuts_duo.hip): the register layout
// of a length-P vector (element e in lane e % 64, register e / 64) and the transposing butterfly.
#pragma once
#include "epx_device.h"
#include "epx_kernels.h"

namespace epx {

template <int NV> struct Vec { double v[NV]; };

#define FORV _Pragma("unroll") for (int i = 0; i < NV; ++i)

template <int NV>
__device__ inline double gatherV(const Vec<NV> &x, int e) {
    double r = 0.0;
    FORV {
        const double t = __shfl(x.v[i], e & 63, 64);
        if ((e >> 6) == i) r = t;
    }
    return r;
}
// element e (wave-uniform index) as a scalar
template <int NV>
__device__ inline double elemU(const Vec<NV> &x, int e) {
    double r = 0.0;
    FORV { if ((e >> 6) == i) r = readlane_d(x.v[i], e & 63); }
    return r;
}

// A length-P vector kept in global memory behind the same `.v[i]` syntax (element e of lane e % 64 at
// b[e]): for the vectors of the tree bookkeeping that change once per subtree or per transition.  `b`
// is wave-uniform, so every access is `saddr + lane * 8 + immediate`.
struct GRef {
    gdouble *p; bool ok;       // elements beyond the vector's length are 0 and not stored (fewer cache lines per vector)
    __device__ operator double() const { return ok ? *p : 0.0; }
    __device__ const GRef &operator=(double x) const { if (ok) *p = x; return *this; }
    __device__ const GRef &operator=(const GRef &o) const { const double x = o; if (ok) *p = x; return *this; }
    __device__ const GRef &operator+=(double x) const { if (ok) *p = *p + x; return *this; }
};
struct GIdx {
    gdouble *b; int lane, len;
    __device__ GRef operator[](int i) const { return GRef{b + (lane + 64 * i), lane + 64 * i < len}; }
};
struct GVec { GIdx v; };
// A wave-uniform scalar of the bookkeeping kept in the chain's global store: the adaptation state and the run's
// statistics are touched once per transition -- as registers they would be live through every leapfrog
struct GScal {
    gdouble *p;
    __device__ operator double() const { return *p; }
    __device__ const GScal &operator=(double x) const { *p = x; return *this; }
    __device__ const GScal &operator=(const GScal &o) const { const double x = *o.p; *p = x; return *this; }
    __device__ const GScal &operator+=(double x) const { *p = *p + x; return *this; }
};
struct RScal {            // the same interface on a register
    double x;
    __device__ operator double() const { return x; }
    __device__ RScal &operator=(double v) { x = v; return *this; }
    __device__ RScal &operator+=(double v) { x += v; return *this; }
};
enum { GV_QS, GV_GS, GV_PQ, GV_PP, GV_PG, GV_MQ, GV_MP, GV_MG, GV_RHO, GV_PSP, GV_PSM, GV_WMEAN, GV_WM2, GV_BQ, GV_BG,
       GV_SCAL,                 // one vector's worth of scalars (GScal)
       GV_COUNT };

__device__ inline gdouble *uniform_ptr(double *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<gdouble *>((uintptr_t)(((unsigned long long)hi32 << 32) | lo32));
}

template <int DP> struct Log2 { static constexpr int v = 1 + Log2<DP / 2>::v; };
template <> struct Log2<1> { static constexpr int v = 0; };

enum { MODE_INIT = 0, MODE_SS = 1, MODE_TREE = 2 };

// Transposing reduction of CNT per-lane partial sums over the 64 lanes of a wave:
// each stage halves the values a lane carries and doubles the lanes summed, so
// CNT values cost CNT-1 exchanges (not 6*CNT).  Stage with selector bit B keeps
// the half of the values chosen by the lane's own bit B and receives the same
// half from a partner lane whose bit B differs.  Fully static indexing.
template <int CNT, int B>
__device__ inline void butterfly(double *acc, int lane) {
    if constexpr (CNT > 1) {
        if constexpr (B >= 4) {
#pragma unroll
            for (int j = 0; j < CNT / 2; ++j) acc[j] = swap_add_d<B>(acc[j], acc[j + CNT / 2]);
        } else {
            const bool upper = (lane >> B) & 1;
#pragma unroll
            for (int j = 0; j < CNT / 2; ++j) {
                const double send = upper ? acc[j] : acc[j + CNT / 2];
                const double keep = upper ? acc[j + CNT / 2] : acc[j];
                acc[j] = keep + partner_d<B>(send, lane);
            }
        }
        butterfly<CNT / 2, B - 1>(acc, lane);
    } else if constexpr (B >= 0) {
        acc[0] += partner_d<B>(acc[0], lane);
        butterfly<1, B - 1>(acc, lane);
    }
}

}  // namespace epx
