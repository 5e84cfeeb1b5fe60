// The row team's pass of nuts_duo.hip (layout 7) alone on a CU: what a pass costs without the state waves, and which
// part of it (LDS reads, matrix products, logistic terms) sets the time.  One workgroup of 4 or 8 waves per CU; X of a
// C3 site (500 x 32, padded to 512 rows) in LDS with the kernel's swizzle.  MODE bits: 1 products, 2 logistic terms,
// 4 LDS reads (off: the operands are constants), 8 partner waves 4..7 run an FP64 vector stream beside the team.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "epx_device.h"
using namespace epx;

typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2f64 *lds_v2p;
__device__ inline double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

constexpr int DP = 32, NRD = DP / 8, KS = DP / 4, ROWB = DP * 8, TILEB = 16 * ROWB, NROW = 512;

// PAT: address pattern of the LDS reads. 0: the kernel's (forward: row = lane & 15, slot 4 r + hi; backward: row 4 bb + hi,
// slot 4 r + lo; slots XOR-swizzled by row); 1: linear (lane x 16 B: the conflict-free reference); 2: forward pattern for
// both; 3: backward pattern for both; 4: the kernel's pattern without the swizzle
template <int MODE, int PAT = 0>
__global__ void __launch_bounds__(512) k(unsigned long long *out, double *sink, int npass, int n) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *Xs = reinterpret_cast<double *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NROW * DP; i += blockDim.x) Xs[i] = 0.001 * ((i * 2654435761u) % 1000) - 0.5;
    __syncthreads();
    const int lo = lane & 3, bb = (lane >> 2) & 3, hi = lane >> 4;
    const int rf = lane & 15, rb = 4 * bb + hi;
    if (wave >= 4) {
        if (!(MODE & 8)) return;
        double v0 = lane, v1 = 1.0 + 1e-9 * lane, v2 = 2.0, v3 = 0.5;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < npass * 250; ++i) { v0 = fma(v0, v1, v2); v3 = fma(v3, v1, v2); v0 = fma(v0, v1, v3); v3 = fma(v3, v1, v0); }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0);
        sink[blockIdx.x * 512 + tid] = v0 + v3;
        return;
    }
    const int wr = wave;
    const int ntile = (n + 15) >> 4, tpw = (ntile + 3) >> 2;
    const int t0 = wr * tpw, t1 = ntile < t0 + tpw ? ntile : t0 + tpw;
    const unsigned xbase = (unsigned)(size_t)Xs;
    unsigned af[NRD], ab[NRD];
#pragma unroll
    for (int r = 0; r < NRD; ++r) {
        af[r] = xbase + (unsigned)rf * ROWB + ((((unsigned)(4 * r + hi)) ^ (unsigned)rf) << 4);
        ab[r] = xbase + (unsigned)rb * ROWB + ((((unsigned)(4 * r + lo)) ^ (unsigned)rb) << 4);
        if (PAT == 1) { af[r] = xbase + lane * 16 + r * 1024; ab[r] = af[r]; }
        if (PAT == 2) ab[r] = af[r];
        if (PAT == 3) af[r] = ab[r];
        if (PAT == 4) { af[r] = xbase + (unsigned)rf * ROWB + ((unsigned)(4 * r + hi) << 4); ab[r] = xbase + (unsigned)rb * ROWB + ((unsigned)(4 * r + lo) << 4); }
    }
    double bop[KS];
#pragma unroll
    for (int c = 0; c < KS; ++c) bop[c] = 0.01 * (c + lo);
    double total = 0.0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int pass = 0; pass < npass; ++pass) {
        double gacc[KS];
#pragma unroll
        for (int c = 0; c < KS; ++c) gacc[c] = 0.0;
        double dsum = 0.0, lsum = 0.0, wprod = 1.0;
        v2f64 xf0[NRD], xf1[NRD];
        if (MODE & 4) {
            const unsigned o0 = (unsigned)t0 * TILEB, o1 = o0 + TILEB;
#pragma unroll
            for (int r = 0; r < NRD; ++r) { xf0[r] = *(lds_v2p)(uintptr_t)(af[r] + o0); xf1[r] = *(lds_v2p)(uintptr_t)(af[r] + o1); }
        } else {
#pragma unroll
            for (int r = 0; r < NRD; ++r) { xf0[r].x = 0.1 * lane; xf0[r].y = 0.2; xf1[r].x = 0.3; xf1[r].y = 0.01 * lane; }
        }
        for (int t = t0; t < t1; t += 2) {
            const unsigned o0 = (unsigned)t * TILEB, o1 = o0 + TILEB;
            double f0 = 0.1 * pass, f1 = 0.2;
            if (MODE & 1) {
#pragma unroll
                for (int r = 0; r < NRD; ++r) {
                    f0 = mfma4(xf0[r].x, bop[2 * r], f0); f1 = mfma4(xf1[r].x, bop[2 * r], f1);
                    __builtin_amdgcn_sched_barrier(0);
                    f0 = mfma4(xf0[r].y, bop[2 * r + 1], f0); f1 = mfma4(xf1[r].y, bop[2 * r + 1], f1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < NRD; ++r) { f0 += xf0[r].x; f1 += xf1[r].y; }
            }
            v2f64 xb0[NRD], xb1[NRD];
            if (MODE & 4) {
#pragma unroll
                for (int r = 0; r < NRD; ++r) { xb0[r] = *(lds_v2p)(uintptr_t)(ab[r] + o0); xb1[r] = *(lds_v2p)(uintptr_t)(ab[r] + o1); }
                const int tn = t + 2 < t1 ? t + 2 : t;
                const unsigned n0 = (unsigned)tn * TILEB, n1 = n0 + TILEB;
#pragma unroll
                for (int r = 0; r < NRD; ++r) { xf0[r] = *(lds_v2p)(uintptr_t)(af[r] + n0); xf1[r] = *(lds_v2p)(uintptr_t)(af[r] + n1); }
            } else {
#pragma unroll
                for (int r = 0; r < NRD; ++r) { xb0[r] = xf0[r]; xb1[r] = xf1[r]; }
            }
            double l0 = f0, l1 = f1, w0 = 1.0, w1 = 1.0, g0 = f0, g1 = f1;
            if (MODE & 2) logistic_split2(f0, f1, 1.0, 0.0, l0, l1, w0, w1, g0, g1);
            lsum += l0; wprod *= w0; dsum += g0;
            lsum += l1; wprod *= w1; dsum += g1;
            if (MODE & 1) {
#pragma unroll
                for (int r = 0; r < NRD; ++r) { gacc[2 * r] = mfma4(xb0[r].x, g0, gacc[2 * r]); gacc[2 * r + 1] = mfma4(xb0[r].y, g0, gacc[2 * r + 1]); }
#pragma unroll
                for (int r = 0; r < NRD; ++r) { gacc[2 * r] = mfma4(xb1[r].x, g1, gacc[2 * r]); gacc[2 * r + 1] = mfma4(xb1[r].y, g1, gacc[2 * r + 1]); }
            } else {
#pragma unroll
                for (int r = 0; r < NRD; ++r) { gacc[2 * r] += xb0[r].x * g0; gacc[2 * r + 1] += xb1[r].y * g1; }
            }
        }
#pragma unroll
        for (int c = 0; c < KS; ++c) total += gacc[c];
        total += dsum + lsum + wprod;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = (c1 - c0);
    sink[blockIdx.x * 512 + tid] = total;
}

template <int MODE, int PAT = 0> void run(const char *what, int nblk) {
    unsigned long long *d; double *s;
    const int npass = 2000;
    (void)hipMalloc(&d, nblk * 64); (void)hipMalloc(&s, (size_t)nblk * 512 * 8);
    (void)hipMemset(d, 0, nblk * 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, NROW * ROWB);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, PAT>), dim3(nblk), dim3(512), NROW * ROWB, 0, d, s, npass, 500);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 8);
    (void)hipMemcpy(h.data(), d, nblk * 64, hipMemcpyDeviceToHost);
    printf("%-66s cycles per pass, waves 0-3: %6.0f %6.0f %6.0f %6.0f", what, h[0] / (double)npass, h[1] / (double)npass, h[2] / (double)npass, h[3] / (double)npass);
    if (MODE & 8) printf(" | partner stream, cycles per 1000 fma: %.0f", h[4] / (double)npass);
    printf("\n");
    (void)hipFree(d); (void)hipFree(s);
}

int main() {
    run<4, 0>("reads only, the kernel's pattern", 1);
    run<4, 1>("reads only, linear", 1);
    run<4, 2>("reads only, forward pattern twice", 1);
    run<4, 3>("reads only, backward pattern twice", 1);
    run<4, 4>("reads only, no swizzle", 1);
    run<7>("full pass (reads + products + logistic terms), 1 workgroup", 1);
    run<7>("full pass, 256 workgroups", 256);
    run<5>("reads + products", 1);
    run<6>("reads + logistic terms", 1);
    run<4>("reads only", 1);
    run<3>("products + logistic terms, operands in registers", 1);
    run<1>("products only", 1);
    run<2>("logistic terms only", 1);
    run<15>("full pass beside an FP64 vector stream on every SIMD", 1);
    run<8 + 2>("logistic terms only beside the FP64 vector stream", 1);
    run<8 + 1>("products only beside the FP64 vector stream", 1);
    return 0;
}
