// What v_mfma_f64_4x4x4f64 costs on gfx950: cycles per instruction back to back (independent and dependent
// accumulators), and what a second wave on the same SIMD pays for its FP64 vector instructions meanwhile.
// One workgroup of 512 threads: wave w and w + 4 share SIMD w % 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 4096

template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned long long *out, double *sink, double seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double a = seed + lane, b = 1.0 + 1e-9 * lane;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    double v0 = a, v1 = b, v2 = a + 1, v3 = b + 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool do_mfma = (MODE == 0 || MODE == 1 || MODE == 3 || MODE == 4) && wave < 4;
    const bool do_valu = (MODE == 2 && wave < 4) || ((MODE == 3 || MODE == 4) && wave >= 4);
    if (do_mfma) {
        if (MODE == 1 || MODE == 4) {
            for (int i = 0; i < N; ++i) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);       // one dependent chain
        } else {
            for (int i = 0; i < N / 4; ++i) {
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
            }
        }
    }
    if (do_valu) {
        for (int i = 0; i < N / 4; ++i) {
            v0 = fma(v0, b, a); v1 = fma(v1, b, a); v2 = fma(v2, b, a); v3 = fma(v3, b, a);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = c0 + c1 + c2 + c3 + v0 + v1 + v2 + v3;
}

template <int MODE> void run(const char *what) {
    unsigned long long *d; double *s;
    hipMalloc(&d, 64); hipMalloc(&s, 512 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, d, s, 1.0);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("%-72s cycles per instruction: waves 0-3 %.2f %.2f %.2f %.2f | waves 4-7 %.2f %.2f %.2f %.2f\n", what,
           h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)N, h[4] / (double)N, h[5] / (double)N, h[6] / (double)N, h[7] / (double)N);
    hipFree(d); hipFree(s);
}

int main() {
    run<0>("mfma_f64_4x4x4, 4 independent accumulators, one wave per SIMD");
    run<1>("mfma_f64_4x4x4, one dependent chain, one wave per SIMD");
    run<2>("v_fma_f64, 4 independent chains, one wave per SIMD");
    run<3>("waves 0-3 mfma (independent) beside waves 4-7 v_fma_f64 on the same SIMDs");
    run<4>("waves 0-3 mfma (dependent chain) beside waves 4-7 v_fma_f64");
    return 0;
}
