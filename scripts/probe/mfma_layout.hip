// Finds the operand layout of v_mfma_f64_4x4x4f64 empirically (one-hot inputs).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    const int lane = threadIdx.x;
    for (int a = 0; a < 64; ++a)
        for (int b = 0; b < 64; ++b) {
            const double A = lane == a ? 1.0 : 0.0, B = lane == b ? 1.0 : 0.0;
            const double D = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, 0.0, 0, 0, 0);
            if (D != 0.0) out[a * 64 + b] = lane;
        }
}
int main() {
    int *d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0xff, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    static int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int a = 0; a < 64; ++a) {
        printf("A lane %2d:", a);
        for (int b = 0; b < 64; ++b) if (h[a * 64 + b] >= 0) printf(" B%d->D%d", b, h[a * 64 + b]);
        printf("\n");
    }
    return 0;
}
