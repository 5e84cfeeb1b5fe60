// Accuracy of v_rcp_f64 and of one / two Newton steps behind it, on w = 1 + e, e in (0, 1] (the logistic terms' argument):
// hipcc --offload-arch=gfx950 -O2 scripts/probe/rcp_accuracy.hip -o variants/rcp_accuracy && variants/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *w, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = w[i];
    double q = __builtin_amdgcn_rcp(x);
    r0[i] = q;
    double t = fma(-x, q, 1.0); q = fma(q, t, q); r1[i] = q;
    t = fma(-x, q, 1.0); q = fma(q, t, q); r2[i] = q;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> w(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w[i] = 1.0 + (double)(s >> 11) * (1.0 / 9007199254740992.0); }
    for (int i = 0; i < 4096; ++i) w[i] = ldexp(w[i], (i % 200) - 100);          // (and other binades)
    double *dw, *d0, *d1, *d2;
    hipMalloc(&dw, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dw, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0; long ne1 = 0, ne2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double ex = 1.0L / (long double)w[i];
        const double ref = (double)ex;
        e0 = fmax(e0, fabs((double)(((long double)a[i] - ex) / ex)));
        e1 = fmax(e1, fabs((double)(((long double)b[i] - ex) / ex)));
        e2 = fmax(e2, fabs((double)(((long double)c[i] - ex) / ex)));
        ne1 += b[i] != ref; ne2 += c[i] != ref;
    }
    printf("v_rcp_f64: max relative error %.3g (2^%.1f); + one Newton step %.3g (%ld of %d not correctly rounded); + two %.3g (%ld not correctly rounded)\n",
           e0, log2(e0), e1, ne1, n, e2, ne2);
    return 0;
}
