// Stand-alone probe of the row streaming engine (epx_stream_tile.h): validates one pass against
// a host computation and times back-to-back passes.  Build: see scripts/probe/build.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "epx_stream_tile.h"
using namespace epx;

template <int DPB>
__global__ void __launch_bounds__(STREAM_THREADS)
k_probe(const double *X, const int *y32, const long long *k_lim, int D, int ntmax, const double *beta, int ticks,
        double *G_out, double *dl_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x;
    PassArgs<DPB> s;
    s.Xg = X + (size_t)k_lim[k] * D; s.yg = y32 + k_lim[k];
    s.n = (int)(k_lim[k + 1] - k_lim[k]); s.D = D; s.ntile = (s.n + TR - 1) / TR;
    s.ngmax = 1; s.ntmax = ntmax;
    s.lds0 = (unsigned)(size_t)smem; s.slot_f = 0; s.slot_i = 0; s.t_i = 0; s.wave = wave; s.lane = lane;
    StreamLds L;
    L.template carve<DPB>(smem, 1, ntmax);
    if (tid == 0) { const long long gl[2] = {0, s.n}; build_tiles(1, gl, L.tdesc); }
    loader_init<DPB>(s, lane);
    for (int i = tid; i < DPB * NCH; i += STREAM_THREADS) L.beta_s[i] = (i / NCH) < D ? beta[(size_t)k * DPB * NCH + i] : 0.0;
    if (tid < 4) L.alpha_s[tid] = 0.1 * (tid + 1);
    __syncthreads();
    if (wave == NCH) ring_prime<DPB>(s, lane);
    double da = 0, ll = 0;
    for (int t = 0; t < ticks; ++t) {
        const PassOut o = stream_pass<DPB>(s);
        s.slot_f = o.slot_f; s.slot_i = o.slot_i; s.t_i = o.t_i; ll = o.ll;
    }
    if (wave < NCH) da = L.da_s[wave];
    if (wave == NCH) wait_vm<0>();
    __syncthreads();
    for (int i = tid; i < DPB * NCH; i += STREAM_THREADS) G_out[(size_t)k * DPB * NCH + i] = L.Gs[i];
    if (lane == 0 && wave < NCH) { dl_out[(k * 4 + wave) * 2] = da; dl_out[(k * 4 + wave) * 2 + 1] = ll; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DPB>
int run(int S, int n, int D, int ticks) {
    const size_t N = (size_t)S * n;
    std::vector<double> X(N * D), beta((size_t)S * DPB * 4, 0.0);
    std::vector<int> y(N);
    std::vector<long long> kl(S + 1);
    srand(1);
    for (auto &v : X) v = (rand() / (double)RAND_MAX - 0.5);
    for (auto &v : y) v = rand() & 1;
    for (int k = 0; k <= S; ++k) kl[k] = (long long)k * n;
    for (int k = 0; k < S; ++k) for (int j = 0; j < D; ++j) for (int c = 0; c < 4; ++c)
        beta[((size_t)k * DPB + j) * 4 + c] = 0.3 * (rand() / (double)RAND_MAX - 0.5);
    double *dX, *dB, *dG, *dL; int *dy; long long *dk;
    CK(hipMalloc(&dX, X.size() * 8 + 1024)); CK(hipMemset(dX, 0, X.size() * 8 + 1024));
    CK(hipMalloc(&dB, beta.size() * 8)); CK(hipMalloc(&dG, beta.size() * 8)); CK(hipMalloc(&dL, S * 8 * 8));
    CK(hipMalloc(&dy, N * 4)); CK(hipMalloc(&dk, (S + 1) * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, beta.data(), beta.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, y.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, kl.data(), (S + 1) * 8, hipMemcpyHostToDevice));
    const int ntmax = (n + TR - 1) / TR;
    const size_t lds = stream_map<DPB>(1, ntmax).end;
    auto kern = k_probe<DPB>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // correctness: one pass
    hipLaunchKernelGGL(kern, dim3(S), dim3(STREAM_THREADS), lds, 0, dX, dy, dk, D, ntmax, dB, 1, dG, dL);
    CK(hipDeviceSynchronize());
    std::vector<double> G(beta.size()), dl(S * 8);
    CK(hipMemcpy(G.data(), dG, G.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dl.data(), dL, dl.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int k : {0, S / 2, S - 1}) {
        for (int c = 0; c < 4; ++c) {
            std::vector<double> Gr(D, 0.0); double da = 0, ll = 0;
            for (int r = 0; r < n; ++r) {
                const double *xr = &X[((size_t)k * n + r) * D];
                double f = 0.1 * (c + 1);
                for (int j = 0; j < D; ++j) f += xr[j] * beta[((size_t)k * DPB + j) * 4 + c];
                const double yy = y[(size_t)k * n + r];
                const double p = 1.0 / (1.0 + exp(-f));
                ll += yy * f - log1p(exp(f)); const double g = yy - p; da += g;
                for (int j = 0; j < D; ++j) Gr[j] += xr[j] * g;
            }
            for (int j = 0; j < D; ++j) worst = fmax(worst, fabs(Gr[j] - G[((size_t)k * DPB + j) * 4 + c]));
            worst = fmax(worst, fabs(da - dl[(k * 4 + c) * 2]));
            worst = fmax(worst, fabs(ll - dl[(k * 4 + c) * 2 + 1]) / fabs(ll));
        }
    }
    printf("DPB=%d S=%d n=%d D=%d: max abs error vs host %.3e\n", DPB, S, n, D, worst);
    // timing
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(S), dim3(STREAM_THREADS), lds, 0, dX, dy, dk, D, ntmax, dB, ticks, dG, dL);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)S * ticks * ((double)n * D * 8 + n * 4);
        printf("  %d passes: %.2f ms, %.1f us/pass, %.2f TB/s (%.1f GB/s per block)\n", ticks, ms, ms * 1e3 / ticks,
               bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / (S < 256 ? S : 256));
    }
    hipFree(dX); hipFree(dB); hipFree(dG); hipFree(dL); hipFree(dy); hipFree(dk);
    return worst < 1e-9 ? 0 : 1;
}

int main(int argc, char **argv) {
    int bad = 0;
    bad |= run<128>(256, 2000, 128, 40);
    bad |= run<128>(512, 2000, 128, 40);
    bad |= run<128>(64, 2000, 128, 40);
    bad |= run<128>(256, 1999, 101, 40);       // odd D: 8-byte aligned rows, ragged last tile
    bad |= run<64>(256, 3000, 64, 40);
    bad |= run<64>(256, 777, 33, 40);
    printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad;
}
