// Does the dispatcher keep every CU busy when the workgroups of ONE grid last unequally long?  The pieced launches
// (epx_pieces.h) have one workgroup per piece, one workgroup per CU at a time (160 KB of LDS) and pieces of 0.03-1.5 s;
// the C5-shard timeline (profiles/r03_stream_piece_timeline.json) shows ~10 % of the CU-time lost BETWEEN workgroups.
// This probe has no sampler in it: a workgroup takes its duration from a table, sleeps on the 100 MHz clock and stamps
// entry, end and the CU it ran on.  Forms: (grid) one workgroup per table entry; (persistent) one workgroup per CU that
// takes entries from an atomic counter.
//   variants/dispatch_gaps [entries] [mean_ms]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline void spin_for(unsigned long long ticks, unsigned long long *rec) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) {
        rec[0] = t0; rec[1] = __builtin_amdgcn_s_memrealtime();
        rec[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID
        rec[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);     // XCC_ID
    }
}

__global__ void __launch_bounds__(512) k_grid(const unsigned *dur, unsigned long long *rec) {
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = 0;
    spin_for((unsigned long long)dur[blockIdx.x], rec + (size_t)blockIdx.x * 4);
}

__global__ void __launch_bounds__(512) k_persistent(const unsigned *dur, unsigned long long *rec, int n, int *next) {
    extern __shared__ unsigned char smem[];
    __shared__ int mine;
    smem[threadIdx.x] = 0;
    for (;;) {
        if (threadIdx.x == 0) mine = atomicAdd(next, 1);
        __syncthreads();
        const int i = __builtin_amdgcn_readfirstlane(mine);          // (uniform: the loop below must not diverge around the barriers)
        __syncthreads();
        if (i >= n) break;
        spin_for((unsigned long long)__builtin_amdgcn_readfirstlane((int)dur[i]), rec + (size_t)i * 4);
        __syncthreads();
    }
}

struct Res { double span_ms, ideal_ms, mean_busy, gap_med_us, gap_p90_us, gap_sum_ms_per_cu; int cus; int xcd_match; };

static Res analyse(const std::vector<unsigned long long> &r, int n, bool by_block) {
    unsigned long long t0 = ~0ull, t1 = 0; double busy = 0;
    for (int i = 0; i < n; ++i) { t0 = std::min(t0, r[4 * i]); t1 = std::max(t1, r[4 * i + 1]); busy += (double)(r[4 * i + 1] - r[4 * i]); }
    std::vector<std::vector<std::pair<unsigned long long, unsigned long long>>> cu(4096);
    int match = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = (unsigned)r[4 * i + 2], xcc = (unsigned)r[4 * i + 3] & 0xF;
        const int key = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF);
        cu[key].push_back({r[4 * i], r[4 * i + 1]});
        if (by_block && (int)xcc == i % 8) ++match;
    }
    std::vector<double> gaps; int ncu = 0;
    for (auto &v : cu) {
        if (v.empty()) continue;
        ++ncu; std::sort(v.begin(), v.end());
        for (size_t j = 1; j < v.size(); ++j) gaps.push_back((double)(v[j].first - v[j - 1].second) * 0.01);
    }
    std::sort(gaps.begin(), gaps.end());
    double gs = 0; for (double g : gaps) gs += g;
    Res o;
    o.span_ms = (double)(t1 - t0) * 1e-5; o.ideal_ms = busy * 1e-5 / ncu; o.mean_busy = busy / (double)(t1 - t0);
    o.gap_med_us = gaps.empty() ? 0 : gaps[gaps.size() / 2]; o.gap_p90_us = gaps.empty() ? 0 : gaps[gaps.size() * 9 / 10];
    o.gap_sum_ms_per_cu = gs * 1e-3 / ncu; o.cus = ncu; o.xcd_match = match;
    return o;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n = argc > 1 ? atoi(argv[1]) : 8704;
    const double mean_ms = argc > 2 ? atof(argv[2]) : 4.0;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const int lds = 160 * 1024 - 64;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_grid), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_persistent), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    unsigned *dur_d; unsigned long long *rec_d; int *next_d;
    CK(hipMalloc(&dur_d, n * sizeof(unsigned))); CK(hipMalloc(&rec_d, (size_t)n * 32)); CK(hipMalloc(&next_d, 4));
    std::vector<unsigned> dur(n); std::vector<unsigned long long> rec((size_t)n * 4);
    const char *names[] = {"equal", "uniform 0.25-1.75 x mean", "bimodal 0.3 / 1.7 x mean", "as the C5 pieces: longest first, 0.3-1.8 x mean"};
    printf("%d CUs, %d workgroups of 512 threads and %d B of LDS, mean duration %.2f ms\n", ncu, n, lds, mean_ms);
    for (int form = 0; form < 4; ++form) {
        srand(7);
        for (int i = 0; i < n; ++i) {
            const double u = rand() / (double)RAND_MAX;
            double f = 1.0;
            if (form == 1) f = 0.25 + 1.5 * u;
            if (form == 2) f = u < 0.5 ? 0.3 : 1.7;
            if (form == 3) f = 0.3 + 1.5 * (1.0 - (double)i / n) * (0.6 + 0.4 * u) / 0.8 * 0.8;
            dur[i] = (unsigned)(f * mean_ms * 1e5);
        }
        CK(hipMemcpy(dur_d, dur.data(), n * sizeof(unsigned), hipMemcpyHostToDevice));
        for (int pers = 0; pers < 2; ++pers) {
            CK(hipMemset(rec_d, 0, (size_t)n * 32)); CK(hipMemset(next_d, 0, 4));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            if (pers) hipLaunchKernelGGL(k_persistent, dim3(ncu), dim3(512), lds, 0, dur_d, rec_d, n, next_d);
            else hipLaunchKernelGGL(k_grid, dim3(n), dim3(512), lds, 0, dur_d, rec_d);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(rec.data(), rec_d, (size_t)n * 32, hipMemcpyDeviceToHost));
            Res o = analyse(rec, n, !pers);
            printf("%-52s %-10s launch %8.1f ms  ideal %8.1f ms  (%.3f)  CUs busy on average %6.1f of %d  gap on a CU: median %8.1f us  p90 %8.1f us  sum %7.2f ms per CU",
                   names[form], pers ? "persistent" : "grid", ms, o.ideal_ms, o.ideal_ms / ms, o.mean_busy, o.cus, o.gap_med_us, o.gap_p90_us, o.gap_sum_ms_per_cu);
            if (!pers) printf("  XCC_ID == blockIdx %% 8 for %d of %d", o.xcd_match, n);
            printf("\n");
        }
    }
    return 0;
}
