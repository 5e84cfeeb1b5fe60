// Probe: do two kernels launched on two streams of one process share the GPU?  Each kernel spins
// for a fixed number of clock ticks on `blocks` workgroups with `lds` bytes of dynamic LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(long long ticks, int *sink) {
    extern __shared__ int sm[];
    sm[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) acc += sm[(threadIdx.x + acc) & 63];
    if (acc == 123456789) *sink = acc;
}
static int g_between = 0, g_prio = 0;     // streams created between s1 and s2; s2 with high priority
static float run(int pattern, int blocks1, int blocks2, int lds, long long ticks, int *sink) {
    hipStream_t s1, s2, dummy[8];
    hipEvent_t e0, e1, ef, ej;
    CK(hipStreamCreate(&s1));
    for (int i = 0; i < g_between; ++i) { CK(hipStreamCreate(&dummy[i])); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 1024, dummy[i], 1000LL, sink); }
    if (g_prio) {
        int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
    } else CK(hipStreamCreate(&s2));
    CK(hipDeviceSynchronize());
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipEventRecord(e0, s1));
    if (pattern == 0) {                 // both on one stream
        hipLaunchKernelGGL(spin, dim3(blocks1), dim3(256), lds, s1, ticks, sink);
        hipLaunchKernelGGL(spin, dim3(blocks2), dim3(256), lds, s1, ticks, sink);
    } else {                            // fork / join through events
        CK(hipEventRecord(ef, s1));
        CK(hipStreamWaitEvent(s2, ef, 0));
        hipLaunchKernelGGL(spin, dim3(blocks1), dim3(256), lds, s2, ticks, sink);
        CK(hipEventRecord(ej, s2));
        hipLaunchKernelGGL(spin, dim3(blocks2), dim3(256), lds, s1, ticks, sink);
        CK(hipStreamWaitEvent(s1, ej, 0));
    }
    CK(hipEventRecord(e1, s1));
    CK(hipStreamSynchronize(s1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
    for (int i = 0; i < g_between; ++i) CK(hipStreamDestroy(dummy[i]));
    return ms;
}
int main() {
    int *sink; CK(hipMalloc(&sink, 4));
    const long long ticks = 20000000;      // 100 MHz wall clock: 200 ms
    for (int prio = 0; prio < 2; ++prio)
        for (int between = 0; between <= 7; ++between) {
            g_between = between; g_prio = prio;
            printf("priority %d, %d streams created between the two: two streams %7.1f ms\n", prio, between,
                   run(1, 128, 128, 1024, ticks, sink));
        }
    g_between = 0; g_prio = 0;
    for (int lds : {1024, 140 * 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            printf("lds %6d B: one stream %7.1f ms, two streams %7.1f ms (128 + 128 workgroups)\n", lds,
                   run(0, 128, 128, lds, ticks, sink), run(1, 128, 128, lds, ticks, sink));
        }
        printf("lds %6d B: two streams, 128 + 480 workgroups: %7.1f ms\n", lds, run(1, 128, 480, lds, ticks, sink));
    }
    return 0;
}
