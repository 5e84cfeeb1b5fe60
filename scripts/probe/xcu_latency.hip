// Probe for HISTORY.md section 7 item 4 (a helper workgroup on another CU integrating the other tree end):
// what does handing a state (3 vectors x 64 lanes x 8 B) from one workgroup to another through global memory cost?
// Producer workgroup: payload stores, release store of a sequence number.  Consumer workgroup: acquire-polls the
// number, reads the payload, checks it.  Ping-pong (the producer waits for an acknowledgement) gives the round trip;
// streaming (no acknowledgement, ring of 64 slots) gives the hand-over rate.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NV = 3, RING = 64, SPIN = 20000000;
struct Ctl { unsigned long long seq, ack; int bad, timeout; };

__device__ inline unsigned long long ld_acq(unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_rel(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void k(double *ring, Ctl *c, int n, int pingpong, int consumer_block, long long *cycles, int batch) {
    const int lane = threadIdx.x;
    if (blockIdx.x != 0 && blockIdx.x != consumer_block) return;
    const bool prod = blockIdx.x == 0;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= n; ++i) {
        double *slot = ring + (size_t)(i % RING) * NV * 64;
        if (prod) {
            if (!pingpong && i > RING) {            // do not lap the consumer
                int s = 0;
                while (ld_acq(&c->ack) + RING < (unsigned long long)i && ++s < SPIN) {}
                if (s >= SPIN) { c->timeout = 1; return; }
            }
            for (int v = 0; v < NV; ++v) slot[v * 64 + lane] = (double)i + 0.001 * lane + v;
            __syncthreads();                        // all lanes' stores issued (one wave: cheap)
            if (lane == 0 && (pingpong || i % batch == 0 || i == n)) st_rel(&c->seq, (unsigned long long)i);    // one release per `batch` states
            if (pingpong) {
                int s = 0;
                while (ld_acq(&c->ack) < (unsigned long long)i && ++s < SPIN) {}
                if (s >= SPIN) { c->timeout = 1; return; }
            }
        } else {
            int s = 0;
            while (ld_acq(&c->seq) < (unsigned long long)i && ++s < SPIN) {}
            if (s >= SPIN) { c->timeout = 2; return; }
            double chk = 0.0;
            for (int v = 0; v < NV; ++v) chk += __builtin_nontemporal_load(&slot[v * 64 + lane]);
            const double want = NV * ((double)i + 0.001 * lane) + 3.0;
            if (fabs(chk - want) > 1e-9) atomicAdd(&c->bad, 1);
            __syncthreads();
            if (lane == 0 && (pingpong || i % batch == 0 || i == n)) st_rel(&c->ack, (unsigned long long)i);
        }
    }
    if (lane == 0) cycles[prod ? 0 : 1] = wall_clock64() - t0;
}

int main() {
    double *ring; Ctl *c; long long *cyc;
    CK(hipMalloc(&ring, sizeof(double) * RING * NV * 64)); CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&cyc, 16));
    const int n = 20000;
    for (int consumer : {1, 8, 9, 255}) {           // workgroups are dealt round-robin to the 8 XCDs: 1 -> another XCD, 8 -> the same one
        for (int mode : {-1, 1, 4, 16}) {            // -1: ping-pong; else streaming with one release per `mode` states
            const int pp = mode < 0, batch = mode < 0 ? 1 : mode;
            CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(cyc, 0, 16));
            hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, ring, c, n, pp, consumer, cyc, batch);
            CK(hipDeviceSynchronize());
            Ctl h; long long hc[2];
            CK(hipMemcpy(&h, c, sizeof h, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
            printf("consumer workgroup %3d, %s (release every %2d): %.3f us per state (producer), %.3f us (consumer), wrong payloads %d, timeout %d\n",
                   consumer, pp ? "ping-pong " : "streaming ", batch, hc[0] / 100.0 / n, hc[1] / 100.0 / n, h.bad, h.timeout);
        }
    }
    return 0;
}
