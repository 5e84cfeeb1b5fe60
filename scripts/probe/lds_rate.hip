// What the LDS reads of the row team's pass cost on gfx950, and what else issues beside an FP64 matrix stream.
//   part 1: ds_read_b128 / ds_read_b64 / ds_read2_b64 throughput, 16 KB per batch and wave (the pass of nuts_duo.hip's TEAM
//           form reads 64 KB per wave and pass as ds_read_b128), linear addresses and the kernel's two swizzled patterns,
//           with one and with two waves per SIMD;
//   part 2: waves 0-3 run v_mfma_f64_4x4x4 back to back, waves 4-7 (same SIMDs) a stream of ONE kind of vector
//           instruction: which kinds share the FP64 pipe with the matrix instruction and which issue beside it.
// One workgroup of 256 or 512 threads on one CU; cycles by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RD64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define USE(x) asm volatile("" ::"v"(x))

constexpr int NREP = 512;

// PAT 0: linear (lane x 16 B, batches 1 KB apart); 1: the kernel's forward pattern (row = lane & 15 of a 16-row tile of 256-B
// rows, slot (4 r + hi) ^ row); 2: its backward pattern (row 4 bb + hi, slot (4 r + lo) ^ row)
template <int MODE, int PAT>
__global__ void __launch_bounds__(512) k_lds(unsigned long long *out, float *sink) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32768; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = (float)i;
    __syncthreads();
    const int lo = lane & 3, bb = (lane >> 2) & 3, hi = lane >> 4, rf = lane & 15, rb = 4 * bb + hi;
    unsigned a[4];
    const unsigned base = (unsigned)(size_t)smem + (unsigned)(wave & 3) * 32768u;
    for (int r = 0; r < 4; ++r) {
        if (PAT == 0) a[r] = base + lane * 16 + r * 1024;
        if (PAT == 1) a[r] = base + rf * 256 + (((4 * r + hi) ^ rf) << 4);
        if (PAT == 2) a[r] = base + rb * 256 + (((4 * r + lo) ^ rb) << 4);
    }
    // the same bytes as 8-byte reads: lane reads the first, then the second double of its 16-byte slot
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < NREP; ++it) {
        if (MODE == 0) {            // 16 ds_read_b128 = 16 KB per wave and batch (4 tiles x 4 slots)
            v4f d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11, d12, d13, d14, d15;
            RD128(d0, a[0], 0); RD128(d1, a[1], 0); RD128(d2, a[2], 0); RD128(d3, a[3], 0);
            RD128(d4, a[0], 4096); RD128(d5, a[1], 4096); RD128(d6, a[2], 4096); RD128(d7, a[3], 4096);
            RD128(d8, a[0], 8192); RD128(d9, a[1], 8192); RD128(d10, a[2], 8192); RD128(d11, a[3], 8192);
            RD128(d12, a[0], 12288); RD128(d13, a[1], 12288); RD128(d14, a[2], 12288); RD128(d15, a[3], 12288);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            USE(d0); USE(d1); USE(d2); USE(d3); USE(d4); USE(d5); USE(d6); USE(d7);
            USE(d8); USE(d9); USE(d10); USE(d11); USE(d12); USE(d13); USE(d14); USE(d15);
        } else if (MODE == 1) {     // the same 16 KB as 32 ds_read_b64
            v2f d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11, d12, d13, d14, d15;
            v2f e0, e1, e2, e3, e4, e5, e6, e7, e8, e9, e10, e11, e12, e13, e14, e15;
            RD64(d0, a[0], 0); RD64(e0, a[0], 8); RD64(d1, a[1], 0); RD64(e1, a[1], 8);
            RD64(d2, a[2], 0); RD64(e2, a[2], 8); RD64(d3, a[3], 0); RD64(e3, a[3], 8);
            RD64(d4, a[0], 4096); RD64(e4, a[0], 4104); RD64(d5, a[1], 4096); RD64(e5, a[1], 4104);
            RD64(d6, a[2], 4096); RD64(e6, a[2], 4104); RD64(d7, a[3], 4096); RD64(e7, a[3], 4104);
            RD64(d8, a[0], 8192); RD64(e8, a[0], 8200); RD64(d9, a[1], 8192); RD64(e9, a[1], 8200);
            RD64(d10, a[2], 8192); RD64(e10, a[2], 8200); RD64(d11, a[3], 8192); RD64(e11, a[3], 8200);
            RD64(d12, a[0], 12288); RD64(e12, a[0], 12296); RD64(d13, a[1], 12288); RD64(e13, a[1], 12296);
            RD64(d14, a[2], 12288); RD64(e14, a[2], 12296); RD64(d15, a[3], 12288); RD64(e15, a[3], 12296);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            USE(d0); USE(d1); USE(d2); USE(d3); USE(d4); USE(d5); USE(d6); USE(d7);
            USE(d8); USE(d9); USE(d10); USE(d11); USE(d12); USE(d13); USE(d14); USE(d15);
            USE(e0); USE(e1); USE(e2); USE(e3); USE(e4); USE(e5); USE(e6); USE(e7);
            USE(e8); USE(e9); USE(e10); USE(e11); USE(e12); USE(e13); USE(e14); USE(e15);
        } else {                    // MODE 2: 16 ds_read_b128 with a wait after every 4 (short batches)
            v4f d0, d1, d2, d3;
            for (int q = 0; q < 4; ++q) {
                RD128(d0, a[0], 0); RD128(d1, a[1], 0); RD128(d2, a[2], 0); RD128(d3, a[3], 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                USE(d0); USE(d1); USE(d2); USE(d3);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[wave] = t1 - t0;
    sink[tid] = acc;
}

template <int MODE, int PAT> void run_lds(const char *what, int nthreads) {
    unsigned long long *d; float *s;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 512 * 4);
    (void)hipMemset(d, 0, 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds<MODE, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_lds<MODE, PAT>), dim3(1), dim3(nthreads), 131072, 0, d, s);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    const double per = h[0] / (double)NREP;
    const int nw = nthreads / 64;
    printf("%-58s %d waves: %7.1f cycles per 16-KB batch of a wave = %5.1f B/clk/CU (waves 0..: %.0f %.0f %.0f %.0f)\n", what, nw, per,
           16384.0 * nw / per, h[0] / (double)NREP, h[1] / (double)NREP, h[2] / (double)NREP, h[3] / (double)NREP);
    (void)hipFree(d); (void)hipFree(s);
}

// ---------------------------------------------------------------- part 2
#define N2 2048
template <int KIND>
__global__ void __launch_bounds__(512) k_mix(unsigned long long *out, double *sink, double seed, int with_mfma) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double a = seed + lane, b = 1.0 + 1e-9 * lane;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    double v0 = a, v1 = b, v2 = a + 1, v3 = b + 1;
    unsigned u0 = lane, u1 = lane + 1, u2 = lane + 2, u3 = lane + 3;
    float f0 = lane, f1 = lane + 1.f, f2 = 2.f, f3 = 3.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (with_mfma == 2)
            for (int i = 0; i < N2 / 4; ++i) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v1) : "v"(b), "v"(a));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v2) : "v"(b), "v"(a)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v3) : "v"(b), "v"(a));
            }
        else if (with_mfma)
            for (int i = 0; i < N2 / 4; ++i) {
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
            }
    } else {
        for (int i = 0; i < N2 / 4; ++i) {
            if (KIND == 0) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v1) : "v"(b), "v"(a));
                             asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v2) : "v"(b), "v"(a)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v3) : "v"(b), "v"(a)); }
            if (KIND == 1) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(v0) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(v1) : "v"(b));
                             asm volatile("v_add_f64 %0, %0, %1" : "+v"(v2) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(v3) : "v"(b)); }
            if (KIND == 2) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u1) : "v"(u2));
                             asm volatile("v_add_u32 %0, %0, %1" : "+v"(u2) : "v"(u3)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u3) : "v"(u0)); }
            if (KIND == 3) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u0) : "v"(u1)); asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(u1) : "v"(u2));
                             asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(u2) : "v"(u3)); asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(u3) : "v"(u0)); }
            if (KIND == 4) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f2), "v"(f3)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
                             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(f1), "v"(f3)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f2), "v"(f1)); }
            if (KIND == 5) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u0) : "v"(u1)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u1) : "v"(u2));
                             asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u2) : "v"(u3)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u3) : "v"(u0)); }
            if (KIND == 6) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v0) : "v"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v1) : "v"(b));
                             asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v2) : "v"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v3) : "v"(b)); }
            if (KIND == 7) { asm volatile("v_max_f64 %0, %0, %1" : "+v"(v0) : "v"(b)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(v1) : "v"(b));
                             asm volatile("v_max_f64 %0, %0, %1" : "+v"(v2) : "v"(b)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(v3) : "v"(b)); }
            if (KIND == 8) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(b), "v"(a));
                             asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(b), "v"(a)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(b), "v"(a)); }
            if (KIND == 10) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u0) : "v"(u1) : "s10", "s11"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u1) : "v"(u2) : "s10", "s11");
                              asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u2) : "v"(u3) : "s10", "s11"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u3) : "v"(u0) : "s10", "s11"); }
            if (KIND == 11) { asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u0) : "v"(u1), "v"(u2) : "vcc"); asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u1) : "v"(u2), "v"(u3) : "vcc"); }
            if (KIND == 12) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a));
                              asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a)); }
            if (KIND == 13) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v1) : "v"(b), "v"(a));
                              asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(b), "v"(a)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v1) : "v"(b), "v"(a)); }
            if (KIND == 14) { asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v0) : "v"(u0)); asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v1) : "v"(u0));
                              asm volatile("v_rndne_f64 %0, %0" : "+v"(v2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u1) : "v"(v3)); }
            if (KIND == 9) { asm volatile("v_rcp_f64 %0, %0" : "+v"(v0)); asm volatile("v_rcp_f64 %0, %0" : "+v"(v1));
                             asm volatile("v_rcp_f64 %0, %0" : "+v"(v2)); asm volatile("v_rcp_f64 %0, %0" : "+v"(v3)); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = c0 + c1 + c2 + c3 + v0 + v1 + v2 + v3 + u0 + u1 + u2 + u3 + f0 + f1 + f2 + f3;
}

template <int KIND> void run_mix(const char *what) {
    unsigned long long *d; double *s;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 512 * 8);
    double r[3][2];
    for (int with = 0; with < 3; ++with) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_mix<KIND>, dim3(1), dim3(512), 0, 0, d, s, 1.0, with);
        (void)hipDeviceSynchronize();
        unsigned long long h[8];
        (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        r[with][0] = h[0] / (double)N2; r[with][1] = h[4] / (double)N2;
    }
    printf("%-22s alone %5.2f cycles per instruction | beside the matrix stream %5.2f (the matrix instruction: %5.2f) | beside a v_fma_f64 stream of the SIMD's other wave %5.2f (that stream: %5.2f)\n",
           what, r[0][1], r[1][1], r[1][0], r[2][1], r[2][0]);
    (void)hipFree(d); (void)hipFree(s);
}

int main() {
    run_lds<0, 0>("ds_read_b128, linear", 256);
    run_lds<0, 0>("ds_read_b128, linear", 512);
    run_lds<1, 0>("ds_read_b64 x 2, linear", 256);
    run_lds<1, 0>("ds_read_b64 x 2, linear", 512);
    run_lds<0, 1>("ds_read_b128, the kernel's forward pattern", 256);
    run_lds<0, 2>("ds_read_b128, the kernel's backward pattern", 256);
    run_lds<1, 1>("ds_read_b64 x 2, forward pattern", 256);
    run_lds<1, 2>("ds_read_b64 x 2, backward pattern", 256);
    run_lds<0, 1>("ds_read_b128, forward pattern", 512);
    run_lds<1, 1>("ds_read_b64 x 2, forward pattern", 512);
    run_lds<2, 0>("ds_read_b128, linear, batches of 4", 256);
    run_lds<2, 1>("ds_read_b128, forward pattern, batches of 4", 256);
    run_mix<0>("v_fma_f64");
    run_mix<1>("v_add_f64");
    run_mix<6>("v_mul_f64");
    run_mix<7>("v_max_f64");
    run_mix<9>("v_rcp_f64");
    run_mix<2>("v_add_u32");
    run_mix<3>("v_mov_b32_dpp");
    run_mix<5>("v_cndmask_b32");
    run_mix<4>("v_fma_f32");
    run_mix<8>("v_pk_fma_f32");
    run_mix<10>("v_cndmask_b32 e64 (sgpr mask)");
    run_mix<11>("v_cmp + v_cndmask (pair)");
    run_mix<12>("v_fma_f64, ONE dependent chain");
    run_mix<13>("v_fma_f64, two chains");
    run_mix<14>("ldexp ldexp rndne cvt");
    return 0;
}
