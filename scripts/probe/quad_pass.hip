// The GATE of layout 8 ("put a skeleton of the new pass on the box first; <= 8 000 cycles or stop").
// The new pass of the C3 site (500 rows x 32 columns, four chains): FOUR waves per workgroup, one per SIMD; wave c is a
// quarter of the row team (epx_quad.h: layout 7's row phase on the matrix pipe), the integrator of chain c (the view
// update with its real arithmetic, the relay of the finished state to vector order through LDS) and the bookkeeper of
// chain c -- here a SYNTHETIC one that spends a leaf's worth of vector instructions, wave sums and LDS stack accesses,
// with the merge depth of a real leaf index and a subtree end every 2^depth leaves.  Two workgroup barriers per pass, no
// wave waits for another ROLE.  No NUTS.  Layout 7 at the same site: 9 560 cycles per pass (profiles/r04_c3_instruction_mix.json).
//
//   quad_pass [workgroups] [passes] [bookkeeping work x100] [mode bits: 1 rows, 2 view update + relay, 4 bookkeeping]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "epx_quad.h"
using namespace epx;

constexpr int DP = 32, D = 32, NROWS = 500, P = 3 * D + 3, d = 2 * D + 2, PS = (P + 1) & ~1;
using S = Q8Slot<DP>;
constexpr int STK_LDS = 3;

struct ProbeArgs {
    const double *X; const uint8_t *y; const double *Om; const double *mu;
    unsigned long long *out; double *sink; double *gstack;
    int passes, work_pct, mode, off_slot, off_scr, off_stack;
};

__device__ inline void fake_fma(double &a, double &b, int cnt, double c) {
    for (int i = 0; i < cnt; ++i) { a = fma(a, 0.999999, c); b = fma(b, 1.000001, -c); }
}

__global__ void __launch_bounds__(256) k_probe(ProbeArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *Xs = reinterpret_cast<double *>(smem);
    constexpr int SPR = DP / 2;
    for (int s = tid; s < q8_rows(NROWS) * SPR; s += blockDim.x) {
        const int r = s / SPR, jp = s % SPR;
        double2 v;
        if (r >= NROWS) { v.x = 0; v.y = 0; }
        else { v.x = a.X[(size_t)r * D + 2 * jp]; v.y = a.X[(size_t)r * D + 2 * jp + 1]; }
        *reinterpret_cast<double2 *>(Xs + (size_t)r * DP + 2 * (jp ^ (r & (SPR - 1)))) = v;
    }
    double *s0 = reinterpret_cast<double *>(smem + a.off_slot);
    for (int i = tid; i < 4 * S::DOUBLES; i += blockDim.x) s0[i] = 0.0;
    double *k0 = reinterpret_cast<double *>(smem + a.off_stack);
    for (int i = tid; i < 4 * STK_LDS * 2 * PS; i += blockDim.x) k0[i] = 1e-3 * (i % 97);
    __syncthreads();
    q8_lds *slot0 = reinterpret_cast<q8_lds *>((uintptr_t)(unsigned)(size_t)(smem + a.off_slot));
    q8_lds *sc = slot0 + wave * S::DOUBLES;
    q8_lds *scr = reinterpret_cast<q8_lds *>((uintptr_t)(unsigned)(size_t)(smem + a.off_scr)) + wave * PS;
    q8_lds *stk = reinterpret_cast<q8_lds *>((uintptr_t)(unsigned)(size_t)(smem + a.off_stack)) + wave * STK_LDS * 2 * PS;
    double *gst = a.gstack + ((size_t)blockIdx.x * 4 + wave) * 12 * 256;

    QuadRows<DP> rows;
    rows.init((unsigned)(size_t)Xs, NROWS, a.y, a.Om, d, wave, lane, slot0);
    constexpr int LA = 32;
    const bool v_lane = lane < D || lane == LA;
    const int ve1 = !v_lane ? 0 : (lane == LA ? 0 : 2 + lane), ve2 = !v_lane ? 0 : (lane == LA ? d : d + 1 + lane), ve3 = !v_lane ? 0 : (lane == LA ? 1 : 2 + D + lane);
    const int tj = lane == LA ? DP : (lane < DP ? lane : 0);
    const double vmu1 = a.mu[ve1], vmu3 = a.mu[ve3];
    double vq1 = 0.02 * ((lane * 7 + wave) % 9 - 4), vq2 = 0.01 * ((lane * 5 + wave) % 7 - 3), vq3 = -0.5 + 0.01 * (lane % 5);
    double vp1 = 0.1, vp2 = -0.1, vp3 = 0.05, vm1 = 1.0, vm2 = 1.0, vm3 = 1.0, vex3 = exp_d(vq3);
    const double eps_l = 0.003;
    double acc0 = 1.0, acc1 = 0.5;
    double inv0 = 1.0, inv1 = 1.0;
    int leaf = 0, depth = 6, nleaf = 1 << depth;
    const int W = a.work_pct, e0 = lane, e1 = lane + 64;
    unsigned long long tph[4] = {0, 0, 0, 0};
    // first job
    {
        const double ba = vq1 + vq2 * vex3;
        if (lane < DP) sc[S::BOFF + lane] = ba;
        if (lane == LA) sc[0] = ba;
        if (v_lane) { sc[S::VOFF + ve1] = vq1 - vmu1; sc[S::VOFF + ve3] = vq3 - vmu3; }
    }
    q8_barrier();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long tp = c0;
#define PH(i_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[i_] += t_ - tp; tp = t_; } while (0)
    for (int pass = 0; pass < a.passes; ++pass) {
        if (a.mode & 1) rows.pass();
        PH(0);
        q8_barrier();                                                      // "the results are in"
        PH(1);
        if (a.mode & 2) {
            // ---- finish the leapfrog on the view (nuts_duo.hip's formulas)
            const double r0 = sc[S::RESO + 0 * S::RREC + tj], r1 = sc[S::RESO + 1 * S::RREC + tj], r2 = sc[S::RESO + 2 * S::RREC + tj], r3 = sc[S::RESO + 3 * S::RREC + tj];
            const double vo1 = sc[S::OVOFF + ve1], vo3 = sc[S::OVOFF + ve3];
            const double l4 = (((0.0 + sc[S::RESO + 0 * S::RREC + DP + 1]) + sc[S::RESO + 1 * S::RREC + DP + 1]) + sc[S::RESO + 2 * S::RREC + DP + 1]) + sc[S::RESO + 3 * S::RREC + DP + 1];
            const double t = (((0.0 + r0) + r1) + r2) + r3;
            const double g1 = -vo1 + t, g2 = t * vex3 - vq2, g3 = -vo3 + t * vq2 * vex3;
            const double q1o = vq1, q2o = vq2, q3o = vq3;
            const double fp1 = vp1 + 0.5 * eps_l * g1, fp2 = vp2 + 0.5 * eps_l * g2, fp3 = vp3 + 0.5 * eps_l * g3;
            const double lp1 = -0.5 * (q1o - vmu1) * vo1, lp3 = -0.5 * (q3o - vmu3) * vo3, lp2 = -0.5 * q2o * q2o;
            // ---- the finished state to vector order: three vectors through the chain's own slot (private to this wave between
            // the two barriers), the fourth through the scratch line
            if (v_lane) {
                sc[ve1] = q1o; sc[ve2] = q2o; sc[ve3] = q3o;
                sc[PS + ve1] = fp1; sc[PS + ve2] = fp2; sc[PS + ve3] = fp3;
                sc[2 * PS + ve1] = g1; sc[2 * PS + ve2] = g2; sc[2 * PS + ve3] = g3;
                scr[ve1] = lp1; scr[ve2] = lp2; scr[ve3] = lp3;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double zq0 = sc[e0], zq1 = e1 < P ? sc[e1] : 0.0, zp0 = sc[PS + e0], zp1 = e1 < P ? sc[PS + e1] : 0.0;
            const double zg0 = sc[2 * PS + e0], zg1 = e1 < P ? sc[2 * PS + e1] : 0.0;
            double lpt = scr[e0] + (e1 < P ? scr[e1] : 0.0);
            double ks = inv0 * zp0 * zp0 + inv1 * zp1 * zp1;
            PH(2);
            if (a.mode & 4) {
                // ---- synthetic books of a leaf: energy, Gumbel key, merges with the pending left siblings, park
                wave_sum2_packed(lpt, ks);
                acc0 += lpt + l4; acc1 += 0.5 * ks;
                fake_fma(acc0, acc1, (25 * W) / 100, zq0 + zg0);
                int ii = leaf, l = 0;
                double nr0 = zp0, nr1 = zp1;
                while (ii & 1) {
                    double L0, L1, L2, L3;
                    if (l < STK_LDS) { L0 = stk[(2 * l) * PS + e0]; L1 = e1 < P ? stk[(2 * l) * PS + e1] : 0.0; L2 = stk[(2 * l + 1) * PS + e0]; L3 = e1 < P ? stk[(2 * l + 1) * PS + e1] : 0.0; }
                    else { L0 = gst[l * 256 + e0]; L1 = gst[l * 256 + 64 + e0]; L2 = gst[l * 256 + 128 + e0]; L3 = gst[l * 256 + 192 + e0]; }
                    nr0 += L0; nr1 += L1;
                    double c1 = zp0 * nr0 + zp1 * nr1, c2 = L2 * nr0 + L3 * nr1;
                    wave_sum2_packed(c1, c2);
                    fake_fma(acc0, acc1, (4 * W) / 100, c1);
                    if (!(c1 > -1e300 && c2 > -1e300)) break;
                    ii >>= 1; ++l;
                }
                if (l < STK_LDS) { stk[(2 * l) * PS + e0] = nr0; if (e1 < P) stk[(2 * l) * PS + e1] = nr1; stk[(2 * l + 1) * PS + e0] = zp0; if (e1 < P) stk[(2 * l + 1) * PS + e1] = zp1; }
                else { gst[l * 256 + e0] = nr0; gst[l * 256 + 64 + e0] = nr1; gst[l * 256 + 128 + e0] = zp0; gst[l * 256 + 192 + e0] = zp1; }
                ++leaf;
                if ((leaf & 63) == 0) fake_fma(acc0, acc1, (250 * W) / 100, 1e-9);      // flush_dh, Philox, Gumbel keys of the next 64 leaves
                if (leaf == nleaf) {                                                     // a subtree ends: weights, draw, U-turn, new doubling
                    fake_fma(acc0, acc1, (450 * W) / 100, 1e-9);
                    wave_sum2_packed(acc0, acc1);
                    leaf = 0; depth = depth >= 10 ? 5 : depth + 1; nleaf = 1 << depth;
                    inv0 = 1.0 + 1e-12 * acc0; inv1 = 1.0;
                }
            }
            PH(3);
            // ---- the trajectory goes on: second half kick, first half of the next leapfrog, the next job
            vp1 = fp1 + 0.5 * eps_l * g1; vp2 = fp2 + 0.5 * eps_l * g2; vp3 = fp3 + 0.5 * eps_l * g3;
            vq1 = vq1 + eps_l * vm1 * vp1; vq2 = vq2 + eps_l * vm2 * vp2; vq3 = vq3 + eps_l * vm3 * vp3;
            vex3 = exp_d_vc(vq3);
            const double ba = vq1 + vq2 * vex3;
            // (the slot served as the relay's scratch: the job and V are written last, whole)
            if (lane < DP) sc[S::BOFF + lane] = lane < D ? ba : 0.0;
            if (lane == LA) sc[0] = ba;
            for (int e = lane; e < S::VN; e += 64) sc[S::VOFF + e] = 0.0;
            if (v_lane) { sc[S::VOFF + ve1] = vq1 - vmu1; sc[S::VOFF + ve3] = vq3 - vmu3; }
        }
        PH(2);
        q8_barrier();                                                      // "the jobs are in"
        PH(1);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        unsigned long long *o = a.out + (size_t)blockIdx.x * 32 + wave * 8;
        o[0] = c1 - c0; o[1] = tph[0]; o[2] = tph[1]; o[3] = tph[2]; o[4] = tph[3];
    }
    a.sink[(size_t)blockIdx.x * 256 + tid] = acc0 + acc1 + vq1 + vq2 + vq3;
}

int main(int argc, char **argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 256;
    const int passes = argc > 2 ? atoi(argv[2]) : 20000;
    const int work = argc > 3 ? atoi(argv[3]) : 100;
    const int mode = argc > 4 ? atoi(argv[4]) : 7;
    std::vector<double> X((size_t)NROWS * D), Om((size_t)d * d, 0.0), mu(d, 0.0);
    std::vector<uint8_t> y(NROWS);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24); };
    for (auto &v : X) v = 2.0 * rnd() - 1.0;
    for (auto &v : y) v = rnd() < 0.6;
    for (int i = 0; i < d; ++i) { Om[(size_t)i * d + i] = 1.0; mu[i] = 0.1 * (rnd() - 0.5); }
    for (int i = 0; i + 1 < d; ++i) { Om[(size_t)i * d + i + 1] = 0.05; Om[(size_t)(i + 1) * d + i] = 0.05; }
    ProbeArgs a;
    double *dX, *dOm, *dmu; uint8_t *dy;
    (void)hipMalloc(&dX, X.size() * 8); (void)hipMalloc(&dOm, Om.size() * 8); (void)hipMalloc(&dmu, mu.size() * 8); (void)hipMalloc(&dy, y.size());
    (void)hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dOm, Om.data(), Om.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dmu, mu.data(), mu.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dy, y.data(), y.size(), hipMemcpyHostToDevice);
    a.X = dX; a.y = dy; a.Om = dOm; a.mu = dmu;
    (void)hipMalloc(&a.out, (size_t)nblk * 32 * 8); (void)hipMalloc(&a.sink, (size_t)nblk * 256 * 8);
    (void)hipMalloc(&a.gstack, (size_t)nblk * 4 * 12 * 256 * 8); (void)hipMemset(a.gstack, 0, (size_t)nblk * 4 * 12 * 256 * 8);
    a.passes = passes; a.work_pct = work; a.mode = mode;
    size_t off = (size_t)q8_rows(NROWS) * DP * 8;
    a.off_slot = (int)off; off += (size_t)4 * S::DOUBLES * 8;
    a.off_scr = (int)off; off += (size_t)4 * PS * 8;
    a.off_stack = (int)off; off += (size_t)4 * STK_LDS * 2 * PS * 8;
    printf("LDS: %zu B (rows %d, slots %d, scratch lines %d, %d stack levels %d)\n", off, q8_rows(NROWS) * DP * 8, 4 * S::DOUBLES * 8, 4 * PS * 8, STK_LDS, 4 * STK_LDS * 2 * PS * 8);
    if (off > 160 * 1024) { printf("does not fit\n"); return 1; }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)off);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(a.out, 0, (size_t)nblk * 32 * 8);
        hipLaunchKernelGGL(k_probe, dim3(nblk), dim3(256), off, 0, a);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    }
    std::vector<unsigned long long> h((size_t)nblk * 32);
    (void)hipMemcpy(h.data(), a.out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cpp, p0, p1, p2, p3;
    for (int b = 0; b < nblk; ++b) {
        const unsigned long long *o = &h[(size_t)b * 32];
        double worst = 0;
        for (int w = 0; w < 4; ++w) worst = std::max(worst, (double)o[w * 8]);
        cpp.push_back(worst / passes);
        p0.push_back(o[1] / (double)passes); p1.push_back(o[2] / (double)passes); p2.push_back(o[3] / (double)passes); p3.push_back(o[4] / (double)passes);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("workgroups %d, passes %d, bookkeeping work x%.2f, mode %d\n", nblk, passes, work / 100.0, mode);
    printf("cycles per pass (median over workgroups): %.0f   [min %.0f max %.0f]\n", med(cpp), *std::min_element(cpp.begin(), cpp.end()), *std::max_element(cpp.begin(), cpp.end()));
    printf("  wave 0, per pass: rows %.0f | the two barriers %.0f | view update + relay + job %.0f | books %.0f\n", med(p0), med(p1), med(p2), med(p3));
    return 0;
}
