// The GATE of layout 8: the new pass of the C3 site -- the row team of epx_team8.h, which
// integrates the view itself behind two team-only barriers -- with SYNTHETIC bookkeepers on waves 0..3 that follow the real
// mailbox protocol (mail / early acknowledgement / control records / restarts) and spend a realistic number of vector
// instructions, LDS stack accesses and wave sums per finished state.  No NUTS: what the probe answers is what a pass
// costs when no wave of the team ever waits for a bookkeeper.  Target: <= 8 000 cycles per pass (layout 7: 9 560).
//
//   team8_pass [workgroups] [passes] [work multiplier x100] [row-wave priority]
//
// Output: cycles per pass (s_memtime of the slowest row wave / its passes), the shares of the pass's four phases, the
// share of passes in which a chain lost its turn (mailbox not free), per-bookkeeper busy share.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "epx_team8.h"
using namespace epx;

constexpr int DP = 32, D = 32, NROWS = 500, P = 3 * D + 3, d = 2 * D + 2, PS = (P + 1) & ~1;
constexpr int ADB = 4 * PS + 4, SDB = t8_slot_doubles(DP);
constexpr int STK_LEVELS_LDS = 1;

struct ProbeArgs {
    const double *X; const uint8_t *y; const double *Om; const double *mu;
    unsigned long long *out;        // per workgroup 32 words
    double *sink; double *gstack;   // per workgroup 4 x 12 x 256 doubles
    int *err;
    int leaves_total, work_pct, prio, off_slot, off_area, off_words, off_stack;
};

// a bookkeeper's vector work: `cnt` dependent-pair FMAs on two chains (ILP 2, like the state machine's FORV pairs)
__device__ inline void fake_fma(double &a, double &b, int cnt, double c) {
    for (int i = 0; i < cnt; ++i) { a = fma(a, 0.999999, c); b = fma(b, 1.000001, -c); }
}

__global__ void __launch_bounds__(512) k_probe(ProbeArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *Xs = reinterpret_cast<double *>(smem);
    constexpr int SPR = DP / 2;
    const int nrows_l = t8_rows(NROWS);
    for (int s = tid; s < nrows_l * SPR; s += blockDim.x) {
        const int r = s / SPR, jp = s % SPR;
        double2 v;
        if (r >= NROWS) { v.x = 0; v.y = 0; }
        else { v.x = a.X[(size_t)r * D + 2 * jp]; v.y = a.X[(size_t)r * D + 2 * jp + 1]; }
        const int sw = r & (SPR - 1);
        *reinterpret_cast<double2 *>(Xs + (size_t)r * DP + 2 * (jp ^ sw)) = v;
    }
    double *s0 = reinterpret_cast<double *>(smem + a.off_slot);
    for (int i = tid; i < 4 * SDB; i += blockDim.x) s0[i] = 0.0;
    double *a0 = reinterpret_cast<double *>(smem + a.off_area);
    for (int i = tid; i < 4 * ADB; i += blockDim.x) a0[i] = 0.0;
    double *k0 = reinterpret_cast<double *>(smem + a.off_stack);
    for (int i = tid; i < 4 * STK_LEVELS_LDS * 2 * PS; i += blockDim.x) k0[i] = 0.001 * i;
    volatile int *w0 = reinterpret_cast<volatile int *>(smem + a.off_words);
    if (tid < T8_WORDS) w0[tid] = tid == T8_W_LIVE ? 4 : 0;
    __syncthreads();
    t8_word *words = reinterpret_cast<t8_word *>((uintptr_t)(unsigned)(size_t)(smem + a.off_words));
    t8_lds *slot0 = reinterpret_cast<t8_lds *>((uintptr_t)(unsigned)(size_t)(smem + a.off_slot));
    t8_lds *area0 = reinterpret_cast<t8_lds *>((uintptr_t)(unsigned)(size_t)(smem + a.off_area));

    if (wave >= 4) {
        Team8Site s;
        s.xbase = (unsigned)(size_t)Xs; s.n = NROWS; s.y = a.y; s.Om_g = a.Om; s.mu_g = a.mu;
        s.D = D; s.d = d; s.nch = 4; s.laplace = false;
        s.slot0 = slot0; s.sdb = SDB; s.area0 = area0; s.adb = ADB; s.PS = PS; s.words = words; s.err = a.err;
        unsigned long long tacc[4] = {0, 0, 0, 0};
        if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        team8_row_wave<DP, true>(s, wave - 4, lane, tacc);
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long *o = a.out + (size_t)blockIdx.x * 32 + (wave - 4) * 6;
            o[0] = c1 - c0; o[1] = tacc[0]; o[2] = tacc[1]; o[3] = tacc[2]; o[4] = tacc[3];
            o[5] = (unsigned long long)words[T8_W_TB2] / 4;                 // passes
        }
        return;
    }
    // ================================================================= synthetic bookkeeper of chain `wave`
    const int c = wave;
    t8_lds *A = area0 + c * ADB;
    t8_word *w_mail = words + T8_W_PER_CHAIN * c + T8_W_MAIL, *w_ack = words + T8_W_PER_CHAIN * c + T8_W_ACK, *w_ctl = words + T8_W_PER_CHAIN * c + T8_W_CTL;
    t8_lds *stk = reinterpret_cast<t8_lds *>((uintptr_t)(unsigned)(size_t)(smem + a.off_stack)) + c * STK_LEVELS_LDS * 2 * PS;
    double *gst = a.gstack + ((size_t)blockIdx.x * 4 + c) * 12 * 256;
    const int e0 = lane, e1 = lane + 64;
    auto post = [&](double q0, double q1, double p0, double p1, double g0, double g1, int cmd, int consumed, int &gen) {
        A[e0] = q0; if (e1 < P) A[e1] = q1;
        A[PS + e0] = p0; if (e1 < P) A[PS + e1] = p1;
        A[2 * PS + e0] = g0; if (e1 < P) A[2 * PS + e1] = g1;
        A[3 * PS + e0] = 1.0; if (e1 < P) A[3 * PS + e1] = 1.0;
        if (lane == 0) { A[4 * PS + T8_S_EPS] = 0.004; A[4 * PS + T8_S_CMD] = (double)cmd; A[4 * PS + T8_S_CONSUMED] = (double)consumed; }
        ++gen;
        asm volatile("" ::: "memory");
        *w_ctl = gen;
        asm volatile("" ::: "memory");
        *w_ack = consumed;
    };
    int gen = 0;
    {
        const double q0 = 0.05 * ((lane * 37 + c * 11) % 17 - 8) / 8.0, q1 = e1 < P ? 0.03 * ((lane * 13 + c) % 11 - 5) / 5.0 : 0.0;
        post(q0, q1, 0.3, e1 < P ? -0.2 : 0.0, 0.0, 0.0, T8_CMD_RESTART, 0, gen);
    }
    unsigned long long busy = 0, waited = 0;
    int leaf = 0, depth = 6, nleaf = 1 << depth, restarts = 0, stalls_seen = 0;
    double acc0 = 1.0, acc1 = 0.5;
    const int W = a.work_pct;
    int m = 0;
    for (int done = 0; done < a.leaves_total; ++done) {
        ++m;
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
        {
            bool ok = false;
            for (int spin = 0; spin < T8_SPIN_LIMIT; ++spin) {
                const int v = __builtin_amdgcn_readfirstlane(*w_mail);
                if (v >= m) { ok = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
            if (!ok || m >= T8_GONE) { if (lane == 0) atomicOr(a.err, 8); return; }
        }
        const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
        waited += tw1 - tw0;
        // the entry to registers
        double zq0 = A[e0], zq1 = e1 < P ? A[e1] : 0.0, zp0 = A[PS + e0], zp1 = e1 < P ? A[PS + e1] : 0.0;
        double zg0 = A[2 * PS + e0], zg1 = e1 < P ? A[2 * PS + e1] : 0.0;
        double lpt = A[3 * PS + e0] + (e1 < P ? A[3 * PS + e1] : 0.0);
        const double ll = A[4 * PS + T8_S_LL];
        const bool last = leaf == nleaf - 1;
        bool acked = false;
        if (!last) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); *w_ack = m; acked = true; }
        // ---- the books of an ordinary leaf: energy, Gumbel key, merges with the pending left siblings
        double ks = zp0 * zp0 + zp1 * zp1;
        wave_sum2_packed(lpt, ks);
        acc0 += lpt + ll; acc1 += 0.5 * ks;
        fake_fma(acc0, acc1, (30 * W) / 100, zq0);
        int ii = leaf, l = 0;
        double nr0 = zp0, nr1 = zp1;
        while (ii & 1) {
            double L0, L1, L2, L3;
            if (l < STK_LEVELS_LDS) { L0 = stk[(2 * l) * PS + e0]; L1 = e1 < P ? stk[(2 * l) * PS + e1] : 0.0; L2 = stk[(2 * l + 1) * PS + e0]; L3 = e1 < P ? stk[(2 * l + 1) * PS + e1] : 0.0; }
            else { L0 = gst[l * 256 + e0]; L1 = gst[l * 256 + 64 + e0]; L2 = gst[l * 256 + 128 + e0]; L3 = gst[l * 256 + 192 + e0]; }
            nr0 += L0; nr1 += L1;
            double c1 = zp0 * nr0 + zp1 * nr1, c2 = L2 * nr0 + L3 * nr1;
            wave_sum2_packed(c1, c2);
            fake_fma(acc0, acc1, (6 * W) / 100, c1);
            if (!(c1 > -1e300 && c2 > -1e300)) break;
            ii >>= 1; ++l;
        }
        if (l < STK_LEVELS_LDS) { stk[(2 * l) * PS + e0] = nr0; if (e1 < P) stk[(2 * l) * PS + e1] = nr1; stk[(2 * l + 1) * PS + e0] = zp0; if (e1 < P) stk[(2 * l + 1) * PS + e1] = zp1; }
        else { gst[l * 256 + e0] = nr0; gst[l * 256 + 64 + e0] = nr1; gst[l * 256 + 128 + e0] = zp0; gst[l * 256 + 192 + e0] = zp1; }
        ++leaf;
        if ((leaf & 63) == 0) fake_fma(acc0, acc1, (250 * W) / 100, 1e-9);       // flush_dh + Philox + Gumbel keys of the next 64 leaves
        bool moved = false;
        if (last) {
            // a subtree ends: weights, multinomial draw, U-turn of the whole tree, tree ends through the cold store
            fake_fma(acc0, acc1, (700 * W) / 100, 1e-9);
            gst[11 * 256 + e0] = zq0; gst[11 * 256 + 64 + e0] = zq1;
            const double back = gst[11 * 256 + ((e0 + 1) & 63)];
            acc0 += back * 1e-30;
            leaf = 0;
            depth = depth >= 10 ? 5 : depth + 1; nleaf = 1 << depth;
            moved = (restarts++ & 1) == 0;                                     // half of the doublings go to the other end
        }
        if (!last && (done % 1531) == 1530) moved = true;                      // a rare U-turn inside an ordinary leaf's merges
        if (moved) {
            if (acked) {
                // the one entry the early acknowledgement allowed: wait for it, drop it
                ++m;
                bool ok = false;
                for (int spin = 0; spin < T8_SPIN_LIMIT; ++spin) {
                    const int v = __builtin_amdgcn_readfirstlane(*w_mail);
                    if (v >= m) { ok = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");
                if (!ok || m >= T8_GONE) { if (lane == 0) atomicOr(a.err, 8); return; }
                ++stalls_seen;
            }
            post(zq0, zq1, zp0, zp1, zg0, zg1, T8_CMD_RESTART, m, gen);
        } else if (!acked) { asm volatile("" ::: "memory"); *w_ack = m; }
        busy += __builtin_amdgcn_s_memtime() - tw1;
    }
    // the chain is done
    {
        ++m;
        for (int spin = 0; spin < T8_SPIN_LIMIT; ++spin) {
            const int v = __builtin_amdgcn_readfirstlane(*w_mail);
            if (v >= m) break;
            __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
        post(0, 0, 0, 0, 0, 0, T8_CMD_LEAVE, m, gen);
    }
    if (lane == 0) {
        unsigned long long *o = a.out + (size_t)blockIdx.x * 32 + 24 + c * 2;
        o[0] = busy; o[1] = waited;
    }
    a.sink[(size_t)blockIdx.x * 512 + tid] = acc0 + acc1;
}

int main(int argc, char **argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 256;
    const int leaves = argc > 2 ? atoi(argv[2]) : 20000;
    const int work = argc > 3 ? atoi(argv[3]) : 100;
    const int prio = argc > 4 ? atoi(argv[4]) : 0;
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
    (void)hipMalloc(&a.out, (size_t)nblk * 32 * 8); (void)hipMalloc(&a.sink, (size_t)nblk * 512 * 8);
    (void)hipMalloc(&a.gstack, (size_t)nblk * 4 * 12 * 256 * 8); (void)hipMemset(a.gstack, 0, (size_t)nblk * 4 * 12 * 256 * 8);
    (void)hipMalloc(&a.err, 4); (void)hipMemset(a.err, 0, 4);
    a.leaves_total = leaves; a.work_pct = work; a.prio = prio;
    size_t off = (size_t)t8_rows(NROWS) * DP * 8;
    a.off_slot = (int)off; off += (size_t)4 * SDB * 8;
    a.off_area = (int)off; off += (size_t)4 * ADB * 8;
    a.off_words = (int)off; off += 96;
    a.off_stack = (int)off; off += (size_t)4 * STK_LEVELS_LDS * 2 * PS * 8;
    printf("LDS: %zu B (rows %d, slots %d, areas %d, stack levels in LDS %d)\n", off, t8_rows(NROWS) * DP * 8, 4 * SDB * 8, 4 * ADB * 8, STK_LEVELS_LDS);
    if (off > 160 * 1024) { printf("does not fit\n"); return 1; }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)off);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(a.out, 0, (size_t)nblk * 32 * 8);
        hipLaunchKernelGGL(k_probe, dim3(nblk), dim3(512), off, 0, a);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    }
    int herr = 0; (void)hipMemcpy(&herr, a.err, 4, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> h((size_t)nblk * 32);
    (void)hipMemcpy(h.data(), a.out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cpp, ph0, ph1, ph2, ph3, bz, lost;
    for (int b = 0; b < nblk; ++b) {
        const unsigned long long *o = &h[(size_t)b * 32];
        double worst = 0, passes = (double)o[5];
        for (int w = 0; w < 4; ++w) worst = std::max(worst, (double)o[w * 6]);
        if (passes < 1) continue;
        cpp.push_back(worst / passes);
        ph0.push_back(o[1] / passes); ph1.push_back(o[2] / passes); ph2.push_back(o[3] / passes); ph3.push_back(o[4] / passes);
        double busy = 0; for (int c = 0; c < 4; ++c) busy += (double)o[24 + 2 * c] / (double)(o[24 + 2 * c] + o[25 + 2 * c] + 1);
        bz.push_back(busy / 4);
        lost.push_back(1.0 - (double)leaves / passes);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    printf("workgroups %d, leaves per chain %d, bookkeeper work x%.2f, row-wave priority %d, err %d\n", nblk, leaves, work / 100.0, prio, herr);
    printf("cycles per pass (median over workgroups; stamped build of the row wave): %.0f   [min %.0f max %.0f]\n", med(cpp),
           cpp.empty() ? 0.0 : *std::min_element(cpp.begin(), cpp.end()), cpp.empty() ? 0.0 : *std::max_element(cpp.begin(), cpp.end()));
    printf("  row wave 0, per pass: T1 wait %.0f | U (view update, mailbox) %.0f | T2 wait + operands %.0f | R (rows, cavity term) %.0f\n",
           med(ph0), med(ph1), med(ph2), med(ph3));
    printf("  passes beyond one per finished state (a chain's lost turns + restarts' dropped jobs): %.1f %%\n", 100 * med(lost));
    printf("  bookkeepers busy %.0f %% of their time (the rest: waiting for mail)\n", 100 * med(bz));
    return 0;
}
