// Layout 8 -- the row team INTEGRATES, the state waves keep the books behind a mailbox (round 5).
//
// Replaces the Stan subprocess of /root/reference/epstan/method.py:43-118, 349-363 for the models with per-coefficient
// scales (m4b_sg / m5b_sg: /root/reference/experiment/models/m4b_sg.stan:19-43) at the site sizes that fill the chip.
//
// Layout 7 (nuts_duo.hip, TEAM form) couples eight waves by two workgroup barriers per pass: the four state waves update
// the "view" of their chain between "results in" and "jobs in" while the row team waits, and the row team multiplies
// while the state waves keep the books at one instruction per matrix instruction.  A subtree end of ONE chain parks four
// chains, the relay of every finished state from the view's lanes to vector order costs the state wave 60 instructions
// and eight LDS exchanges, and the stretch between the barriers is 1 500 of a pass's ~9 500 cycles (HISTORY.md 3.1g).
//
// Here the row waves own the leapfrog: row wave c carries the view of chain c (location, raw coefficient, log scale of
// column `lane`; alpha's triple on lane 32) in ten registers.  A pass is
//   R  every row wave takes its quarter of the site's 16-row tiles through F = alpha + X B, the logistic terms and
//      G += X' g on the matrix pipe for the four chains (layout 7's pass, same arithmetic in the same order) and its
//      16-row group of the cavity term Omega V; partial sums to LDS;
//   T1 team barrier (the four row waves only: a counter in LDS -- the state waves are not part of it);
//   U  row wave c sums the four partial X'g of chain c, finishes the leapfrog on the view (second half kick), hands the
//      FINISHED state to the chain's bookkeeper -- written in vector order into the chain's mailbox, which is the relay --
//      and takes the first half of the next leapfrog: the next job (alpha, beta, V = phi - mu) of chain c;
//   T2 team barrier; the operands of all four chains are read; next R.
// The state wave of chain c (a BOOKKEEPER now) waits for mail, runs nuts_state_machine.inc on the finished state -- beside
// the team's next pass, at whatever pace its SIMD leaves it -- and answers ONLY when the trajectory does not continue
// from the state it was handed (other tree end, new transition, step-size trial, new metric): a control record
// (position, momentum, gradient, metric, signed step size) with a new generation number, from which the row wave
// restarts the view.  No wave of the team ever waits for a bookkeeper except for a free mailbox, and then it is ONE chain
// that loses a pass (the team runs that chain's job again: same results), not four.
//
// Mailbox protocol (one area of 4 vectors + 4 scalars per chain, shared by both directions):
//   words per chain: mail (row wave -> bookkeeper: entries sent), ack (bookkeeper -> row wave: entries consumed),
//   ctl (bookkeeper -> row wave: generation of the control record).
//   * the row wave writes an entry only when ack == sent and ctl == its generation, and posts mail = ++sent behind it;
//   * the bookkeeper copies an entry to registers, and acknowledges AT ONCE when the leaf cannot end a subtree
//     (leaf != nleaf - 1 in tree mode), otherwise only when its bookkeeping says the trajectory goes on;
//   * a restart: the bookkeeper owns the area when it has NOT acknowledged (the row wave is stalling the chain), or after
//     it has waited for -- and dropped -- the one entry an early acknowledgement allowed; it writes the record, then
//     ctl = ++generation, then ack = consumed (the row wave reads ack BEFORE ctl: a new ack implies the new ctl);
//   * the row wave, on a new generation, re-loads the view from the record, takes `sent` from it, and posts the job.
// The LDS performs the operations of one wave in issue order, so a word stored behind data is seen behind the data.
//
// Same algorithm, same arithmetic as layout 7 operation by operation (the products' order, the view's formulas, the
// state machine): layout 8 returns layout 7's draws bit for bit (tests/test_gpu_round5.py).
#pragma once
#include "nuts_common.h"

namespace epx {

typedef double t8_v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) t8_v2 *t8_v2p;
typedef volatile __attribute__((address_space(3))) int t8_word;
typedef __attribute__((address_space(3))) int t8_iword;
typedef __attribute__((address_space(3))) double t8_lds;

// words (ints) at NutsArgs::off_flag: four per chain, then the team's
enum { T8_W_MAIL = 0, T8_W_ACK = 1, T8_W_CTL = 2, T8_W_PER_CHAIN = 4,
       T8_W_TB1 = 16, T8_W_TB2 = 17, T8_W_LIVE = 18, T8_W_PANIC = 19, T8_WORDS = 24 };
enum { T8_CMD_RESTART = 1, T8_CMD_LEAVE = 2 };
enum { T8_SPIN_LIMIT = 1 << 22, T8_GONE = 1 << 29 };
// area of a chain (NutsArgs::off_scr, scr_doubles = 4 PS + 4): vectors 0..3 of PS doubles, then
//   entry:   q, p (half-kicked twice = the finished momentum), g, per-element log-density terms | ll, -, -, -
//   control: q, p, g, metric                                                                    | eps_l, command, consumed, -
enum { T8_S_LL = 0, T8_S_EPS = 0, T8_S_CMD = 1, T8_S_CONSUMED = 2 };

__host__ __device__ inline int t8_tiles_per_wave(int n) { const int t = ((n + 15) / 16 + 3) / 4; return (t + 1) & ~1; }
__host__ __device__ inline int t8_rows(int n) { return 4 * t8_tiles_per_wave(n) * 16; }
__host__ __device__ constexpr int t8_vn(int dp) { return 2 * dp + 8; }                    // doubles of a V / Omega V line
__host__ __device__ constexpr int t8_slot_doubles(int dp) { return (dp + 2) + t8_vn(dp) + 4 * (dp + 2) + t8_vn(dp); }

__device__ inline double t8_mfma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ inline void t8_add(t8_word *w, int v) {
    __hip_atomic_fetch_add(const_cast<t8_iword *>(w), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// a look at a word that only grows; false when the wait gives up or the workgroup is in panic
__device__ inline bool t8_wait_ge(t8_word *w, int want, t8_word *panic) {
    for (int spin = 0; spin < T8_SPIN_LIMIT; ++spin) {
        const int v = __builtin_amdgcn_readfirstlane(*w);
        if (v >= want) { asm volatile("" ::: "memory"); return true; }
        if ((spin & 63) == 63 && __builtin_amdgcn_readfirstlane(*panic)) return false;
    }
    return false;
}
// every wait of the workgroup ends: the team's counters jump past any pass, no chain is live
__device__ inline void t8_panic(t8_word *words, int *err, int code, int lane) {
    if (lane == 0) {
        atomicOr(err, code);
        words[T8_W_PANIC] = 1; words[T8_W_LIVE] = 0; words[T8_W_TB1] = T8_GONE; words[T8_W_TB2] = T8_GONE;
#pragma unroll
        for (int c = 0; c < 4; ++c) { words[T8_W_PER_CHAIN * c + T8_W_MAIL] = T8_GONE; words[T8_W_PER_CHAIN * c + T8_W_ACK] = T8_GONE; }
    }
}

struct Team8Site {
    unsigned xbase;             // LDS byte address of the row images
    int n;                      // rows of the site
    const uint8_t *y;           // its responses
    const double *Om_g;         // cavity precision (d x d, column major, symmetric), global
    const double *mu_g;         // cavity mean (d), global
    int D, d, nch;              // covariates, dimension of phi, chains of this workgroup
    bool laplace;               // m5b
    t8_lds *slot0; int sdb;     // chain 0's slot, doubles per chain
    t8_lds *area0; int adb, PS; // chain 0's mailbox area, doubles per chain, doubles per vector
    t8_word *words;
    int *err;
};

// ---------------------------------------------------------------------------------------------------------------
// The row wave `wr` of the team (also the integrator of chain `wr`).  Returns when no chain of the workgroup is live.
// STAMP: cycle stamps of the pass's phases into `tacc` (diagnostics / probe): 0 T1 wait, 1 U, 2 T2 wait + operands, 3 R.
template <int DP, bool STAMPS>
__device__ __forceinline__ void team8_row_wave(const Team8Site &s, const int wr, const int lane, unsigned long long *tacc) {
    constexpr int SPR = DP / 2, RPL = DP >= 32 ? 1 : 32 / DP;
    constexpr int RES = DP + 2, VN = t8_vn(DP), BOFF = 2, VOFF = RES, RREC = RES, RESO = RES + VN, OVOFF = RESO + 4 * RREC;
    constexpr int KS = DP / 4, NRD = DP / 8, ROWB = DP * 8, TILEB = 16 * ROWB, TILEV = TILEB / 16;
    constexpr int DMAX = 2 * DP + 2, NJ = (DMAX + 3) / 4, NGF = DMAX / 16, NJT = (NJ + 3) / 4;
    static_assert(NGF <= 4, "one 16-row group of the cavity term per row wave");
    constexpr int LA = 32;
    const int n = s.n, D = s.D, d = s.d;
    constexpr int PS = 3 * DP + 4;
    const int lo = lane & 3, bb = (lane >> 2) & 3, hi = lane >> 4;
    const int tpw = t8_tiles_per_wave(n), t0 = wr * tpw, t1 = t0 + tpw;
    t8_word *const words = s.words;
    t8_word *const w_mail = words + T8_W_PER_CHAIN * wr + T8_W_MAIL, *const w_ack = words + T8_W_PER_CHAIN * wr + T8_W_ACK,
            *const w_ctl = words + T8_W_PER_CHAIN * wr + T8_W_CTL;
    t8_word *const w_panic = words + T8_W_PANIC;

#ifndef T8_OM_L2
    // ---- cavity precision as A operands (layout 7's: group wr, rows 16 wr + 4 bb + lo, columns 4 J + hi; wave 3 the tail rows)
    double om[NJ], omt[NJT];
    {
        const int e = 16 * wr + 4 * bb + lo;
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
            const int c = 4 * J + hi;
            om[J] = (wr < NGF && e < d && c < d) ? s.Om_g[(size_t)c * d + e] : 0.0;
        }
        const int et = 16 * NGF + lo;
#pragma unroll
        for (int tt = 0; tt < NJT; ++tt) {
            const int c = 4 * (4 * tt + bb) + hi;
            omt[tt] = (wr == 3 && et < d && c < d) ? s.Om_g[(size_t)c * d + et] : 0.0;
        }
    }
#else
    // ---- cavity precision as A operands (layout 7's: group wr, rows 16 wr + 4 bb + lo, columns 4 J + hi; wave 3 the tail
    // rows): fetched from the L2 once per pass, behind the rows and in front of the view update that hides the latency --
    // as loop-invariant registers they are 44 of the wave's 256 through the whole pass, and the allocator spills them
    const int om_e = 16 * wr + 4 * bb + lo, om_et = 16 * NGF + lo;
    const bool om_row = wr < NGF && om_e < d, omt_row = wr == 3 && om_et < d;
    double om[NJ], omt[NJT];
    auto fetch_om = [&]() {
        unsigned z = 0;
        asm volatile("" : "+v"(z));                      // (an offset the compiler cannot see through: the loads stay in the loop)
        const unsigned vo = z + (unsigned)(hi * d + om_e) * 8u, vt_ = z + (unsigned)(hi * d + om_et) * 8u;
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
            const char *bj = reinterpret_cast<const char *>(s.Om_g) + (size_t)(4 * J) * d * 8;       // (scalar)
            om[J] = (om_row && 4 * J + hi < d) ? *reinterpret_cast<const double *>(bj + vo) : 0.0;
        }
#pragma unroll
        for (int tt = 0; tt < NJT; ++tt) {
            const char *bj = reinterpret_cast<const char *>(s.Om_g) + (size_t)(4 * 4 * tt) * d * 8;
            omt[tt] = (omt_row && 4 * (4 * tt + bb) + hi < d) ? *reinterpret_cast<const double *>(bj + vt_ + (unsigned)(4 * bb * d) * 8u) : 0.0;
        }
    };
    fetch_om();
#endif
    const bool g_on = wr < NGF && 16 * wr < d, t_on = wr == 3 && d > 16 * NGF;
    const int rf = lane & 15, rb = 4 * bb + hi;
    unsigned ybits = 0;
    for (int t = t0; t < t1; ++t) {
        const int r = 16 * t + rb;
        if (r < n && s.y[r]) ybits |= 1u << (t - t0);
    }
    const unsigned swf = (unsigned)((rf / RPL) & (SPR - 1)), swb = (unsigned)((rb / RPL) & (SPR - 1));
    unsigned af[NRD], ab[NRD];
#pragma unroll
    for (int r = 0; r < NRD; ++r) {
        af[r] = s.xbase + (unsigned)t0 * TILEB + (unsigned)rf * ROWB + ((((unsigned)(4 * r + hi)) ^ swf) << 4);
        ab[r] = s.xbase + (unsigned)t0 * TILEB + (unsigned)rb * ROWB + ((((unsigned)(4 * r + lo)) ^ swb) << 4);
    }
    t8_lds *const sl = s.slot0 + lo * s.sdb;              // the slot of chain lo: this lane's column of the products
    t8_lds *const sc = s.slot0 + wr * s.sdb;              // the slot of chain wr: the chain this wave integrates
    t8_lds *const A = s.area0 + wr * s.adb;

    // ---- the view of chain wr (nuts_duo.hip's: lane j < D carries coefficient j, lane LA the intercept)
    const bool exists = wr < s.nch;
    double vmu1, vmu3;
    {
        const bool v_lane = lane < D || lane == LA;
        const int ve1 = !v_lane ? 0 : (lane == LA ? 0 : 2 + lane), ve3 = !v_lane ? 0 : (lane == LA ? 1 : 2 + D + lane);
        vmu1 = s.mu_g[ve1]; vmu3 = s.mu_g[ve3];
    }
    double vq1 = 0, vq2 = 0, vq3 = 0, vp1 = 0, vp2 = 0, vp3 = 0, vm1 = 1, vm2 = 1, vm3 = 1, vex3 = 1;
    double eps_l = 0.0;
    int sent = 0, gen = 0;
    bool armed = false, dead = !exists, mail_due = false;
    unsigned long long tprev = 0;
    if constexpr (STAMPS) tprev = __builtin_amdgcn_s_memtime();
#define T8_STAMP(i_) do { if constexpr (STAMPS) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); tacc[i_] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)

    for (int pass = 0;; ++pass) {
        // ================================================================= T1: the partial sums of this pass are in
        // (every look asks for what the update reads as well: the LDS serves a wave's reads in order, so values requested
        // behind a counter that reads "complete" are this pass's -- one round trip instead of two)
        // (the view's element indices are re-derived from an opaque copy of the lane index in every pass: hoisted out of the
        // loop they are a dozen address registers that stay live through the rows)
        int lane_u = lane;
        asm volatile("" : "+v"(lane_u));
        const bool v_lane = lane_u < D || lane_u == LA;
        const int ve1 = !v_lane ? 0 : (lane_u == LA ? 0 : 2 + lane_u);            // location:  mu_a | mu_b[j]
        const int ve2 = !v_lane ? 0 : (lane_u == LA ? d : d + 1 + lane_u);        // raw:       eta  | etb[j]
        const int ve3 = !v_lane ? 0 : (lane_u == LA ? 1 : 2 + D + lane_u);        // log scale: lsig_a | lsig_b[j]
        const int tj = lane_u == LA ? DP : (lane_u < DP ? lane_u : 0);            // its entry of the results: X'g[j] | sum g
        double r0 = 0, r1 = 0, r2 = 0, r3 = 0, vo1 = 0, vo3 = 0;
        int ackv = 0, ctlv = 0;
        {
            bool ok = false;
            for (int spin = 0; spin < T8_SPIN_LIMIT; ++spin) {
                asm volatile("" ::: "memory");
                const int v = words[T8_W_TB1];
                asm volatile("" ::: "memory");
                r0 = sc[RESO + 0 * RREC + tj]; r1 = sc[RESO + 1 * RREC + tj]; r2 = sc[RESO + 2 * RREC + tj]; r3 = sc[RESO + 3 * RREC + tj];
                vo1 = sc[OVOFF + ve1]; vo3 = sc[OVOFF + ve3];
                asm volatile("" ::: "memory");
                ackv = *w_ack;                              // (ack BEFORE ctl: see the protocol)
                asm volatile("" ::: "memory");
                ctlv = *w_ctl;
                asm volatile("" ::: "memory");
                if (__builtin_amdgcn_readfirstlane(v) >= 4 * pass) { ok = true; break; }
            }
            if (!ok) { t8_panic(words, s.err, 1, lane); return; }
        }
        ackv = __builtin_amdgcn_readfirstlane(ackv); ctlv = __builtin_amdgcn_readfirstlane(ctlv);
        T8_STAMP(0);
        // ================================================================= U: chain wr
        if (!dead) {
            bool newjob = false;
            if (ctlv != gen) {
                // ---- the bookkeeper says: continue from HERE (the job in flight, if any, is dropped): the first half of the
                // leapfrog from the record (nuts_duo.hip first_half(), element by element)
                gen = ctlv;
                vm1 = A[3 * PS + ve1]; vm2 = A[3 * PS + ve2]; vm3 = A[3 * PS + ve3];
                eps_l = uniform_d(A[4 * PS + T8_S_EPS]);
                const int cmd = __builtin_amdgcn_readfirstlane((int)A[4 * PS + T8_S_CMD]);
                sent = __builtin_amdgcn_readfirstlane((int)A[4 * PS + T8_S_CONSUMED]);
                vp1 = A[PS + ve1] + 0.5 * eps_l * A[2 * PS + ve1]; vp2 = A[PS + ve2] + 0.5 * eps_l * A[2 * PS + ve2]; vp3 = A[PS + ve3] + 0.5 * eps_l * A[2 * PS + ve3];
                vq1 = A[ve1] + eps_l * vm1 * vp1; vq2 = A[ve2] + eps_l * vm2 * vp2; vq3 = A[ve3] + eps_l * vm3 * vp3;
                armed = cmd != T8_CMD_LEAVE; newjob = armed;
                if (!armed) {
                    dead = true;
                    // (the column of a chain that is done stays finite: zeros)
                    if (lane_u < DP) sc[BOFF + lane_u] = 0.0;
                    if (lane_u == LA) sc[0] = 0.0;
                    if (lane_u < VN - 64) sc[VOFF + 64 + lane_u] = 0.0;
                    sc[VOFF + lane_u] = 0.0;
                    if (lane_u == 0) t8_add(words + T8_W_LIVE, -1);
                }
            } else if (armed && ackv == sent) {
                // ---- the results of the job in flight: second half of this leapfrog, first half of the next one
                const double t = (((0.0 + r0) + r1) + r2) + r3;      // (layout 7's order of additions: s = 0; s += ... in wave order)
                const double pr2 = s.laplace ? (double)((vq2 > 0) - (vq2 < 0)) : vq2;
                const double g1 = -vo1 + t, g2 = t * vex3 - pr2, g3 = -vo3 + t * vq2 * vex3;
                const double q1o = vq1, q2o = vq2, q3o = vq3;
                const double fp1 = vp1 + 0.5 * eps_l * g1, fp2 = vp2 + 0.5 * eps_l * g2, fp3 = vp3 + 0.5 * eps_l * g3;
                vp1 = fp1 + 0.5 * eps_l * g1; vp2 = fp2 + 0.5 * eps_l * g2; vp3 = fp3 + 0.5 * eps_l * g3;
                vq1 = vq1 + eps_l * vm1 * vp1; vq2 = vq2 + eps_l * vm2 * vp2; vq3 = vq3 + eps_l * vm3 * vp3;
                // the finished state to the bookkeeper, in vector order (the mailbox IS the relay)
                const double lp1 = -0.5 * (q1o - vmu1) * vo1, lp3 = -0.5 * (q3o - vmu3) * vo3;
                const double lp2 = s.laplace ? -fabs(q2o) : -0.5 * q2o * q2o;
                if (v_lane) {
                    A[ve1] = q1o; A[ve2] = q2o; A[ve3] = q3o;
                    A[PS + ve1] = fp1; A[PS + ve2] = fp2; A[PS + ve3] = fp3;
                    A[2 * PS + ve1] = g1; A[2 * PS + ve2] = g2; A[2 * PS + ve3] = g3;
                    A[3 * PS + ve1] = lp1; A[3 * PS + ve2] = lp2; A[3 * PS + ve3] = lp3;
                }
                mail_due = true;                            // (its log-likelihood follows behind T2, then the word)
                newjob = true;
            }
            // else: the bookkeeper has not taken the last entry yet -- the chain loses this pass (the job stays: same results)
            if (newjob) {
                vex3 = exp_d(vq3);
                const double ba = vq1 + vq2 * vex3;
                if (lane_u < DP) sc[BOFF + lane_u] = lane_u < D ? ba : 0.0;
                if (lane_u == LA) sc[0] = ba;
                if (v_lane) { sc[VOFF + ve1] = vq1 - vmu1; sc[VOFF + ve3] = vq3 - vmu3; }
            }
        }
        T8_STAMP(1);
        // ================================================================= T2: the jobs of all chains are in
        if (lane == 0) t8_add(words + T8_W_TB2, 1);
        if (!t8_wait_ge(words + T8_W_TB2, 4 * (pass + 1), w_panic)) { t8_panic(words, s.err, 1, lane); return; }
        const int live_raw = words[T8_W_LIVE];
        // ---- operands of this pass (all requested before the first product)
        double bop[KS], vb[NJ], vt[NJT];
#pragma unroll
        for (int r = 0; r < NRD; ++r) {
            const t8_v2 v = *(t8_v2p)(sl + BOFF + 8 * r + 2 * hi);
            bop[2 * r] = v.x; bop[2 * r + 1] = v.y;
        }
        const double alpha_c = sl[0];
        if (g_on) {
#pragma unroll
            for (int J = 0; J < NJ; ++J) vb[J] = sl[VOFF + 4 * J + hi];
        }
        if (t_on) {
#pragma unroll
            for (int tt = 0; tt < NJT; ++tt) vt[tt] = sl[VOFF + 4 * (4 * tt + bb) + hi];
        }
        if (mail_due) {
            // the log-likelihood of the state just handed over: the four waves' parts (stored behind their T1 arrival, in
            // front of their T2 arrival), in wave order; then the word
            const double l4 = (((0.0 + sc[RESO + 0 * RREC + DP + 1]) + sc[RESO + 1 * RREC + DP + 1]) + sc[RESO + 2 * RREC + DP + 1]) + sc[RESO + 3 * RREC + DP + 1];
            if (lane == 0) A[4 * PS + T8_S_LL] = l4;
            ++sent;
            asm volatile("" ::: "memory");
            *w_mail = sent;
            mail_due = false;
        }
        if (__builtin_amdgcn_readfirstlane(live_raw) <= 0) return;
        T8_STAMP(2);
        // ================================================================= R: the rows (layout 7's pass)
        if (g_on) {
            double acc = 0.0, acc1 = 0.0;
#pragma unroll
            for (int J = 0; J < NJ; J += 2) {
                acc = t8_mfma(om[J], vb[J], acc);
                if (J + 1 < NJ) acc1 = t8_mfma(om[J + 1 < NJ ? J + 1 : J], vb[J + 1 < NJ ? J + 1 : J], acc1);
            }
            sl[OVOFF + 16 * wr + rb] = acc + acc1;
        }
        if (t_on) {
            double acc = 0.0;
#pragma unroll
            for (int tt = 0; tt < NJT; ++tt) acc = t8_mfma(omt[tt], vt[tt], acc);
            acc += dpp_d<0x124>(acc); acc += dpp_d<0x128>(acc);
            if (bb == 0) sl[OVOFF + 16 * NGF + hi] = acc;
        }
        double gacc[KS];
#pragma unroll
        for (int c = 0; c < KS; ++c) gacc[c] = 0.0;
        double dsum = 0.0, lsum = 0.0, wprod = 1.0;
        t8_v2 xf0[NRD], xf1[NRD];
#pragma unroll
        for (int r = 0; r < NRD; ++r) {
            xf0[r] = *reinterpret_cast<const t8_v2p>((uintptr_t)af[r]);
            xf1[r] = *reinterpret_cast<const t8_v2p>((uintptr_t)(af[r] + TILEB));
        }
        t8_v2p pf[NRD], pb[NRD];
#pragma unroll
        for (int r = 0; r < NRD; ++r) { pf[r] = reinterpret_cast<t8_v2p>((uintptr_t)af[r]); pb[r] = reinterpret_cast<t8_v2p>((uintptr_t)ab[r]); }
        auto do_round = [&](const int t, const double y0, const double y1) {
            double f0 = alpha_c, f1 = alpha_c;
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                f0 = t8_mfma(xf0[r].x, bop[2 * r], f0); f1 = t8_mfma(xf1[r].x, bop[2 * r], f1);
                __builtin_amdgcn_sched_barrier(0);
                f0 = t8_mfma(xf0[r].y, bop[2 * r + 1], f0); f1 = t8_mfma(xf1[r].y, bop[2 * r + 1], f1);
                __builtin_amdgcn_sched_barrier(0);
            }
            t8_v2 xb0[NRD], xb1[NRD];
#pragma unroll
            for (int r = 0; r < NRD; ++r) { xb0[r] = pb[r][0]; xb1[r] = pb[r][TILEV]; }
#pragma unroll
            for (int r = 0; r < NRD; ++r) { xf0[r] = pf[r][2 * TILEV]; xf1[r] = pf[r][3 * TILEV]; }
            double l0, l1, w0, w1, g0, g1;
            logistic_pair_lean(f0, f1, y0, y1, l0, l1, w0, w1, g0, g1);
            if (16 * (t + 2) > n) {
                const bool v0 = 16 * t + rb < n, v1 = 16 * (t + 1) + rb < n;
                l0 = v0 ? l0 : 0.0; w0 = v0 ? w0 : 1.0; g0 = v0 ? g0 : 0.0;
                l1 = v1 ? l1 : 0.0; w1 = v1 ? w1 : 1.0; g1 = v1 ? g1 : 0.0;
            }
            lsum += l0; wprod *= w0; dsum += g0;
            lsum += l1; wprod *= w1; dsum += g1;
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                gacc[2 * r] = t8_mfma(xb0[r].x, g0, gacc[2 * r]); gacc[2 * r + 1] = t8_mfma(xb0[r].y, g0, gacc[2 * r + 1]);
            }
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                gacc[2 * r] = t8_mfma(xb1[r].x, g1, gacc[2 * r]); gacc[2 * r + 1] = t8_mfma(xb1[r].y, g1, gacc[2 * r + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        {
            unsigned yb = ybits;
            for (int t = t0; t < t1; t += 2, yb >>= 2) {
                do_round(t, (double)(yb & 1u), (double)((yb >> 1) & 1u));
#pragma unroll
                for (int r = 0; r < NRD; ++r) { pf[r] += 2 * TILEV; pb[r] += 2 * TILEV; }
            }
        }
#pragma unroll
        for (int c = 0; c < KS; ++c) { gacc[c] += dpp_d<0x124>(gacc[c]); gacc[c] += dpp_d<0x128>(gacc[c]); }
        t8_lds *res = sl + RESO + wr * RREC;
        double dz = t8_mfma(1.0, dsum, 0.0);
        dz += dpp_d<0x124>(dz); dz += dpp_d<0x128>(dz);
        if (bb == 0) {
#pragma unroll
            for (int c = 0; c < KS; ++c) res[8 * (c >> 1) + 2 * hi + (c & 1)] = gacc[c];
            if (hi == 0) res[DP] = dz;
        }
        // T1 arrival: the partial sums are out (the LDS performs this wave's operations in order) ...
        asm volatile("" ::: "memory");
        if (lane == 0) t8_add(words + T8_W_TB1, 1);
        // ... and the pass's log-likelihood -- one logarithm of 27 dependent instructions, only the books read it -- behind it
        double lz = t8_mfma(1.0, lsum - log_ge1_d_vc(wprod), 0.0);
        lz += dpp_d<0x124>(lz); lz += dpp_d<0x128>(lz);
        if (bb == 0 && hi == 0) res[DP + 1] = lz;
#ifdef T8_OM_L2
        fetch_om();
#endif
        T8_STAMP(3);
    }
#undef T8_STAMP
}

}  // namespace epx
