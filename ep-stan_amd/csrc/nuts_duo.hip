// k_nuts_duo: the LDS-resident sampler with the work of a chain split over waves by ROLE.
//
// Replaces the Stan subprocess of /root/reference/epstan/method.py:43-118, 349-363 exactly like
// k_nuts (nuts.hip); same algorithm (nuts_state_machine.inc), same arithmetic in the same order,
// hence the same draws bit for bit as k_nuts with one wave per chain (tested).
//
// Why: with one wave per chain (k_nuts, layout 1) that wave carries the 23 state vectors of the
// tree bookkeeping AND the 64 + 64 row / accumulator registers of the gradient: 256 VGPRs + 247
// AGPRs of spill at D = 32, and every part of a leapfrog runs one after the other (measured at
// the C3 site size: 23 800 cycles per leapfrog, 36 % row loop, 31 % cavity mat-vec, 14 % tree
// bookkeeping, 10 % chain rule).  Here a chain is
//   * RW row waves (R): wait for (alpha, beta) -> fused row pass over their rows of the LDS-resident
//     X -> transposing butterfly -> publish (X'g, sum g, log-lik);
//   * one state wave (S): owns the NUTS state.  Per leapfrog it takes the half kick + drift,
//     publishes (alpha, beta) and, WHILE the row waves sweep the rows, computes the cavity term
//     Omega (phi - mu) and runs the tree bookkeeping of the PREVIOUS leapfrog's state; then the
//     chain rule and the second half kick.
// The trajectory is integrated speculatively one leapfrog ahead of the bookkeeping (as in
// k_nuts_spec): when the bookkeeping decides to continue elsewhere (other tree end, new
// transition, step-size trial, new metric) the job in flight is dropped -- one wasted gradient per
// change of direction.  Accepted states are those of the sequential algorithm.
//
// Hand-offs go through LDS slots with sequence numbers (ds_write data, s_waitcnt, ds_write flag /
// poll): the chains of a workgroup are NOT coupled by a barrier, a finished chain's waves leave.
// Every spin is bounded; a spin that gives up raises NutsArgs::err and ends the chain (the host
// reports it) instead of hanging the GPU.
#include "nuts_common.h"
#include "epx_pieces.h"

#include <type_traits>

namespace epx {

// s_setprio levels (A/B, scripts/ab_duo.py): with the critical-path shortcut the state wave is the longer of the
// two, so the row wave must NOT outrank it (row wave 2 / state wave 0: 1 390 ms; all equal: 1 175 ms per launch)
// Tuning constants.  (The A/B switches of rounds 2-4 whose losing side was measured are gone: polled hand-offs of the row
// team, EPX_T7_ROWPREF / _LLAFTER / _EARLYT / _LATELL, EPX_ROW_IMM, EPX_PUBLISH_NOWAIT, EPX_L6_LLAFTER, EPX_PIECE_INLINE --
// the measurements are in HISTORY.md.  What stays switchable: EPX_PIECE_FENCE, EPX_STAMPS, and EPX_YIELD at run time.)
#define EPX_DUO_SLEEP 1          // s_sleep between two looks at a polled word (layouts 5 / 6)
#define EPX_DUO_SLEEP_BKW 0      // layout 6 (one chain per workgroup): the waves that wait on the critical chain look again at once
#define EPX_PRIO_R 0
#define EPX_PRIO_S_BG 0
#define EPX_PRIO_S_CRIT 2
// Hand-off words and slots live in LDS and are reached through address_space(3) pointers: through a generic pointer
// the compiler emits flat_load / flat_store, which are counted in vmcnt AND lgkmcnt -- a wait for a polled flag or for
// a result then also waits for every global store in flight (the tree stack's, ~1-2 us each)
typedef double lds_v2f64 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) lds_v2f64 *lds_v2f64_p;
typedef volatile __attribute__((address_space(3))) int duo_flag_t;
typedef __attribute__((address_space(3))) double duo_lds_f64;
__device__ inline duo_flag_t *duo_flags_at(const void *generic) { return reinterpret_cast<duo_flag_t *>((uintptr_t)(unsigned)(size_t)generic); }
__device__ inline duo_lds_f64 *duo_lds_at(const void *generic) { return reinterpret_cast<duo_lds_f64 *>((uintptr_t)(unsigned)(size_t)generic); }
enum { DUO_EXIT = -7, DUO_TIMEOUT = -99, DUO_SPIN_LIMIT = 1 << 23, DUO_NO_MORE = 1 << 30 };
enum { DUO_RESTART = 1, DUO_LEAVE = 2 };

__device__ inline int duo_wait(duo_flag_t *flag, int want) {
    for (int spin = 0; spin < DUO_SPIN_LIMIT; ++spin) {
        const int v = __builtin_amdgcn_readfirstlane(*flag);
        if (v == want || v == DUO_EXIT) { asm volatile("" ::: "memory"); return v; }
        __builtin_amdgcn_s_sleep(EPX_DUO_SLEEP);
    }
    return DUO_TIMEOUT;
}
// for counters that only grow (acknowledgements, generations)
__device__ inline int duo_wait_ge(duo_flag_t *flag, int want) {
    for (int spin = 0; spin < DUO_SPIN_LIMIT; ++spin) {
        const int v = __builtin_amdgcn_readfirstlane(*flag);
        if (v >= want || v == DUO_EXIT) { asm volatile("" ::: "memory"); return v; }
        __builtin_amdgcn_s_sleep(EPX_DUO_SLEEP);
    }
    return DUO_TIMEOUT;
}
// Rows the TEAM form keeps in LDS for a site of n rows: every row wave takes the same EVEN number of 16-row tiles (its
// rounds are pairs of tiles at compile-time distances), the tiles behind the site's rows hold zeros
__host__ __device__ inline int team_tiles_per_wave(int n) { const int t = ((n + 15) / 16 + 3) / 4; return (t + 1) & ~1; }
__host__ __device__ inline int team_rows(int n) { return 4 * team_tiles_per_wave(n) * 16; }
// Lanes of ONE wave exchange values through LDS: for the compiler that is a data race unless the stores are released
// and the loads acquire (without it a load behind `if (lane writes) store` is taken to return what an earlier load of
// the address returned in the lanes that did not store).  No instruction beyond the wait for the stores.
__device__ inline void wave_lds_exchange() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Between the steps of the state wave's relay (one wave writes a scratch line in one lane order and reads it back in
// another): -DEPX_RELAY_NOWAIT (A/B) keeps only the compiler from reordering -- the LDS executes one wave's accesses in
// issue order -- instead of draining the wave's LDS operations at every step.
__device__ inline void relay_step() {
#ifdef EPX_RELAY_NOWAIT
    asm volatile("" ::: "memory");
#else
    wave_lds_exchange();
#endif
}
// The TEAM form's hand-off: every wave of the workgroup is in lock step with the passes anyway, so "the jobs are in"
// and "the results are in" are the two s_barriers of a pass (LDS traffic drained first; vector-memory operations stay
// in flight).  A waiting wave is parked by the hardware: no polls on the SIMD its partner computes on, no wake-up latency.
//
// BARRIER PARITY (the invariant every exit of duo_piece keeps; there is no time-out on an s_barrier, a slip is a hang):
//   * a pass is exactly two barriers for EVERY wave of the workgroup: B1 "jobs in" (odd), B2 "results in" (even);
//   * the live-chain word f_live is read by every wave right behind B1 and nowhere else, and changed only between a
//     B2 and the next B1 (team_leave's decrement comes before its first barrier, which is a B1);
//   * a ROW wave leaves only behind a B1 at which it read f_live == 0 -- the last barrier of the piece;
//   * a STATE wave never returns with a barrier outstanding: every `return` below that a state wave can reach is
//     preceded by team_leave(), which keeps arriving at B1/B2 pairs until it, too, reads f_live == 0 behind a B1.
//     EPX_CHAIN_EXIT inside the state machine only breaks out of the pass loop, to the team_leave of the epilogue.
//   Each `return` below carries a [parity] note saying which of these applies.  tests/test_gpu_round4.py's litmus and the
//   chains < 4 / failed-chain cases of tests/test_gpu_round3.py run every one of them.
__device__ inline void team_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ inline double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ inline void ck_assign(GScal &x, double v) { x = v; }
__device__ inline void ck_assign(RScal &x, double v) { x = v; }
// everything this wave wrote to LDS is visible before the flag that follows
__device__ inline void duo_publish(duo_flag_t *flag, int v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *flag = v;
}
// The same without the wait, for the hand-offs on the critical chain of the one-chain-per-workgroup form (layout 6): the LDS
// performs the operations of ONE wave in issue order (which is what lets `lgkmcnt(N)` count them), so the word's store is
// performed behind the data's stores of the same wave, and a reader's data reads are issued behind its read of the word.
template <bool NOWAIT>
__device__ inline void duo_publish_c(duo_flag_t *flag, int v) {
    if constexpr (NOWAIT) { asm volatile("" ::: "memory"); *flag = v; }
    else duo_publish(flag, v);
}

// In-kernel cycle stamps exist only in the diagnostic build (-DEPX_STAMPS); its run time is never
// quoted, only the shares.  Slots per workgroup (chain 0): state wave 0 prep / 1 bookkeeping /
// 2 cavity term / 3 waiting for the row waves / 4 chain rule; row wave 5 waiting for a job / 6 row pass;
// 7 = leapfrogs
#ifdef EPX_STAMPS
#define STAMP(i)                                                                   \
    do {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                        \
        tacc[i] += t_ - tprev; tprev = t_;                                         \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)
#define STAMP_INIT unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#define TSTAMP(i)                                                                  \
    do {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                        \
        tdet[i] += t_ - tprev2; tprev2 = t_;                                       \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)
#define TSTAMP_INIT unsigned long long tdet[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tprev2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define STAMP(i) do { } while (0)
#define STAMP_INIT do { } while (0)
#define TSTAMP(i) do { } while (0)
#define TSTAMP_INIT do { } while (0)
#endif

typedef const __attribute__((address_space(4))) NutsArgs DuoArgsK;       // the kernel arguments where they are: kernarg segment

// One piece of a chain's run: transitions [t_begin, t_end) of one site by the waves of one workgroup (the whole run
// in a plain launch).  PIECED is a template parameter because a piece loop around this body INLINED costs it its register
// allocation (76 -> 500 B of scratch per lane, 15 % of the time): the plain launch keeps the kernel without the loop, and
// the pieced launch's loop calls the body as a function (k_nuts_duo_loop / duo_piece_call below).

template <int NV, int DP, int CPB, int RW, bool STL, bool COLD, bool PIECED>
__device__ __forceinline__ void duo_piece(DuoArgsK *kargs_p, int tid, bool queued, int q_site, int q_t0) {
    extern __shared__ __align__(16) unsigned char smem[];
    DuoArgsK &a_piece = *kargs_p;
#define a a_piece
    using V = Vec<NV>;
    constexpr int LOG = Log2<DP>::v;
    constexpr int SPR = DP / 2;                       // 16-B slots per row
    constexpr int RPL = DP >= 32 ? 1 : 32 / DP;       // rows per 256-B bank line
    constexpr int SREC = nuts_stack_record(NV);       // per-level stack record (doubles)
    constexpr int RES = DP + 2;                       // result of a row wave: X'g (DP), sum g, log-lik
    constexpr int JOB = 0;                            // job (alpha, beta): the head of the chain's slot ...
    // One chain per workgroup (RW > 1): [job RES | v = phi - mu (NV x 64) | per row wave: X'g, sum g, log-lik (RES) |
    // Omega v (NV x 64)] -- the cavity term is taken off the state wave, the longest role of this form, by a wave of
    // its own (O) that works beside the row waves
    // (VN: doubles of the v and Omega v lines -- the TEAM form keeps only the d <= 2 DP + 2 live ones, in whole 16-byte
    // pairs, and gives the LDS it saves to one more level of the tree stack)
    constexpr int VN = (CPB == 4 && RW == 4 && 2 * DP + 8 < NV * 64) ? 2 * DP + 8 : NV * 64;
    constexpr int VOFF = RES, RREC = RES, RESO = RES + VN, OVOFF = RESO + RW * RREC;
    // One chain per workgroup (CPB == 1): the tree bookkeeping gets a wave of its own (BK), as in k_nuts_spec --
    // the state wave integrates on speculatively and hands every finished state over through a two-entry mailbox;
    // BK answers with a control record only when the trajectory continues elsewhere (other tree end, new
    // transition, step-size trial): generation-numbered, states of an old generation are dropped.
    constexpr bool BKW = CPB == 1;
    // Row TEAM (CPB == 4, RW == 4; layout 7): the four row waves are not a chain's own -- they serve the four chains of
    // the site together, in lock step, on the matrix pipe.  Pass p: every chain has posted job p; wave w takes a quarter
    // of the site's 16-row tiles through  F = alpha + X B  (v_mfma_f64_4x4x4: 16 rows x 4 chains per instruction), the
    // logistic terms on the product's own lanes (one (row, chain) per lane), and  G += X' g  with g as it stands in
    // those lanes (the product's D layout IS the B layout of the transposed product); it also takes a 16-row group of
    // the cavity term  Omega V  (Omega in registers as A operands, V = phi - mu of the four chains published with the
    // jobs).  The rows are read from LDS twice per pass for FOUR gradients (eight times in the one-wave-per-chain form),
    // by instructions that cost one issue slot per 16 x 4 x 4 multiply-adds.  The state waves are the ones of the
    // RW > 1 form: they sum the four waves' partial results in wave order.
    constexpr bool TEAM = CPB == 4 && RW == 4;
    constexpr int BOFF = TEAM ? 2 : 1;                // beta behind alpha in the job (TEAM: 16-byte aligned pairs)
    constexpr bool TBAR = TEAM;                       // the TEAM form's hand-offs are workgroup barriers (see team_barrier)
    constexpr int MREC = 4 * NV * 64 + 4;             // mailbox entry: q, p, grad, per-element log-density terms; ll, -, generation, -
    constexpr int CREC = 4 * NV * 64 + 4;             // control record: q, p, grad, metric, eps_l, command
    constexpr int NFLAG = TEAM ? 1 : 1 + RW + (BKW ? 4 : 0);     // per chain: job, results, (mail, acknowledged, control generation, cavity term); TEAM: the job word, the team's four words behind the chains'


    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves 0..CPB-1: state wave of chain c; then the row waves.  A workgroup's waves go to the SIMDs
    // round robin, so wave w and w + 4 share one: the row waves of chain c sit CPB waves behind the
    // state wave of chain c + 1 -- a chain that is the last one running keeps its two roles on different SIMDs
    // A piece that ends before the chain's last transition (pieced launch) leaves a checkpoint record; the piece that
    // continues it restores, re-evaluates the gradient at the current sample (the same arithmetic on the same position:
    // the same bits) and goes on.  Same draws as the plain launch.
    // (one chain per workgroup: wave 0 bookkeeping, 1 state, 2 .. 1 + RW rows, 2 + RW cavity term -- waves w and w + 4
    // share a SIMD: the bookkeeping and the cavity-term wave; measured against state + cavity term on one: 374 vs 382 ms
    // per C2 iteration)
    const bool is_state = BKW ? wave == 1 : wave < CPB;
    const bool is_bk = BKW && wave == 0;
    const bool is_om = BKW && wave == 2 + RW;
    const int team = BKW ? 0 : (is_state ? wave : (TEAM ? 0 : ((wave - CPB) / RW + CPB - 1) % CPB));
    const int wr = BKW ? (wave >= 2 && wave < 2 + RW ? wave - 2 : 0) : (is_state ? 0 : (TEAM ? wave - CPB : (wave - CPB) % RW));
    const int bps = (a.chains + CPB - 1) / CPB;
    const bool segmented = queued;
    const int sb = queued ? q_site : (a.order ? a.order[blockIdx.x / bps] : (int)(blockIdx.x / bps));
    const int cb = queued ? 0 : blockIdx.x % bps;
    const int t_begin = queued ? q_t0 : 0;
    const int q_len = queued ? piece_len_at(a, q_site, q_t0) : 0;
    const int t_end = queued ? (q_t0 + q_len < a.iter ? q_t0 + q_len : a.iter) : a.iter;
    const bool resume = t_begin > 0;
    const int k = a.k0 + sb;
    const int chain = cb * CPB + team;
    const int D = a.D, d = a.d, P = a.P, model = a.model;
    const int64_t row0 = a.k_lim[k];
    const int n = (int)(a.k_lim[k + 1] - row0);
    constexpr int OUP = NV > 1 ? 4 : 8;               // the cavity precision is zero padded to whole groups of OUP column pairs
    constexpr int OU = OUP;                           // column pairs per round of the mat-vec
    const int dm = d < 64 ? d : 64;                   // cavity precision: rows / columns held pair-interleaved
    const int tr = d - dm;                            // ... and the rows beyond (0..2 for D <= 32)
    const int npair = (dm + 1) / 2, npad = (npair + OUP - 1) / OUP * OUP;    // pairs, zero padded to whole groups
    const int tstride = 2 * npad + 2;                 // tail rows: [2 rows (zero when absent)][column], NV > 1 only

    double *Xs = reinterpret_cast<double *>(smem);
    double *Oms = reinterpret_cast<double *>(smem + a.off_Om);        // [(pair p, row e)] -> (Om[e][2p], Om[e][2p+1])
    double *Ots = reinterpret_cast<double *>(smem + a.off_tail);      // [row r - 64][column], stride tstride
    duo_lds_f64 *slot = duo_lds_at(smem + a.off_slot) + team * a.slot_doubles;
    duo_flag_t *flags = duo_flags_at(smem + a.off_flag) + team * NFLAG;
    duo_flag_t *f_job = flags, *f_res = TEAM ? duo_flags_at(smem + a.off_flag) + CPB : flags + 1;
    duo_flag_t *f_mail = flags + 1 + RW, *f_ack = f_mail + 1, *f_ctl = f_mail + 2, *f_ov = f_mail + 3;
    // TBAR: the chains of the workgroup that are still running.  A state wave whose chain is done (or does not exist)
    // takes the word down and then keeps the others' barriers company until it reads 0 -- right behind "the jobs are
    // in", where every wave looks at it (a wave that simply left, or waited at a later barrier, would be counted as
    // arrived at the others' next one)
    duo_flag_t *f_live = duo_flags_at(smem + a.off_flag) + 2 * CPB;
    auto team_leave = [&](bool was_live) {
        if constexpr (TBAR) {
            if (was_live && lane == 0)
                __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) int *>(const_cast<__attribute__((address_space(3))) int *>(f_live)), -1,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (;;) {
                team_barrier();
                if (__builtin_amdgcn_readfirstlane(*f_live) == 0) break;
                team_barrier();
            }
        }
    };
    double *mbox = reinterpret_cast<double *>(smem + a.off_spec);     // BKW: 2 x MREC, then 2 x CREC
    double *ctrl = mbox + 2 * MREC;
    (void)f_mail; (void)f_ack; (void)f_ctl; (void)f_ov; (void)mbox; (void)ctrl;

    // ---- stage the site: rows HBM -> LDS once per site update (as k_nuts), cavity precision re-laid
    {
        const double *Xg = a.X + (size_t)row0 * D;
        const int nslot = (TEAM ? team_rows(n) : n) * SPR;       // (TEAM: an even number of whole 16-row tiles per row wave, zero rows behind the site's)
        for (int s = tid; s < nslot; s += blockDim.x) {
            const int r = s / SPR, jp = s % SPR, c0 = 2 * jp;
            double2 v;
            // (read once per piece: non-temporal, so that the 128 KB of a site do not push the workgroups' tree stacks and
            // cold stores out of the XCD's L2 at every piece start)
            if (r >= n) { v.x = 0.0; v.y = 0.0; }
            else if ((D & 1) == 0 && c0 + 1 < D) {
                typedef double nt_v2 __attribute__((ext_vector_type(2)));
                const nt_v2 t2 = __builtin_nontemporal_load(reinterpret_cast<const nt_v2 *>(Xg + (size_t)r * D + c0));
                v.x = t2.x; v.y = t2.y;
            } else {
                v.x = c0 < D ? Xg[(size_t)r * D + c0] : 0.0;
                v.y = c0 + 1 < D ? Xg[(size_t)r * D + c0 + 1] : 0.0;
            }
            const int sw = (r / RPL) & (SPR - 1);
            *reinterpret_cast<double2 *>(Xs + (size_t)r * DP + 2 * (jp ^ sw)) = v;
        }
        const double *Om_g = a.cav_Om + (size_t)k * d * d;                // column-major, symmetric
        if constexpr (!TEAM) {
            for (int idx = tid; idx < npad * dm; idx += blockDim.x) {
                const int p = idx / dm, e = idx % dm;
                double2 v;
                v.x = 2 * p < dm ? Om_g[(size_t)(2 * p) * d + e] : 0.0;
                v.y = 2 * p + 1 < dm ? Om_g[(size_t)(2 * p + 1) * d + e] : 0.0;
                *reinterpret_cast<double2 *>(Oms + 2 * (size_t)idx) = v;
            }
            if constexpr (NV > 1) {
                for (int idx = tid; idx < 2 * tstride; idx += blockDim.x) {
                    const int r = idx / tstride, j = idx % tstride;
                    Ots[idx] = (r < tr && j < d) ? Om_g[(size_t)j * d + dm + r] : 0.0;
                }
            }
        } else {
            // (the cavity precision lives in the row waves' registers; jobs and results start as zeros: the column of a
            // chain that does not exist stays finite)
            double *s0 = reinterpret_cast<double *>(smem + a.off_slot);
            for (int idx = tid; idx < CPB * a.slot_doubles; idx += blockDim.x) s0[idx] = 0.0;
            double *s1 = reinterpret_cast<double *>(smem + a.off_scr);
            for (int idx = tid; idx < CPB * a.scr_doubles; idx += blockDim.x) s1[idx] = 0.0;
        }
        if (tid < (TEAM ? 2 * CPB + 1 : CPB * NFLAG))
            reinterpret_cast<volatile int *>(smem + a.off_flag)[tid] =
                (TEAM && tid == 2 * CPB) ? (a.chains - cb * CPB < CPB ? a.chains - cb * CPB : CPB) : 0;
        if constexpr (BKW) { for (int idx = tid; idx < 2 * MREC; idx += blockDim.x) mbox[idx] = 0.0; }     // (entries beyond P stay 0)
    }
    __syncthreads();                                   // the only workgroup barrier of a piece
    if (chain >= a.chains) { team_leave(false); return; }      // [parity] a state wave without a chain: was never counted in f_live, keeps the barriers company

    if constexpr (TEAM) {
        if (!is_state) {
            // ================================================================= row team (see TEAM above)
            // v_mfma_f64_4x4x4f64 operand layout (measured, scripts/probe/mfma_layout.hip): A[b][i][k] in lane 16 k + 4 b + i,
            // B[b][k][j] in lane 16 k + 4 b + j, D[b][i][j] in lane 16 i + 4 b + j.  With lane = (hi, bb, lo):
            //   forward   A = X[row 4 bb + lo of the tile][column 8 r + 2 hi (+1)],   B = beta of chain lo at those columns,
            //             D = f[row 4 bb + hi][chain lo]  -- and g = y - sigmoid(f) in the same lanes is the
            //   backward  B = g[row 4 bb + hi][chain lo],   A = X[row 4 bb + hi][column 8 r + 2 lo (+1)],
            //             D = (X' g)[column 8 r + 2 hi (+1)][chain lo], one partial sum per row block bb.
            // One ds_read_b128 feeds two MFMAs (the two columns of a 16-byte slot are two k-steps / two column groups);
            // the row images' XOR swizzle (slot ^ row) keeps both read patterns free of bank conflicts.
            constexpr int KS = DP / 4, NRD = DP / 8, ROWB = DP * 8, TILEB = 16 * ROWB;
            constexpr int DMAX = 2 * DP + 2;               // cavity term: d <= 2 D + 2 rows
            constexpr int NJ = (DMAX + 3) / 4;             // its k-steps
            constexpr int NGF = DMAX / 16;                 // 16-row groups, one per row wave; rows 16 NGF .. + 3 (wave 3) by a k-split
            constexpr int NJT = (NJ + 3) / 4;
            static_assert(NGF <= 4, "one 16-row group of the cavity term per row wave");
            const int lo = lane & 3, bb = (lane >> 2) & 3, hi = lane >> 4;
            const int tpw = team_tiles_per_wave(n);       // (even; tiles beyond the site's rows are zeros and masked)
            const int t0 = wr * tpw, t1 = t0 + tpw;
            const int sdb = a.slot_doubles;
            // cavity precision as A operands: group wr (rows 16 wr + 4 bb + lo, columns 4 J + hi); wave 3 also the rows
            // beyond the groups, block bb taking the k-steps J = bb, bb + 4, ...
            double om[NJ], omt[NJT];
            {
                const double *Om_g = a.cav_Om + (size_t)k * d * d;
                const int e = 16 * wr + 4 * bb + lo;
#pragma unroll
                for (int J = 0; J < NJ; ++J) {
                    const int c = 4 * J + hi;
                    om[J] = (wr < NGF && e < d && c < d) ? Om_g[(size_t)c * d + e] : 0.0;
                }
                const int et = 16 * NGF + lo;
#pragma unroll
                for (int tt = 0; tt < NJT; ++tt) {
                    const int c = 4 * (4 * tt + bb) + hi;
                    omt[tt] = (wr == 3 && et < d && c < d) ? Om_g[(size_t)c * d + et] : 0.0;
                }
            }
            const bool g_on = wr < NGF && 16 * wr < d, t_on = wr == 3 && d > 16 * NGF;
            const int rf = lane & 15, rb = 4 * bb + hi;    // row within a tile: forward operand; backward operand and products
            unsigned ybits = 0;                            // responses of this lane's product rows, one bit per tile
            for (int t = t0; t < t1; ++t) {
                const int r = 16 * t + rb;
                if (r < n && a.y[row0 + r]) ybits |= 1u << (t - t0);
            }
            const unsigned xbase = (unsigned)(size_t)Xs;
            const unsigned swf = (unsigned)((rf / RPL) & (SPR - 1)), swb = (unsigned)((rb / RPL) & (SPR - 1));
            unsigned af[NRD], ab[NRD];
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                // (of the wave's first tile pair; a round's other reads sit at compile-time distances)
                af[r] = xbase + (unsigned)t0 * TILEB + (unsigned)rf * ROWB + ((((unsigned)(4 * r + hi)) ^ swf) << 4);
                ab[r] = xbase + (unsigned)t0 * TILEB + (unsigned)rb * ROWB + ((((unsigned)(4 * r + lo)) ^ swb) << 4);
            }
            duo_lds_f64 *const sl = slot + lo * sdb;   // the slot of chain lo (team == 0 here: `slot` is chain 0's)
            STAMP_INIT;
            TSTAMP_INIT;
            __builtin_amdgcn_s_setprio(EPX_PRIO_R);
            for (int pass = 1;; ++pass) {
#ifdef EPX_STAMPS
                const unsigned long long tw0_ = __builtin_amdgcn_s_memtime();
#endif
                // The look at the live word behind "the jobs are in" used to be an LDS round trip of its own IN FRONT of the
                // operands' reads (read, wait, branch, then the reads): the word is requested here and looked at behind the
                // cavity term, whose operands' reads go out right behind it.  A pass that finds no chain left has then
                // multiplied stale operands into a slot nobody reads.  (Requesting the operands in front of the look -- as
                // volatile reads, or pinned by empty statements -- was measured slower, -1.8 % and -2.5 %: the wait for the
                // look then covers every operand and the rows' reads start behind it.)
                int live, live_raw = 1;
                double bop[KS];
                double alpha_c = 0.0;
                double vb[NJ];
                team_barrier(); live_raw = *f_live; live = 1;
#ifdef EPX_STAMPS
                {   // histogram of this wait: bins of 512 cycles, the last one open (fourth record of the stamps)
                    const unsigned long long dw_ = __builtin_amdgcn_s_memtime() - tw0_;
                    int bin_ = (int)(dw_ >> 9); bin_ = bin_ > 15 ? 15 : bin_;
                    if (a.stamps && wr == 0 && lane == 0 && pass > 1) atomicAdd(&a.stamps[((size_t)3 * a.stamps_nrec) * 8 + bin_], 1ull);
                }
#endif
                STAMP(5);
                TSTAMP(0);
                if (live <= 0) {
                    if (live < 0) { if (lane == 0) atomicOr(a.err, 1); f_res[wr] = DUO_EXIT; }
#ifdef EPX_STAMPS
                    if (a.stamps && wr == 0 && lane == 0) {
                        a.stamps[(size_t)blockIdx.x * 8 + 5] += tacc[5]; a.stamps[(size_t)blockIdx.x * 8 + 6] += tacc[6];
                        for (int i = 0; i < 7; ++i) a.stamps[((size_t)a.stamps_nrec + blockIdx.x) * 8 + i] += tdet[i];
                        a.stamps[((size_t)a.stamps_nrec + blockIdx.x) * 8 + 7] += (unsigned long long)(pass - 1);
                    }
                    // (third record: every row wave's waiting and working cycles -- who the team waits for)
                    if (a.stamps && lane == 0) {
                        a.stamps[((size_t)2 * a.stamps_nrec + blockIdx.x) * 8 + wr] += tacc[5];
                        a.stamps[((size_t)2 * a.stamps_nrec + blockIdx.x) * 8 + 4 + wr] += tacc[6];
                    }
#endif
                    if (a.team_passes && wr == 0 && lane == 0) atomicAdd(a.team_passes + k, (double)(pass - 1));
                    return;                            // [parity] row wave, behind a B1 with f_live == 0: no wave waits at a barrier again
                }
                // ---- operands of this pass
#pragma unroll
                for (int r = 0; r < NRD; ++r) {
                    const lds_v2f64 v = *(lds_v2f64_p)(sl + JOB + BOFF + 8 * r + 2 * hi);
                    bop[2 * r] = v.x; bop[2 * r + 1] = v.y;
                }
                alpha_c = sl[JOB];
                // ---- cavity term Omega V of the four chains
                if (g_on) {
                    // (all B operands requested first: a product behind its own LDS round trip would pay the latency 17 times;
                    // k-steps beyond d meet zero A operands)
#pragma unroll
                    for (int J = 0; J < NJ; ++J) vb[J] = sl[VOFF + 4 * J + hi];
                    double acc = 0.0, acc1 = 0.0;             // (two chains: a dependent product issues 4 cycles later than a free one)
#pragma unroll
                    for (int J = 0; J < NJ; J += 2) {
                        acc = mfma4(om[J], vb[J], acc);
                        if (J + 1 < NJ) acc1 = mfma4(om[J + 1 < NJ ? J + 1 : J], vb[J + 1 < NJ ? J + 1 : J], acc1);
                    }
                    sl[OVOFF + 16 * wr + rb] = acc + acc1;
                }
                if (t_on) {
                    double vt[NJT];
#pragma unroll
                    for (int tt = 0; tt < NJT; ++tt) vt[tt] = sl[VOFF + 4 * (4 * tt + bb) + hi];
                    double acc = 0.0;
#pragma unroll
                    for (int tt = 0; tt < NJT; ++tt) acc = mfma4(omt[tt], vt[tt], acc);
                    acc += dpp_d<0x124>(acc); acc += dpp_d<0x128>(acc);          // the four blocks' k-shares (row_ror 4, 8)
                    if (bb == 0) sl[OVOFF + 16 * NGF + hi] = acc;
                }
                TSTAMP(1);
                {
                    if (__builtin_amdgcn_readfirstlane(live_raw) <= 0) {
#ifdef EPX_STAMPS
                        if (a.stamps && wr == 0 && lane == 0) {
                            a.stamps[(size_t)blockIdx.x * 8 + 5] += tacc[5]; a.stamps[(size_t)blockIdx.x * 8 + 6] += tacc[6];
                            for (int i = 0; i < 7; ++i) a.stamps[((size_t)a.stamps_nrec + blockIdx.x) * 8 + i] += tdet[i];
                            a.stamps[((size_t)a.stamps_nrec + blockIdx.x) * 8 + 7] += (unsigned long long)(pass - 1);
                        }
                        if (a.stamps && lane == 0) {
                            a.stamps[((size_t)2 * a.stamps_nrec + blockIdx.x) * 8 + wr] += tacc[5];
                            a.stamps[((size_t)2 * a.stamps_nrec + blockIdx.x) * 8 + 4 + wr] += tacc[6];
                        }
#endif
                        if (a.team_passes && wr == 0 && lane == 0) atomicAdd(a.team_passes + k, (double)(pass - 1));
                        return;                        // [parity] row wave, behind a B1 with f_live == 0 (the look deferred behind the cavity term)
                    }
                }
                // ---- the rows: two tiles per round (their logistic terms overlap).  The LDS reads run ahead of their
                // use: the backward operands of a round and the forward operands of the NEXT round are requested
                // before the round's logistic terms, so no product waits for an LDS round trip.
                double gacc[KS];
#pragma unroll
                for (int c = 0; c < KS; ++c) gacc[c] = 0.0;
                double dsum = 0.0, lsum = 0.0, wprod = 1.0;
                lds_v2f64 xf0[NRD], xf1[NRD];
#pragma unroll
                for (int r = 0; r < NRD; ++r) {
                    xf0[r] = *reinterpret_cast<const lds_v2f64_p>((uintptr_t)af[r]);
                    xf1[r] = *reinterpret_cast<const lds_v2f64_p>((uintptr_t)(af[r] + TILEB));
                }
                // one round: tiles t, t + 1 of the site = byte offset `off` from the wave's first tile pair
                // (running LDS POINTERS, bumped once per round; the reads of a round sit at compile-time distances from them.
                // Integer addresses converted at every use cost an extra vector add per read)
                lds_v2f64_p pf[NRD], pb[NRD];
#pragma unroll
                for (int r = 0; r < NRD; ++r) { pf[r] = reinterpret_cast<lds_v2f64_p>((uintptr_t)af[r]); pb[r] = reinterpret_cast<lds_v2f64_p>((uintptr_t)ab[r]); }
                constexpr int TILEV = TILEB / 16;              // a tile in 16-byte units
                auto do_round = [&](const int t, const double y0, const double y1) {       // (y0, y1: 1/2 - y of the lane's row in either tile)
                    double f0 = alpha_c, f1 = alpha_c;
#pragma unroll
                    for (int r = 0; r < NRD; ++r) {
                        // (the two tiles' chains alternate: a product that waits for its own accumulator issues 4 cycles late)
                        f0 = mfma4(xf0[r].x, bop[2 * r], f0); f1 = mfma4(xf1[r].x, bop[2 * r], f1);
                        __builtin_amdgcn_sched_barrier(0);
                        f0 = mfma4(xf0[r].y, bop[2 * r + 1], f0); f1 = mfma4(xf1[r].y, bop[2 * r + 1], f1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    TSTAMP(2);
                    lds_v2f64 xb0[NRD], xb1[NRD];
#pragma unroll
                    for (int r = 0; r < NRD; ++r) {
                        xb0[r] = pb[r][0];
                        xb1[r] = pb[r][TILEV];
                    }
                    // (the round after the last one reads what lies behind the wave's tiles: values unused)
#pragma unroll
                    for (int r = 0; r < NRD; ++r) {
                        xf0[r] = pf[r][2 * TILEV];
                        xf1[r] = pf[r][3 * TILEV];
                    }
                    TSTAMP(3);
                    double l0, l1, w0, w1, g0, g1;
                    logistic_pair_lean(f0, f1, y0, y1, l0, l1, w0, w1, g0, g1);
                    if (16 * (t + 2) > n) {                // the site's last tile(s): rows beyond n add nothing
                        const bool v0 = 16 * t + rb < n, v1 = 16 * (t + 1) + rb < n;
                        l0 = v0 ? l0 : 0.0; w0 = v0 ? w0 : 1.0; g0 = v0 ? g0 : 0.0;
                        l1 = v1 ? l1 : 0.0; w1 = v1 ? w1 : 1.0; g1 = v1 ? g1 : 0.0;
                    }
                    lsum += l0; wprod *= w0; dsum += g0;
                    lsum += l1; wprod *= w1; dsum += g1;
                    TSTAMP(4);
#pragma unroll
                    for (int r = 0; r < NRD; ++r) {
                        gacc[2 * r] = mfma4(xb0[r].x, g0, gacc[2 * r]); gacc[2 * r + 1] = mfma4(xb0[r].y, g0, gacc[2 * r + 1]);
                    }
#pragma unroll
                    for (int r = 0; r < NRD; ++r) {
                        gacc[2 * r] = mfma4(xb1[r].x, g1, gacc[2 * r]); gacc[2 * r + 1] = mfma4(xb1[r].y, g1, gacc[2 * r + 1]);
                    }
                    TSTAMP(5);
                    __builtin_amdgcn_sched_barrier(0);      // (nothing moves between the rounds: the unrolled form must not stretch live ranges over them)
                };
                {
                    // (the rounds are NOT unrolled to compile-time distances: two or four rounds in one body cost the wave 8 / 44
                    // spilled registers -- the cavity operands, reloaded from scratch at the top of every pass)
                    unsigned yb = ybits;
                    for (int t = t0; t < t1; t += 2, yb >>= 2) {
                        // 1/2 - y as a double, from the row's bit: sign = the bit, the rest 0x3FE00000'00000000
                        do_round(t, __hiloint2double((int)((yb << 31) | 0x3FE00000u), 0),
                                 __hiloint2double((int)(((yb << 30) & 0x80000000u) | 0x3FE00000u), 0));
#pragma unroll
                        for (int r = 0; r < NRD; ++r) { pf[r] += 2 * TILEV; pb[r] += 2 * TILEV; }
                    }
                }
                // ---- sums over the row blocks (lanes ^ 4, ^ 8), then over the rows hi of a block for the two scalars
                // (a product with ones: D[.][j] = sum over k of B[k][j])
#pragma unroll
                for (int c = 0; c < KS; ++c) { gacc[c] += dpp_d<0x124>(gacc[c]); gacc[c] += dpp_d<0x128>(gacc[c]); }
                duo_lds_f64 *res = sl + RESO + wr * RREC;
                {
                    // The next job needs X'g and sum g; the log-likelihood -- a logarithm of 27 dependent instructions, a
                    // product with ones and two exchanges behind it -- is for the books only: it is formed behind "the
                    // results are in", in the stretch this wave waits through anyway, and read by the state wave behind the
                    // next job's barrier (which this wave passes only after the store).
                    double dz = mfma4(1.0, dsum, 0.0);
                    dz += dpp_d<0x124>(dz); dz += dpp_d<0x128>(dz);
                    if (bb == 0) {
#pragma unroll
                        for (int c = 0; c < KS; ++c) res[8 * (c >> 1) + 2 * hi + (c & 1)] = gacc[c];
                        if (hi == 0) res[DP] = dz;
                    }
                    team_barrier();
                    double lz = mfma4(1.0, lsum - log_ge1_d_vc(wprod), 0.0);
                    lz += dpp_d<0x124>(lz); lz += dpp_d<0x128>(lz);
                    if (bb == 0 && hi == 0) res[DP + 1] = lz;
                }
                STAMP(6);
                TSTAMP(6);
            }
        }
    }

    if (!TEAM && !is_state && !is_bk && !is_om) {
        // ================================================================= row wave
        // responses of this lane's rows as a bit mask (row of iteration `it`: wr*64 + lane + it*64*RW)
        unsigned long long ybits = 0;
        {
            int it = 0;
            for (int r = wr * 64 + lane; r < n; r += 64 * RW, ++it)
                if (a.y[row0 + r]) ybits |= 1ull << it;
        }
        STAMP_INIT;
        // the row pass is the longest link of a leapfrog's critical chain: it wins the SIMD's issue arbitration
        // against the state wave (of another chain) it shares the SIMD with, whose bookkeeping has slack
        __builtin_amdgcn_s_setprio(EPX_PRIO_R);
        // One chain per workgroup, a site of at most 2 x 64 x RW rows (C2: 200): a lane's two rows are the same in every
        // pass -- they stay in registers for the whole piece instead of being read from LDS behind every job (16 reads
        // and their latency on the critical chain of the leapfrog)
        constexpr bool KEEP_ROWS = BKW && DP <= 16;
        const bool one_round = KEEP_ROWS && n <= 2 * 64 * RW;
        double xk0[KEEP_ROWS ? DP : 1], xk1[KEEP_ROWS ? DP : 1];
        bool k_has = false, k_two = false;
        if constexpr (KEEP_ROWS) {
            const int r = wr * 64 + lane, r1 = r + 64 * RW;
            k_has = one_round && r < n; k_two = r1 < n;
#pragma unroll
            for (int j = 0; j < DP; ++j) { xk0[j] = 0.0; xk1[j] = 0.0; }
            if (k_has) {
                const int r1c = k_two ? r1 : r;
                const double2 *row0p = reinterpret_cast<const double2 *>(Xs + (size_t)r * DP);
                const double2 *row1p = reinterpret_cast<const double2 *>(Xs + (size_t)r1c * DP);
                const int sw0 = (r / RPL) & (SPR - 1), sw1 = (r1c / RPL) & (SPR - 1);
#pragma unroll
                for (int jp = 0; jp < SPR; ++jp) {
                    const double2 v = row0p[jp ^ sw0], w = row1p[jp ^ sw1];
                    xk0[2 * jp] = v.x; xk0[2 * jp + 1] = v.y;
                    xk1[2 * jp] = w.x; xk1[2 * jp + 1] = w.y;
                }
            }
        }
        for (int seq = 1;; ++seq) {
            const duo_lds_f64 *job = slot + JOB;
            double alpha = 0.0;
            double bs[DP];
            int got;
            if constexpr (BKW && DP <= 16) {
                // One chain per workgroup (layout 6): every look at the job's word asks for the job's DATA as well -- the LDS
                // serves a wave's reads in order, so data requested behind a word that reads `seq` is that job's -- and the
                // look that finds the word finds (alpha, beta) with it: one LDS round trip less on the critical chain of a
                // leapfrog that is nothing but such links (HISTORY.md section 3.1f, round 4).  The compiler barriers keep
                // the reads inside the look and in this order.
                got = DUO_TIMEOUT;
                for (int spin = 0; spin < DUO_SPIN_LIMIT; ++spin) {
                    asm volatile("" ::: "memory");
                    const int v = *f_job;
                    asm volatile("" ::: "memory");
                    alpha = job[0];
#pragma unroll
                    for (int j = 0; j < DP; ++j) bs[j] = job[1 + j];
                    asm volatile("" ::: "memory");
                    const int vu = __builtin_amdgcn_readfirstlane(v);
                    if (vu == seq || vu == DUO_EXIT) { got = vu; break; }
                    __builtin_amdgcn_s_sleep(EPX_DUO_SLEEP_BKW);
                }
            } else got = duo_wait(f_job, seq);
            STAMP(5);
            if (got != seq) {
                if (got == DUO_TIMEOUT && lane == 0) atomicOr(a.err, 1);
#ifdef EPX_STAMPS
                if (a.stamps && team == 0 && wr == 0 && lane == 0) { a.stamps[(size_t)blockIdx.x * 8 + 5] += tacc[5]; a.stamps[(size_t)blockIdx.x * 8 + 6] += tacc[6]; }
#endif
                return;                               // [parity] layouts 5/6 only (polled flags with DUO_SPIN_LIMIT, no barriers)
            }
            if constexpr (!(BKW && DP <= 16)) alpha = job[0];
            if constexpr (BKW && DP <= 16) {
                // (beta came with the look above)
            } else if constexpr (DP <= 16) {
                // beta to every lane by LDS reads at a uniform address (a broadcast): 8 reads beside the rows' instead of
                // 32 v_readlane in front of them -- at 16 columns the pass is one round of fixed costs, and the vector
                // pipe is what it runs on (layout 6 at C2: HISTORY.md section 3.1f).  The writer keeps the entries beyond
                // D at zero.  32 columns stay on scalar registers: 64 more vector registers would spill the row wave.
#pragma unroll
                for (int j = 0; j < DP; ++j) bs[j] = job[1 + j];
            } else {
                double beta_l = job[1 + (lane < DP ? lane : 0)];
                if (lane >= D) beta_l = 0.0;
#pragma unroll
                for (int j = 0; j < DP; ++j) bs[j] = readlane_d(beta_l, j);
            }
            // ---- fused row pass: f = alpha + x.beta, g = y - sigmoid(f), acc += g x   (nuts_gradient.inc)
            double acc[DP];
#pragma unroll
            for (int j = 0; j < DP; ++j) acc[j] = 0.0;
            double da = 0.0, ll = 0.0, wprod = 1.0;
            unsigned long long yb = ybits;
            // two rows per round: their logistic terms (a chain of ~25 dependent FP64 operations each)
            // overlap; the sums still take the rows in order, so every value equals the row-by-row loop's.
            auto round2 = [&](const double (&x0)[DP], const double (&x1)[DP], bool two) {
                double fa0 = alpha, fa1 = 0.0, fb0 = alpha, fb1 = 0.0;
#pragma unroll
                for (int j = 0; j < DP; j += 2) {
                    fa0 = fma(x0[j], bs[j], fa0); fa1 = fma(x0[j + 1], bs[j + 1], fa1);
                    fb0 = fma(x1[j], bs[j], fb0); fb1 = fma(x1[j + 1], bs[j + 1], fb1);
                }
                const double fa = fa0 + fa1, fb = fb0 + fb1;
                double la, wa, ga, lb, wb, gb;
                logistic_split2(fa, fb, (double)(yb & 1ull), (double)((yb >> 1) & 1ull), la, lb, wa, wb, ga, gb);
                yb >>= 2;
                lb = two ? lb : 0.0; wb = two ? wb : 1.0; gb = two ? gb : 0.0;
                ll += la; wprod *= wa; da += ga;
                ll += lb; wprod *= wb; da += gb;
#pragma unroll
                for (int j = 0; j < DP; ++j) acc[j] = fma(gb, x1[j], fma(ga, x0[j], acc[j]));
            };
            int r = wr * 64 + lane;
            if constexpr (KEEP_ROWS) {
                if (one_round) {
                    if (k_has) round2(xk0, xk1, k_two);
                    r = n;                                   // (no other round)
                }
            }
            // the other rounds.  A lane whose second row is beyond n re-reads its first row and adds zeros.
            for (; r < n; r += 2 * 64 * RW) {
                const int r1 = r + 64 * RW;
                const bool two = r1 < n;
                const int r1c = two ? r1 : r;
                const double2 *row0p = reinterpret_cast<const double2 *>(Xs + (size_t)r * DP);
                const double2 *row1p = reinterpret_cast<const double2 *>(Xs + (size_t)r1c * DP);
                const int sw0 = (r / RPL) & (SPR - 1), sw1 = (r1c / RPL) & (SPR - 1);
                double x0[DP], x1[DP];
#pragma unroll
                for (int jp = 0; jp < SPR; ++jp) {
                    const double2 v = row0p[jp ^ sw0];
                    x0[2 * jp] = v.x; x0[2 * jp + 1] = v.y;
                }
#pragma unroll
                for (int jp = 0; jp < SPR; ++jp) {
                    const double2 v = row1p[jp ^ sw1];
                    x1[2 * jp] = v.x; x1[2 * jp + 1] = v.y;
                }
                round2(x0, x1, two);
            }
            duo_lds_f64 *res = slot + (RW == 1 ? 0 : RESO + wr * RREC);
            {
            ll -= log_ge1_d(wprod);
            butterfly<DP, 5>(acc, lane);
            // (layout 6 -- one chain per workgroup, two row waves whose partial sums the state wave adds -- never had
            // layout 1's order of additions: its two sums take the packed form, 22 instead of 40 vector instructions of
            // a pass that is all fixed costs; layout 5 keeps wave_sum2 and with it the draws of layout 1, bit for bit)
            if constexpr (BKW) wave_sum2_packed(da, ll); else wave_sum2(da, ll);
            if ((lane & ((1 << (6 - LOG)) - 1)) == 0) res[lane >> (6 - LOG)] = acc[0];
            if (lane == 0) { res[DP] = da; res[DP + 1] = ll; }
            duo_publish_c<BKW>(f_res + wr, seq);
            }
            STAMP(6);
        }
    }

    if constexpr (BKW) {
        if (is_om) {
            // ============================================================= cavity-term wave (one chain per workgroup)
            // Omega v for the position of every job, v = phi - mu published with the job: lane e holds row e (the rows
            // beyond 64 on lanes 0, 1 of a second register); OUP column pairs per round, loads first
            const duo_lds_f64 *vj = slot + VOFF;
            const int e0 = lane < dm ? lane : dm - 1;
            const double2 *Op = reinterpret_cast<const double2 *>(Oms) + e0;
            const int rcw = lane < 2 ? lane : 1;
            const double2 *Tp = reinterpret_cast<const double2 *>(Ots + (size_t)rcw * tstride);
            for (int seq = 1;; ++seq) {
                const int got = duo_wait(f_job, seq);
                if (got != seq) {
                    if (got == DUO_TIMEOUT && lane == 0) atomicOr(a.err, 1);
                    return;                           // [parity] layouts 5/6 only (polled flags)
                }
                double ov0 = 0.0, ov1 = 0.0;
                for (int p0 = 0; p0 < npad; p0 += OUP) {
                    double2 o[OUP], vp[OUP], tt[OUP];
#pragma unroll
                    for (int u = 0; u < OUP; ++u) {
                        o[u] = Op[(size_t)(p0 + u) * dm];
                        { const lds_v2f64 t2 = *(lds_v2f64_p)(vj + 2 * ((p0 + u) & 31)); vp[u].x = t2.x; vp[u].y = t2.y; }      // uniform address: a broadcast
                        if constexpr (NV > 1) tt[u] = Tp[p0 + u]; else tt[u] = o[u];
                    }
#pragma unroll
                    for (int u = 0; u < OUP; ++u) {
                        ov0 = fma(o[u].x, vp[u].x, ov0);
                        ov0 = fma(o[u].y, vp[u].y, ov0);
                        if constexpr (NV > 1) {
                            ov1 = fma(tt[u].x, vp[u].x, ov1);
                            ov1 = fma(tt[u].y, vp[u].y, ov1);
                        }
                    }
                }
                if constexpr (NV > 1) {                                      // columns 64.. : the tail rows by symmetry
                    const double a0 = Ots[e0], a1 = Ots[tstride + e0];
                    const double b0 = Ots[(size_t)rcw * tstride + dm], b1 = Ots[(size_t)rcw * tstride + dm + 1];
                    const double w0 = vj[64], w1 = vj[65];
                    ov0 = fma(a0, w0, ov0); ov1 = fma(b0, w0, ov1);
                    ov0 = fma(a1, w1, ov0); ov1 = fma(b1, w1, ov1);
                }
                slot[OVOFF + lane] = lane < d ? ov0 : 0.0;
                if constexpr (NV > 1) slot[OVOFF + 64 + lane] = 64 + lane < d ? ov1 : 0.0;
                duo_publish_c<BKW>(f_ov, seq);
            }
        }
    }

    // ===================================================================== state wave
    const int wt = 0;
    // global memory of the chain: the tree stack (unless it is in LDS), then the cold store (COLD)
    double *stk_l = reinterpret_cast<double *>(smem + a.off_stack) + (size_t)team * a.max_depth * SREC;
    const size_t g_stack = (size_t)a.max_depth * SREC;                 // the cold store sits behind the stack's place
    // (pieced launches: the tree stack and the cold store belong to the persistent WORKGROUP, so no line of them is ever
    // cached by two XCDs; what a chain carries from piece to piece goes through the checkpoint record)
    gdouble *stk_g = (STL && !COLD) ? nullptr : uniform_ptr(a.stack + ((size_t)(segmented ? (int)blockIdx.x : sb) * a.chains + chain) * a.stack_stride);
    // (the checkpoint record's address is formed where it is used: nothing of a pieced launch stays live through the loops)
    auto ck_rec = [&](int t_boundary) -> double * { return piece_record(a, sb, t_boundary, chain, NV); };
    // When the whole stack does not fit, its lowest levels still may (stack_lds_levels; level l is touched by every
    // 2^(l+1)-th leaf: two levels take 3 of 4 records off the global store and its latency off the bookkeeping)
    // (those records are packed: a vector takes stack_ps >= P doubles there, not NV x 64)
    const int stk_nl = STL ? 0 : a.stack_lds_levels, stk_ps = a.stack_ps;
    duo_lds_f64 *stk_h = duo_lds_at(smem + a.off_stack) + team * stk_nl * 2 * stk_ps;
    int stk_lane = lane;                // (the loop's opaque copy of the lane index: set at the top of every iteration)
    auto ld_stk = [&](int l, int v, int i) -> double {
        const int off = l * SREC + (v * NV + i) * 64 + stk_lane;
        if constexpr (STL) return stk_l[off];
        else { if (l < stk_nl) return stk_h[(2 * l + v) * stk_ps + i * 64 + stk_lane]; return stk_g[off]; }
    };
    auto st_stk = [&](int l, int v, int i, double x) {
        const int off = l * SREC + (v * NV + i) * 64 + stk_lane;
        if constexpr (STL) stk_l[off] = x;
        else { if (l < stk_nl) stk_h[(2 * l + v) * stk_ps + i * 64 + stk_lane] = x; else stk_g[off] = x; }
    };
    const RngKey key = make_key((uint64_t)a.seeds[sb], chain);
    const bool laplace = (model == 4);

    // The 13 vectors that change once per subtree / transition (current sample and its gradient, both
    // tree ends, rho, p-sharps, Welford sums): registers, or (COLD) a per-chain store in global memory --
    // at two registers per vector they are 52 VGPRs that the state wave spilled to scratch
    using CV = typename std::conditional<COLD, GVec, V>::type;
    V mu, inv_e, zq, zp, zg;
    CV qs, gs, pq, pp, pg, mq, mp, mg, rho, psp, psm, wmean, wm2, bq, bg;
    gdouble *cold = COLD ? stk_g + g_stack : nullptr;
    auto bind = [&](CV &x, int which, int ln) {
        if constexpr (COLD) { x.v.b = cold + (size_t)which * NV * 64; x.v.lane = ln; x.v.len = P; }
    };
#define EPX_BIND_COLD(ln)                                                                              \
    bind(qs, GV_QS, ln); bind(gs, GV_GS, ln); bind(pq, GV_PQ, ln); bind(pp, GV_PP, ln); bind(pg, GV_PG, ln); \
    bind(mq, GV_MQ, ln); bind(mp, GV_MP, ln); bind(mg, GV_MG, ln); bind(rho, GV_RHO, ln);                 \
    bind(psp, GV_PSP, ln); bind(psm, GV_PSM, ln); bind(wmean, GV_WMEAN, ln); bind(wm2, GV_WM2, ln); \
    bind(bq, GV_BQ, ln); bind(bg, GV_BG, ln)
    EPX_BIND_COLD(lane);
    // the scalars that are touched once per transition (COLD: in the chain's global store, see GScal)
    using CS = typename std::conditional<COLD, GScal, RScal>::type;
    CS lps, da_mu, s_bar, x_bar, da_count, va_n, eps_sum, acc_sum, depth_sum, nleap_tot, plp, mlp, b_plp, lsw, t_end_c;
    if constexpr (COLD) {
        gdouble *sc0 = cold + (size_t)GV_SCAL * NV * 64;
        lps.p = sc0; da_mu.p = sc0 + 1; s_bar.p = sc0 + 2; x_bar.p = sc0 + 3; da_count.p = sc0 + 4; va_n.p = sc0 + 5;
        eps_sum.p = sc0 + 6; acc_sum.p = sc0 + 7; depth_sum.p = sc0 + 8; nleap_tot.p = sc0 + 9;
        plp.p = sc0 + 10; mlp.p = sc0 + 11; b_plp.p = sc0 + 12; lsw.p = sc0 + 13; t_end_c.p = sc0 + 14;
    }
    lps = 0.0; plp = 0.0; mlp = 0.0; b_plp = 0.0; lsw = 0.0;
    t_end_c = (double)t_end;                      // (read once per transition)
    double zlp = 0, b_key = 0;
    FORV {
        const int e = lane + 64 * i;
        mu.v[i] = e < d ? a.cav_mu[(size_t)k * d + e] : 0.0;
        inv_e.v[i] = 1.0;
        zq.v[i] = 0; zp.v[i] = 0; zg.v[i] = 0;
    }
    FORV {
        gs.v[i] = 0; pq.v[i] = 0; pp.v[i] = 0; pg.v[i] = 0;
        mq.v[i] = 0; mp.v[i] = 0; mg.v[i] = 0; rho.v[i] = 0; psp.v[i] = 0; psm.v[i] = 0;
    }
    if (resume) {
        // the sample and the Welford sums of the piece before this one (checkpoint record: qs, wmean, wm2, metric, scalars)
        double *ckp = ck_rec(t_begin);
        FORV {
            qs.v[i] = ck_load(ckp + (0 * NV + i) * 64 + lane);
            wmean.v[i] = ck_load(ckp + (1 * NV + i) * 64 + lane);
            wm2.v[i] = ck_load(ckp + (2 * NV + i) * 64 + lane);
        }
    } else {
        FORV { wmean.v[i] = 0.0; wm2.v[i] = 0.0; }
    }
    if (!resume) {
        const double *lastp = a.last + ((size_t)k * a.chains + chain) * P;
        FORV {
            const int e = lane + 64 * i;
            double q0 = 0.0;
            if (e < P) {
                if (a.init_mode == 2) q0 = lastp[e];
                else if (a.init_mode == 0) {
                    double u1, u2;
                    rng_u2(key, 0, K_INIT, (uint32_t)(e >> 1), 0, u1, u2);
                    q0 = -2.0 + 4.0 * ((e & 1) ? u2 : u1);
                }
            }
            qs.v[i] = q0;
        }
    }
    // adaptation state (stepsize_adaptation.hpp / windowed_adaptation.hpp @ Stan 2.17)
    const double DELTA = 0.8, GAMMA = 0.05, T0 = 10.0, KAPPA = 0.75, LOG08 = -0.2231435513142097558;
    double eps = 1.0;
    da_mu = log(10.0); s_bar = 0.0; x_bar = 0.0; da_count = 0.0;
    int va_init_buf = 75, va_term = 50, va_base = 25;
    if (va_init_buf + va_base + va_term > a.warmup && a.warmup >= 20) {
        va_init_buf = (int)(0.15 * a.warmup);
        va_term = (int)(0.1 * a.warmup);
        va_base = a.warmup - (va_init_buf + va_term);
    }
    int va_counter = 0, va_wsize = va_base, va_next = va_init_buf + va_base - 1;
    va_n = 0.0;
    eps_sum = 0.0; acc_sum = 0.0; depth_sum = 0.0; nleap_tot = 0.0;
    double ngrad = 0;
    int ndiv = 0, npost = 0, kept = 0, failed = 0;
    int t = 0, mode = MODE_INIT, depth = 0, leaf = 0, nleaf = 1, fwd = 1, nleap = 0, divergent = 0, init_try = 0;
    int ss_trial = 0, ss_dir = 0, ss_after_update = 0;
    uint32_t ss_t = 0;
    double H0 = 0, sum_metro = 0, eps_l = 0;
    double u_dir = 0.0, gum = 0.0;
    double dhb = 0.0, lw_m = -INFINITY, lw_s = 0.0;
    int bail = 0;
    FORV { zq.v[i] = qs.v[i]; }
    const bool teacher = a.eps_in != nullptr;       // fixed step size / metric (test hook)
    if (teacher) {
        eps = a.eps_in[(size_t)sb * a.chains + chain];
        if (a.inv_e_in) {
            const double *ie = a.inv_e_in + ((size_t)sb * a.chains + chain) * P;
            FORV { const int e = lane + 64 * i; if (e < P) inv_e.v[i] = ie[e]; }
        }
    }
    // opt-in carried adaptation: last call's step size of the chain, the site's pooled sample variances
    const bool carry = !teacher && a.carry_eps != nullptr && a.carry_eps[(size_t)k * a.chains + chain] > 0.0;
    if (carry) {
        eps = a.carry_eps[(size_t)k * a.chains + chain];
        da_mu = log(10.0 * eps);
        const double *cm = a.carry_metric + (size_t)k * P;
        FORV { const int e = lane + 64 * i; if (e < P) inv_e.v[i] = cm[e]; }
    }
    if constexpr (COLD) {
        if (resume) {
            double *ckp = ck_rec(t_begin);
            FORV inv_e.v[i] = ck_load(ckp + (3 * NV + i) * 64 + lane);
            const double ckv = ck_load(ckp + 4 * NV * 64 + lane);
#define EPX_CK_GET(idx, x) ck_assign(x, readlane_d(ckv, idx));
            EPX_CK_LIST(EPX_CK_GET)
#undef EPX_CK_GET
            ngrad -= 1.0;                             // the gradient at the restored sample is evaluated once more
            if (failed) {                             // it failed in its first piece, where everything was written:
                // hand the mark on to the piece after this one (every boundary has its own record), and leave
                ck_store(ck_rec(t_end) + 4 * NV * 64 + lane, ckv);
                piece_checkpoint_out();
                *f_job = DUO_EXIT;
                team_leave(true);
                return;                               // [parity] state wave: team_leave has seen the last B1
            }
        }
    }
    const uint32_t toff = (uint32_t)a.t_offset + 1u;
    // (row team: the subtree-level elementary functions are the lean ones -- a wave that completes a subtree keeps the three
    // other chains and the row team waiting at the pass's barrier, and libm's exp / log are ~10 x the instructions)
#define SM_EXP(x) (TEAM ? exp_d(x) : exp(x))
    auto flush_dh = [&](int cnt) {
        const bool ok = lane < cnt;
        const double dh = ok ? dhb : -INFINITY;
        const double mb = wave_max(dh);
        const double m_new = fmax(lw_m, mb);
        double w = 0.0, me = 0.0;
        if (ok) {
            w = (m_new == -INFINITY) ? 0.0 : SM_EXP(dh - m_new);
            me = dh > 0 ? 1.0 : SM_EXP(dh);
        }
        wave_sum2(w, me);
        const double scale = (lw_m == -INFINITY) ? 0.0 : SM_EXP(lw_m - m_new);
        lw_s = lw_s * scale + w;
        lw_m = m_new;
        sum_metro += me;
    };
#undef SM_EXP

    if constexpr (BKW) {
        if (is_bk) {
            // ============================================================= bookkeeping wave (owns the chain)
            V sent_e, in_q, in_p, in_g;
            double sent_eps = 0.0;
            int gen = 0;
            // the integration state the state wave has to continue from; generation = number of records so far
            auto post = [&](int cmd) {
                double *cr = ctrl + ((gen + 1) & 1) * CREC;
                FORV {
                    cr[(0 * NV + i) * 64 + lane] = zq.v[i]; cr[(1 * NV + i) * 64 + lane] = zp.v[i];
                    cr[(2 * NV + i) * 64 + lane] = zg.v[i]; cr[(3 * NV + i) * 64 + lane] = inv_e.v[i];
                    sent_e.v[i] = inv_e.v[i];
                }
                if (lane == 0) { cr[4 * NV * 64] = eps_l; cr[4 * NV * 64 + 1] = (double)cmd; }
                sent_eps = eps_l;
                ++gen;
                duo_publish(f_ctl, gen);
            };
            post(DUO_RESTART);                      // the initial point, eps_l = 0: the first "leapfrog" is its gradient
            for (int mexp = 1;; ++mexp) {
                const int got = duo_wait_ge(f_mail, mexp);
                if (got < 0) { bail = 1; post(DUO_LEAVE); break; }       // the state wave gave up, or the wait timed out
                const double *m = mbox + (mexp & 1) * MREC;
                const int gen_m = (int)m[4 * NV * 64 + 2];
                FORV {
                    in_q.v[i] = m[(0 * NV + i) * 64 + lane]; in_p.v[i] = m[(1 * NV + i) * 64 + lane];
                    in_g.v[i] = m[(2 * NV + i) * 64 + lane];
                }
                double lpt = 0.0;
                FORV lpt += m[(3 * NV + i) * 64 + lane];
                const double ll_m = m[4 * NV * 64];
                duo_publish(f_ack, mexp);           // the entry is in registers: the state wave may reuse it
                if (gen_m != gen) continue;         // integrated past a change of state: dropped
                FORV { zq.v[i] = in_q.v[i]; zp.v[i] = in_p.v[i]; zg.v[i] = in_g.v[i]; }
                double ks = 0.0;
                FORV ks += inv_e.v[i] * zp.v[i] * zp.v[i];
                wave_sum2(lpt, ks);
                zlp = lpt + ll_m;
                const double kin = 0.5 * ks;
                ngrad += 1.0;
                V n_rho, n_psl, psr;
#define EPX_CHAIN_EXIT { post(DUO_LEAVE); break; }
#define EPX_DBG_EXIT { post(DUO_LEAVE); bail = 4; break; }
#define STAMP_LEAF do { } while (0)
#include "nuts_state_machine.inc"
#undef STAMP_LEAF
#undef EPX_CHAIN_EXIT
#undef EPX_DBG_EXIT
                // the state wave keeps integrating from the state it handed over; tell it only if that is no
                // longer where (or how) the trajectory continues
                int moved = (eps_l != sent_eps) ? 1 : 0;
                FORV {
                    moved |= (zq.v[i] != in_q.v[i]) | (zp.v[i] != in_p.v[i]) | (zg.v[i] != in_g.v[i]) | (inv_e.v[i] != sent_e.v[i]);
                }
                if (__any(moved)) post(DUO_RESTART);
            }
            *f_ack = DUO_NO_MORE;                   // whatever the state wave still hands over needs no answer
        }
    }

    // per element: column of X'g its chain-rule term reads, clamped into the slot
    int jdx[NV];
    FORV {
        const int e = lane + 64 * i;
        int j;
        if (model == 0) j = e - 1;
        else if (model == 1) j = e - 3;
        else if (model == 2) j = e <= D ? e - 1 : e - d - 1;
        else j = e < 2 + D ? e - 2 : (e < d ? e - 2 - D : e - d - 1);
        jdx[i] = j < 0 ? 0 : (j > DP - 1 ? DP - 1 : j);
    }
    const int rc = lane < 2 ? lane : 1;             // tail row of this lane (lanes beyond the tail re-read row 1)
    auto xtg = [&](int j) -> double {                  // X'g[j] summed over the chain's row waves, in wave order
        if constexpr (RW == 1) return slot[j];
        else {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < RW; ++w) s += slot[RESO + w * RREC + j];
            return s;
        }
    };

    auto rows_in = [&](int sq_) -> bool {               // the row waves' results of job sq_ are in
        if constexpr (TBAR) { team_barrier(); return true; }
        else if constexpr (BKW) {
            // one chain per workgroup: the words of the two row waves AND of the cavity-term wave in ONE look (lane w
            // reads f_res[w], lane RW reads f_ov, a ballot says whether all stand at sq_).  Three waits one after the other
            // were three LDS round trips on the critical stretch even when the second and third word had long been there.
            duo_flag_t *mine = lane < RW ? f_res + lane : (lane == RW ? f_ov : f_res);
            const int llm = ~0;
            for (int spin = 0; spin < DUO_SPIN_LIMIT; ++spin) {
                const int v = *mine;
                if (__builtin_amdgcn_ballot_w64(v != DUO_EXIT && (v & llm) == sq_) == ~0ull) { asm volatile("" ::: "memory"); return true; }
                if (__builtin_amdgcn_ballot_w64(v == DUO_EXIT) != 0ull) return false;
                __builtin_amdgcn_s_sleep(EPX_DUO_SLEEP_BKW);
            }
            return false;
        }
        else {
            bool ok = true;
            for (int w = 0; w < RW; ++w) ok &= duo_wait(f_res + w, sq_) == sq_;
            return ok;
        }
    };
    // TEAM, barrier hand-offs: the bookkeeping of a finished SUBTREE or TRANSITION (weights, Philox draws, copies through the
    // cold store: 5 000-17 000 cycles against ~3 500 for an ordinary leaf) outlasts the team's pass, and the row waves and
    // the three other chains would wait for it at "results in".  At a few points of that code (EPX_SM_YIELD in
    // nuts_state_machine.inc, each with an estimate of the cycles still ahead) the wave looks at the clock: if the cycles
    // since its job went out plus the ones ahead exceed a.yield_cycles (a pass plus the wait that is cheaper than a lost
    // pass of one chain) it takes the two barriers of the pass WITHOUT touching its job -- the row waves work the same job
    // again, same results -- and goes on with the bookkeeping beside the next pass: one lost pass of ONE chain instead of
    // a wait of all four.  The chain's arithmetic does not change (same draws); only how many passes the team makes
    // depends on the clock.
    // [parity] a yield is one B2 + one B1, taken where the wave's next barrier is a B2: the alternation is kept.
    unsigned long long t_job = 0;
    auto sm_yield = [&](int ahead) {
        if constexpr (TBAR) {
            if (a.yield_cycles > 0 && __builtin_amdgcn_s_memtime() - t_job + (unsigned long long)ahead > (unsigned long long)a.yield_cycles) {
                team_barrier();
                team_barrier();
                t_job = __builtin_amdgcn_s_memtime();
            }
        }
    };
    // (zq, zp, zg) holds the last finished leapfrog state; `pending`: its bookkeeping is still to run
    double f_lpt = 0.0, f_ks = 0.0, f_ll = 0.0;     // its log density / kinetic energy, not yet summed over the lanes
    bool pending = false;
    int seq = 0, gen = 0, mseq = 0, ctl_pre = -1, ack_pre = 0;
    (void)gen; (void)mseq; (void)ctl_pre; (void)ack_pre;
    const int lane0 = lane;
    // ---- the critical-path shortcut (m4b / m5b, one row wave).  (alpha, beta) of the NEXT position depend on
    // 3 (D + 1) of the P coordinates only, and the hierarchical structure is the same for every one of them:
    //   beta_j = mu_b[j] + etb[j] exp(lsig_b[j])   (lane j < D)      alpha = mu_a + eta exp(lsig_a)   (lane LA)
    // with gradients  d mu = -Ov + t,  d raw = t exp(lsig) - prior'(raw),  d lsig = -Ov + t raw exp(lsig),
    // t = (X'g)[j] for beta_j and sum g for alpha.  Lane j keeps ITS triple (position, half-kicked momentum,
    // metric) of the position in flight in registers -- a "view" of the state vectors in the row waves' lane
    // order.  When the row wave's sums arrive, the view alone gives the next (alpha, beta): ~60 dependent
    // instructions instead of the whole chain rule + kick + drift + transforms with their cross-lane gathers;
    // the job goes out, THEN the full vectors are brought up to date (same operations on the same values: the
    // view and the vectors agree bit for bit, and so do the draws with layout 1).
    constexpr int LA = 32;                              // the lane that carries alpha's triple (D <= 32)
    const bool fast_ok = model >= 3;
    const bool v_lane = lane0 < D || lane0 == LA;
    const int ve1 = !v_lane ? 0 : (lane0 == LA ? 0 : 2 + lane0);            // location:  mu_a | mu_b[j]
    const int ve2 = !v_lane ? 0 : (lane0 == LA ? d : d + 1 + lane0);        // raw:       eta  | etb[j]
    const int ve3 = !v_lane ? 0 : (lane0 == LA ? 1 : 2 + D + lane0);        // log scale: lsig_a | lsig_b[j]
    double vq1 = 0, vq2 = 0, vq3 = 0, vp1 = 0, vp2 = 0, vp3 = 0, vm1 = 1, vm2 = 1, vm3 = 1, vex3 = 1, vo1 = 0, vo3 = 0;
    const double vmu1 = gatherV(mu, ve1), vmu3 = gatherV(mu, ve3);          // (cavity mean at the view's coordinates)
    (void)vmu1; (void)vmu3;
    bool fast_pub = false;                              // the job of the position in flight went out by the shortcut
    double job_eps = 0.0;
    STAMP_INIT;

    for (; !is_bk && !bail;) {
        // `lane` is re-derived through an opaque move every leapfrog, otherwise the per-element index
        // arithmetic below is hoisted out of the loop and spilled (as in k_nuts_spec)
        int lane_v = lane0;
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v;
        stk_lane = lane;
        EPX_BIND_COLD(lane);
        if constexpr (BKW) {
            // ---- the bookkeeping wave's word: a record of a new generation means "continue from here instead"
            // (read right after the last job went out, see below: a record that arrives in between is seen one leapfrog
            // later, which the generation numbers allow)
            const int ctl = gen == 0 ? duo_wait_ge(f_ctl, 1) : __builtin_amdgcn_readfirstlane(ctl_pre >= 0 ? ctl_pre : *f_ctl);
            ctl_pre = -1;
            if (ctl < 0) { bail = 1; break; }
            if (ctl != gen) {
                if (fast_pub) {                    // the job in flight continues a trajectory nobody wants: let it land
                    if (!rows_in(seq)) bail = 1;                    // (rows AND cavity-term wave: that one reads v, it must be done, too)
                    if (bail) break;
                }
                const double *cr = ctrl + (ctl & 1) * CREC;
                FORV {
                    zq.v[i] = cr[(0 * NV + i) * 64 + lane]; zp.v[i] = cr[(1 * NV + i) * 64 + lane];
                    zg.v[i] = cr[(2 * NV + i) * 64 + lane]; inv_e.v[i] = cr[(3 * NV + i) * 64 + lane];
                }
                eps_l = cr[4 * NV * 64];
                const int cmd = (int)cr[4 * NV * 64 + 1];
                gen = ctl;
                if (cmd == DUO_LEAVE) break;
                fast_pub = false;
            }
        }
        // ---- first half of the leapfrog from (zq, zp, zg): speculative while `pending`.  When the shortcut has
        // already sent the job of this position, the full vectors are only needed AFTER the bookkeeping below
        // (which leaves (zq, zp, zg) alone unless it restarts the trajectory): computing them there keeps
        // three vectors out of the bookkeeping's register budget
        V sq, sp, sg, eq;
        double sa = 0.0, eta = 0.0, sb2 = 0.0;
        auto first_half = [&]() {
            FORV sp.v[i] = zp.v[i] + 0.5 * eps_l * zg.v[i];
            FORV sq.v[i] = zq.v[i] + eps_l * inv_e.v[i] * sp.v[i];
            FORV eq.v[i] = exp_d(sq.v[i]);
            if (model == 0) { sa = elemU(eq, 0); eta = elemU(sq, d); }
            else if (model == 1) { sa = elemU(eq, 0); sb2 = elemU(eq, 1); eta = elemU(sq, 2); }
            else if (model == 2) { sa = elemU(eq, 0); eta = elemU(sq, d); }
            else { sa = elemU(eq, 1); eta = elemU(sq, d); }
        };
        const bool sent = fast_pub;
        if (!sent) first_half();
        if (!fast_pub) {
            double alpha, beta_l;
            if (model == 0) { alpha = eta * sa; beta_l = gatherV(sq, 1 + lane); }
            else if (model == 1) { alpha = eta * sa; beta_l = gatherV(sq, 3 + lane) * sb2; }
            else if (model == 2) { alpha = eta * sa; beta_l = gatherV(sq, d + 1 + lane) * gatherV(eq, 1 + lane); }
            else {
                alpha = elemU(sq, 0) + eta * sa;
                beta_l = gatherV(sq, 2 + lane) + gatherV(sq, d + 1 + lane) * gatherV(eq, 2 + D + lane);
            }
            // ---- hand (alpha, beta) to the row waves
            duo_lds_f64 *job = slot + JOB;
            if constexpr (TEAM || DP <= 16) beta_l = lane < D ? beta_l : 0.0;         // (the padding columns of the B operand stay finite; 16 columns: the row waves read the entries as they are)
            if (lane < DP) job[BOFF + lane] = beta_l;
            if (lane == 0) job[0] = alpha;
            if constexpr (RW > 1) {
                FORV { const int e = lane + 64 * i; if (e < VN) slot[VOFF + e] = e < d ? sq.v[i] - mu.v[i] : 0.0; }       // v for the row waves' cavity term
            }
            ++seq;
            if constexpr (TBAR) { team_barrier(); t_job = __builtin_amdgcn_s_memtime(); } else duo_publish_c<BKW>(f_job, seq);
            job_eps = eps_l;
            if (fast_ok) {
                // (re)build the view of the position in flight from the vectors: start of the chain, or the
                // bookkeeping restarted the trajectory elsewhere
                vq1 = gatherV(sq, ve1); vq2 = gatherV(sq, ve2); vq3 = gatherV(sq, ve3);
                vp1 = gatherV(sp, ve1); vp2 = gatherV(sp, ve2); vp3 = gatherV(sp, ve3);
                vm1 = gatherV(inv_e, ve1); vm2 = gatherV(inv_e, ve2); vm3 = gatherV(inv_e, ve3);
                vex3 = gatherV(eq, ve3);
            }
        }
        __builtin_amdgcn_s_setprio(EPX_PRIO_S_BG);      // from here to the row waves' answer nothing waits for this wave
        STAMP(0);

        // ---- while they sweep the rows: the bookkeeping of the leapfrog that finished before this one
        if (!BKW && pending) {
            pending = false;
            // the two reductions only the bookkeeping needs: off the critical path
            if constexpr (TBAR) {
                // (the row waves stored the log-likelihood of that leapfrog behind its "results are in" and in front of the
                // barrier just passed)
                double l4 = 0.0;
#pragma unroll
                for (int w = 0; w < RW; ++w) l4 += slot[RESO + w * RREC + DP + 1];
                f_ll = uniform_d(l4);
            }
            if constexpr (TEAM) wave_sum2_packed(f_lpt, f_ks); else wave_sum2(f_lpt, f_ks);
            zlp = f_lpt + f_ll;
            const double kin = 0.5 * f_ks;
            const int fwd_was = fwd;
            ngrad += 1.0;
            V n_rho, n_psl, psr;
            int leave = 0, parked = 1;
            do {
#define EPX_CHAIN_EXIT { leave = 1; parked = 0; break; }
#define EPX_DBG_EXIT { leave = 2; parked = 0; break; }
#define STAMP_LEAF do { } while (0)
#define EPX_RESUME resume
#define EPX_T_END (int)(double)t_end_c
#define EPX_WAVE_SUM2(a_, b_) do { if constexpr (TEAM) wave_sum2_packed(a_, b_); else wave_sum2(a_, b_); } while (0)
#define EPX_SM_EXP(x) (TEAM ? exp_d(x) : exp(x))
#define EPX_SM_LOG(x) (TEAM ? log_pos_d(x) : log(x))
#define EPX_SM_LSE2(a_, b_) (TEAM ? log_sum_exp2_lean(a_, b_) : log_sum_exp2(a_, b_))
#define EPX_SM_YIELD(cycles_ahead_) sm_yield(cycles_ahead_)
#include "nuts_state_machine.inc"
#undef EPX_SM_YIELD
#undef EPX_SM_EXP
#undef EPX_SM_LOG
#undef EPX_SM_LSE2
#undef EPX_WAVE_SUM2
#undef EPX_RESUME
#undef EPX_T_END
#undef STAMP_LEAF
#undef EPX_CHAIN_EXIT
#undef EPX_DBG_EXIT
                parked = 0;
            } while (0);
            // The trajectory goes on from the state just booked -- so the job in flight is the wanted
            // one -- in two cases: the leaf was parked as a pending left sibling (`continue` inside the
            // include), or a subtree was completed and the next doubling extends the SAME end of the
            // tree (its first state is the leaf just booked; step size and metric only change between
            // transitions).  Everything else (other end, new transition, step-size trial) restarts.
            const bool same_end = mode == MODE_TREE && depth > 0 && fwd == fwd_was && eps_l == job_eps;
            if (leave || !(parked || same_end)) {
                // the job in flight continues a trajectory that is no longer wanted: let it land, drop it
                if (!rows_in(seq)) bail = 1;
                STAMP(3);
                if (bail || leave) { bail |= leave << 1; break; }
                fast_pub = false;               // the new start goes out by the full transforms at the loop top
                continue;
            }
        }
        // One chain per workgroup, models with per-coefficient scales: the view holds EVERY coordinate (d = 2 D + 2,
        // P = 3 D + 3), so while the shortcut keeps sending the jobs the state wave needs no vector at all -- the finished
        // state goes to the bookkeeping wave straight from the view's lanes (same values: the vectors' formulas, element
        // by element), and the vectors are only rebuilt from a control record when the trajectory restarts
        // (the row TEAM form does the same inside the one state wave: the finished state is re-laid from the view's lanes
        // to vector order through an LDS scratch line -- no first half on the full vectors, no gathers, no chain rule)
        const bool lean = (BKW || (TEAM && !a.no_spec)) && fast_ok && sent;
        if (sent && !lean) first_half();
        STAMP(1);

        // ---- cavity term Ov = Omega (phi - mu) of the position in flight ...
        V vv, Ov;
        FORV { const int e = lane + 64 * i; vv.v[i] = e < d ? sq.v[i] - mu.v[i] : 0.0; Ov.v[i] = 0.0; }
        int bzero = 0;
        asm volatile("" : "+v"(bzero));               // a register holding 0 the compiler cannot fold: address base
        if constexpr (RW == 1) {
            // OU column pairs per round, loads first: the LDS latency is paid once per round.  The padding
            // pairs (and absent tail rows) hold zeros, so they add nothing and the sums keep order and value
            const int e0 = lane < dm ? lane : dm - 1;
            const double2 *Op = reinterpret_cast<const double2 *>(Oms) + e0;
            const double2 *Tp = reinterpret_cast<const double2 *>(Ots + (size_t)rc * tstride);
            for (int p0 = 0; p0 < npad; p0 += OU) {
                const int pb = bzero + 8 * (p0 & 31);                // byte address (4 x lane) of v[2 p0] for ds_bpermute
                double2 o[OU], tt[OU];
#pragma unroll
                for (int u = 0; u < OU; ++u) {
                    o[u] = Op[(size_t)(p0 + u) * dm];
                    if constexpr (NV > 1) tt[u] = Tp[p0 + u]; else tt[u] = o[u];
                }
#pragma unroll
                for (int u = 0; u < OU; ++u) {
                    const int p = p0 + u;
                    // v[2p], v[2p+1] to every lane through the LDS crossbar (ds_bpermute with a uniform address:
                    // `zero` + immediate), not through v_readlane: the vector pipe is what this kernel runs out of
                    const double v0 = bcast_lds(vv.v[0], pb + 8 * u), v1 = bcast_lds(vv.v[0], pb + 8 * u + 4);
                    (void)p;
                    Ov.v[0] = fma(o[u].x, v0, Ov.v[0]);
                    Ov.v[0] = fma(o[u].y, v1, Ov.v[0]);
                    if constexpr (NV > 1) {
                        Ov.v[1] = fma(tt[u].x, v0, Ov.v[1]);
                        Ov.v[1] = fma(tt[u].y, v1, Ov.v[1]);
                    }
                }
            }
            if constexpr (NV > 1) {                                      // columns 64.. : the tail rows by symmetry
                const double a0 = Ots[e0], a1 = Ots[tstride + e0];
                const double b0 = Ots[(size_t)rc * tstride + dm], b1 = Ots[(size_t)rc * tstride + dm + 1];
                const double w0 = readlane_d(vv.v[1], 0), w1 = readlane_d(vv.v[1], 1);
                Ov.v[0] = fma(a0, w0, Ov.v[0]); Ov.v[1] = fma(b0, w0, Ov.v[1]);
                Ov.v[0] = fma(a1, w1, Ov.v[0]); Ov.v[1] = fma(b1, w1, Ov.v[1]);
            }
        }
        if constexpr (RW == 1) {
            FORV { const int e = lane + 64 * i; Ov.v[i] = e < d ? Ov.v[i] : 0.0; }
            if (fast_ok) { vo1 = gatherV(Ov, ve1); vo3 = gatherV(Ov, ve3); }
        }

        STAMP(2);
        // ---- their sums are in: chain rule back to (phi, eta, etb), second half of the leapfrog
        // BKW: the look that finds the three words (row waves, cavity-term wave) has asked for the sums behind them as
        // well -- same reasoning as on the row waves' side -- in xtg's order of additions (bit-identical)
        double pf_da = 0.0, pf_ll = 0.0, pf_t = 0.0, pf_vo1 = 0.0, pf_vo3 = 0.0;
        if constexpr (BKW) {
            duo_flag_t *mine = lane < RW ? f_res + lane : (lane == RW ? f_ov : f_res);
            const int llm = ~0;
            const int tj = lane == LA ? DP : (lane < DP ? lane : 0);
            bool ok = false;
            for (int spin = 0; spin < DUO_SPIN_LIMIT; ++spin) {
                asm volatile("" ::: "memory");
                const int v = *mine;
                asm volatile("" ::: "memory");
                pf_da = 0.0; pf_ll = 0.0; pf_t = 0.0;
#pragma unroll
                for (int w = 0; w < RW; ++w) {
                    pf_da += slot[RESO + w * RREC + DP]; pf_t += slot[RESO + w * RREC + tj];
                    pf_ll += slot[RESO + w * RREC + DP + 1];
                }
                pf_vo1 = slot[OVOFF + ve1]; pf_vo3 = slot[OVOFF + ve3];
                asm volatile("" ::: "memory");
                if (__builtin_amdgcn_ballot_w64(v != DUO_EXIT && (v & llm) == seq) == ~0ull) { ok = true; break; }
                if (__builtin_amdgcn_ballot_w64(v == DUO_EXIT) != 0ull) break;
                __builtin_amdgcn_s_sleep(EPX_DUO_SLEEP_BKW);
            }
            if (!ok) bail = 1;
        } else if (!rows_in(seq)) bail = 1;
        STAMP(3);
        __builtin_amdgcn_s_setprio(EPX_PRIO_S_CRIT);    // chain rule, half kick, drift, publish: the row waves wait for it
        if (bail) break;
        // (TEAM, lean rounds: the next job needs X'g and the cavity term, not sum g and the log-likelihood -- those are for
        // the books.  Summed in front of the view update they were an LDS round trip of their own: the compiler issues the
        // reads of X'g only behind their waits.  They are fetched behind the job's publication instead; the row waves
        // write their results at the END of the pass that starts there, thousands of cycles later.)
        // (TEAM: the reads the view update waits for -- X'g of the lane's column, the cavity term at the view's two
        // coordinates -- go out FIRST, in the block behind the barrier, with the sums' reads behind them: left where the
        // source has them, behind two branches, they were issued only after the sums' waits: a second LDS round trip)
        double et = 0.0, evo1 = 0.0, evo3 = 0.0;
        constexpr bool EARLYT = TBAR;
        if constexpr (EARLYT) {
            et = xtg(lane == LA ? DP : (lane < DP ? lane : 0));
            evo1 = slot[OVOFF + ve1]; evo3 = slot[OVOFF + ve3];
        }
        double da = 0.0, ll = 0.0, dbf[NV];
        auto fetch_da_ll = [&]() {
            if constexpr (RW == 1) { da = slot[DP]; ll = slot[DP + 1]; }
            else {
                da = 0.0; ll = 0.0;
#pragma unroll
                for (int w = 0; w < RW; ++w) {
                    da += slot[RESO + w * RREC + DP];
                    if constexpr (!TBAR) ll += slot[RESO + w * RREC + DP + 1];
                }
            }
        };
        if constexpr (BKW) {
            da = pf_da; ll = pf_ll;
        }
        else fetch_da_ll();
        if (!lean) { FORV dbf[i] = xtg(jdx[i]); }   // (the shortcut reuses the slot for the next job: fetch first)
        else { FORV dbf[i] = 0.0; }
        if constexpr (RW > 1) {
            // the cavity term of this position, from its own wave (lean: only the view's coordinates of it)
            if (!lean) { FORV { const int e = lane + 64 * i; Ov.v[i] = e < VN ? slot[OVOFF + (e < VN ? e : 0)] : 0.0; } }
            if (fast_ok) { if constexpr (BKW) { vo1 = pf_vo1; vo3 = pf_vo3; } else if constexpr (EARLYT) { vo1 = evo1; vo3 = evo3; } else { vo1 = slot[OVOFF + ve1]; vo3 = slot[OVOFF + ve3]; } }
        }
        if (fast_ok) {
            const double t = BKW ? pf_t : (EARLYT ? et : xtg(lane == LA ? DP : (lane < DP ? lane : 0)));     // lane LA: sum g (as `da` above)
            const double pr2 = laplace ? (double)((vq2 > 0) - (vq2 < 0)) : vq2;
            const double g1 = -vo1 + t, g2 = t * vex3 - pr2, g3 = -vo3 + t * vq2 * vex3;
            // second half of this leapfrog, first half of the next one (the loop top's formulas, element by element)
            if (lean) STAMP(1);          // (diagnostic build, lean iterations: slot 1 = fetching the results)
            const double q1o = vq1, q2o = vq2, q3o = vq3;                     // the position of this leapfrog
            const double fp1 = vp1 + 0.5 * eps_l * g1, fp2 = vp2 + 0.5 * eps_l * g2, fp3 = vp3 + 0.5 * eps_l * g3;
            vp1 = fp1 + 0.5 * eps_l * g1; vp2 = fp2 + 0.5 * eps_l * g2; vp3 = fp3 + 0.5 * eps_l * g3;
            vq1 = vq1 + eps_l * vm1 * vp1; vq2 = vq2 + eps_l * vm2 * vp2; vq3 = vq3 + eps_l * vm3 * vp3;
            vex3 = exp_d_vc(vq3);
            const double ba = vq1 + vq2 * vex3;
            duo_lds_f64 *job = slot + JOB;
            if (lane < DP) job[BOFF + lane] = ((!TEAM && DP > 16) || lane < D) ? ba : 0.0;
            if (lane == LA) job[0] = ba;
            if constexpr (RW > 1) {
                // v = phi - mu of the next position for the row waves' cavity term: the view holds ALL of phi
                // (locations and log scales; models with per-coefficient scales: d = 2 D + 2)
                if (v_lane) { slot[VOFF + ve1] = vq1 - vmu1; slot[VOFF + ve3] = vq3 - vmu3; }
            }
            ++seq;
            if constexpr (TBAR) { team_barrier(); t_job = __builtin_amdgcn_s_memtime(); } else duo_publish_c<BKW>(f_job, seq);
            job_eps = eps_l;
            fast_pub = true;
            if (lean || TEAM) STAMP(2);  // (... slot 2 = the view's update and the job's publication; TEAM: results in -> job out)
            if constexpr (BKW) { ctl_pre = *f_ctl; ack_pre = *f_ack; }     // requested now, used after the chain rule: no round trip then
            __builtin_amdgcn_s_setprio(EPX_PRIO_S_BG);  // the row waves are off again: what follows has their whole pass
            if constexpr (BKW) {
                if (lean) {
                    // ---- the finished state to the bookkeeping wave, from the view (entries of the mailbox in vector order)
                    const double lp1 = -0.5 * (q1o - vmu1) * vo1, lp3 = -0.5 * (q3o - vmu3) * vo3;
                    const double lp2 = laplace ? -fabs(q2o) : -0.5 * q2o * q2o;
                    ++mseq;
                    if (mseq > 2 && __builtin_amdgcn_readfirstlane(ack_pre) < mseq - 2) {
                        const int got = duo_wait_ge(f_ack, mseq - 2);
                        if (got < 0) { bail = 1; break; }
                    }
                    double *m = mbox + (mseq & 1) * MREC;
                    if (v_lane) {
                        m[ve1] = q1o; m[ve2] = q2o; m[ve3] = q3o;
                        m[NV * 64 + ve1] = fp1; m[NV * 64 + ve2] = fp2; m[NV * 64 + ve3] = fp3;
                        m[2 * NV * 64 + ve1] = g1; m[2 * NV * 64 + ve2] = g2; m[2 * NV * 64 + ve3] = g3;
                        m[3 * NV * 64 + ve1] = lp1; m[3 * NV * 64 + ve2] = lp2; m[3 * NV * 64 + ve3] = lp3;
                    }
                    if (lane == 0) { m[4 * NV * 64] = uniform_d(ll); m[4 * NV * 64 + 2] = (double)gen; }
                    duo_publish(f_mail, mseq);
                    STAMP(4);
                    continue;
                }
            }
            if constexpr (TEAM) {
                if (lean) {
                    // ---- the finished state from the view, in vector order for the bookkeeping of the next round:
                    // lane j wrote (location, raw, log scale) of coefficient j; element e of a vector is read back by lane
                    // e % 64.  One scratch line serves the four vectors one after the other (the LDS executes a wave's
                    // accesses in order).  Entries beyond P are never written and stay 0.
                    const double lp1 = -0.5 * (q1o - vmu1) * vo1, lp3 = -0.5 * (q3o - vmu3) * vo3;
                    const double lp2 = laplace ? -fabs(q2o) : -0.5 * q2o * q2o;
                    duo_lds_f64 *scr = duo_lds_at(smem + a.off_scr) + team * a.scr_doubles;
                    if (v_lane) { scr[ve1] = q1o; scr[ve2] = q2o; scr[ve3] = q3o; }
                    relay_step();
                    // (the line holds the P live elements; what lies beyond is 0 by definition)
                    const int scr_n = a.scr_doubles;
#define EPX_SCR(i_) (lane + 64 * (i_) < scr_n ? scr[lane + 64 * (i_) < scr_n ? lane + 64 * (i_) : 0] : 0.0)
                    FORV zq.v[i] = EPX_SCR(i);
                    relay_step();
                    if (v_lane) { scr[ve1] = fp1; scr[ve2] = fp2; scr[ve3] = fp3; }
                    relay_step();
                    FORV zp.v[i] = EPX_SCR(i);
                    relay_step();
                    if (v_lane) { scr[ve1] = g1; scr[ve2] = g2; scr[ve3] = g3; }
                    relay_step();
                    FORV zg.v[i] = EPX_SCR(i);
                    relay_step();
                    if (v_lane) { scr[ve1] = lp1; scr[ve2] = lp2; scr[ve3] = lp3; }
                    relay_step();
                    double lpt = 0.0, ks = 0.0;
                    FORV { lpt += EPX_SCR(i); ks += inv_e.v[i] * zp.v[i] * zp.v[i]; }
                    relay_step();
#undef EPX_SCR
                    f_lpt = lpt; f_ks = ks; f_ll = uniform_d(ll);
                    pending = true;
                    STAMP(4);
                    continue;
                }
            }
        }
        da = uniform_d(da); ll = uniform_d(ll);
        // the parts of the chain rule that only need the position (cross-lane gathers)
        V g_etbq, g_sbj;
        FORV { g_etbq.v[i] = 0.0; g_sbj.v[i] = 0.0; }
        if (model == 2) {
            FORV { const int e = lane + 64 * i; const int j = e <= D ? e - 1 : e - d - 1;
                   g_etbq.v[i] = gatherV(sq, d + 1 + j); g_sbj.v[i] = gatherV(eq, 1 + j); }
        } else if (model >= 3) {
            FORV { const int e = lane + 64 * i; const int j = e < 2 + D ? e - 2 : (e < d ? e - 2 - D : e - d - 1);
                   g_etbq.v[i] = gatherV(sq, d + 1 + j); g_sbj.v[i] = gatherV(eq, 2 + D + j); }
        }
        double lpt = 0.0;
        V lpv;
        {
            double dot = 0.0;
            if (model == 1) {
                double tsum = 0.0;
                FORV { const int e = lane + 64 * i; const double t2 = dbf[i]; if (e >= 3 && e < P) tsum += t2 * sq.v[i]; }
                dot = wave_sum(tsum);
            }
            const double c_da = da, c_sa = da * eta * sa, c_eta = da * sa;
            FORV {
                const int e = lane + 64 * i;
                const double q = sq.v[i];
                const bool in_phi = e < d, in_par = e < P;
                const double ov = Ov.v[i];
                double g = in_phi ? -ov : 0.0;
                const double lp_phi = -0.5 * vv.v[i] * ov;
                const double lp_pri = laplace ? -fabs(q) : -0.5 * q * q;
                const double lterm = in_phi ? lp_phi : (in_par ? lp_pri : 0.0);
                lpt += lterm;
                lpv.v[i] = lterm;
                const double pr = laplace ? (double)((q > 0) - (q < 0)) : q;    // d/dq of the N(0,1)/Laplace term
                const double g_eta = c_eta - pr;
                double add = 0.0, g_etb = 0.0;
                const double db = dbf[i];
                if (model == 0) {
                    add = (e >= 1 && e <= D) ? db : add;
                    add = e == 0 ? c_sa : add;
                } else if (model == 1) {
                    g_etb = db * sb2 - pr;
                    add = e == 1 ? dot * sb2 : add;
                    add = e == 0 ? c_sa : add;
                } else if (model == 2) {
                    g_etb = db * g_sbj.v[i] - pr;
                    add = (e >= 1 && e <= D) ? db * g_etbq.v[i] * eq.v[i] : add;
                    add = e == 0 ? c_sa : add;
                } else {
                    g_etb = db * g_sbj.v[i] - pr;
                    add = (e >= 2 + D && in_phi) ? db * g_etbq.v[i] * eq.v[i] : add;
                    add = (e >= 2 && e < 2 + D) ? db : add;
                    add = e == 1 ? c_sa : add;
                    add = e == 0 ? c_da : add;
                }
                g = in_phi ? g + add : g;
                g = e == d ? g_eta : g;
                g = (e > d && in_par) ? g_etb : g;
                sg.v[i] = in_par ? g : 0.0;
            }
        }
        double ks = 0.0;
        FORV { sp.v[i] += 0.5 * eps_l * sg.v[i]; ks += inv_e.v[i] * sp.v[i] * sp.v[i]; }
        // the trajectory continues from here unless the bookkeeping (next round, beside the next
        // row pass) says otherwise
        FORV { zq.v[i] = sq.v[i]; zp.v[i] = sp.v[i]; zg.v[i] = sg.v[i]; }
        f_lpt = lpt; f_ks = ks; f_ll = ll;
        pending = true;
        if constexpr (BKW) {
            // ---- hand the finished state to the bookkeeping wave (it is at most two states behind)
            STAMP(4);
            ++mseq;
            if (mseq > 2 && __builtin_amdgcn_readfirstlane(ack_pre) < mseq - 2) {
                const int got = duo_wait_ge(f_ack, mseq - 2);
                if (got < 0) { bail = 1; break; }
            }
            double *m = mbox + (mseq & 1) * MREC;
            FORV {
                m[(0 * NV + i) * 64 + lane] = sq.v[i]; m[(1 * NV + i) * 64 + lane] = sp.v[i];
                m[(2 * NV + i) * 64 + lane] = sg.v[i];
            }
            FORV m[(3 * NV + i) * 64 + lane] = lpv.v[i];
            if (lane == 0) { m[4 * NV * 64] = ll; m[4 * NV * 64 + 2] = (double)gen; }
            duo_publish(f_mail, mseq);
            STAMP(1);                               // (diagnostic build: the hand-over is booked on the bookkeeping slot)
        }
        STAMP(4);
    }
#ifdef EPX_STAMPS
    if (a.stamps && team == 0 && lane == 0 && is_state) {
        for (int i = 0; i < 5; ++i) a.stamps[(size_t)blockIdx.x * 8 + i] += tacc[i];
        a.stamps[(size_t)blockIdx.x * 8 + 7] += (unsigned long long)seq;
    }
#endif
    if constexpr (BKW) {
        if (is_state) {
            // the state wave is done when the bookkeeping wave says so (or a hand-off failed): release the others
            *f_job = DUO_EXIT;
            if (bail & 1) {
                if (lane == 0) atomicOr(a.err, 2);
                *f_mail = DUO_EXIT;
            }
            return;                                   // [parity] BKW exists for the polled layouts only (static_assert below)
        }
    }

    // ------------------------------------------------------------- epilogue (the state wave owns the chain)
    if constexpr (!BKW) *f_job = DUO_EXIT;             // the row waves leave
    team_leave(true);                                  // [parity] every later `return` of this state wave is behind the last B1
    if (bail & 1) {
        if (lane == 0) atomicOr(a.err, 2);
        failed = 2;
    }
    if (bail & 4) return;                       // test hook (a.dbg): lp and gradient are written
    if constexpr (COLD) {
        if (segmented) {
            // ---- checkpoint at the transition boundary: the sample, the Welford sums, the metric and the scalars of
            // EPX_CK_LIST (the gradient at the sample is re-evaluated by the piece that continues)
            double *ckp = ck_rec(t_end);
            FORV {
                ck_store(ckp + (0 * NV + i) * 64 + lane, qs.v[i]);
                ck_store(ckp + (1 * NV + i) * 64 + lane, wmean.v[i]);
                ck_store(ckp + (2 * NV + i) * 64 + lane, wm2.v[i]);
                ck_store(ckp + (3 * NV + i) * 64 + lane, inv_e.v[i]);
            }
            double ckv = 0.0;
#define EPX_CK_PUT(idx, x) ckv = lane == (idx) ? (double)(x) : ckv;
            EPX_CK_LIST(EPX_CK_PUT)
#undef EPX_CK_PUT
            ck_store(ckp + 4 * NV * 64 + lane, ckv);
            piece_checkpoint_out();                                 // the record is out before the site is put back
        }
    }
    if (!failed && t < a.iter) return;          // suspended at the end of a piece: no final record yet
    {
        double *lastp = a.last + ((size_t)k * a.chains + chain) * P;
        FORV { const int e = lane + 64 * i; if (e < P) lastp[e] = qs.v[i]; }
        if (failed) {
            for (int kk = 0; kk < a.nkeep; ++kk) {
                double *dst = a.draws + (((size_t)k * a.chains + chain) * a.nkeep + kk) * P;
                FORV { const int e = lane + 64 * i; if (e < P) dst[e] = qs.v[i]; }
            }
        }
        if (lane == 0) {
            double *st = a.chain_stats + ((size_t)k * a.chains + chain) * ST_COUNT;
            st[ST_STEPSIZE_MEAN] = a.iter > 0 && !failed ? eps_sum / a.iter : 0.0;
            st[ST_STEPSIZE_FINAL] = eps;
            st[ST_NLEAP] = nleap_tot;
            st[ST_NGRAD] = ngrad;
            st[ST_NDIV] = ndiv;
            st[ST_ACCEPT_MEAN] = npost ? acc_sum / npost : 0.0;
            st[ST_DEPTH_MEAN] = npost ? depth_sum / npost : 0.0;
            st[ST_FAIL] = failed;
        }
    }
}
#undef a

template <int NV, int DP, int CPB, int RW, bool STL, bool COLD, bool PIECED>
__global__ void __launch_bounds__(64 * (CPB == 4 && RW == 4 ? 8 : CPB * (1 + RW) + (CPB == 1 ? 2 : 0)))
k_nuts_duo(NutsArgs a_by_value) {
    extern __shared__ __align__(16) unsigned char smem[];
    (void)a_by_value;
    DuoArgsK *kargs_p = (DuoArgsK *)__builtin_amdgcn_kernarg_segment_ptr();
    DuoArgsK &a_piece = *kargs_p;
#define a a_piece
    const int tid = threadIdx.x;
    if constexpr (!PIECED) {
        duo_piece<NV, DP, CPB, RW, STL, COLD, false>(kargs_p, tid, false, -1, 0);
        return;
    }
    // Piece queue (NutsArgs::dyn_prog; what Master uses at the C3 site size).  The launch has one workgroup per PIECE
    // (sites x pieces per site); the hardware's dispatcher is the loop: a workgroup claims the site with the largest
    // predicted REMAINING work (transitions left x predicted leapfrogs per transition) among the sites nobody holds,
    // runs dyn_len transitions of it, puts it back and ends -- longest remaining processing time first, the
    // preemptive schedule that ends all sites at about the same time, and it adapts to what the sites really cost.
    // This is the form WITHOUT a loop over pieces (NutsArgs::persist == 0; diagnostic builds and A/B): k_nuts_duo_loop
    // below is what a pieced launch runs by default.
    // A claim is a compare-and-swap on the site's word.  There are exactly as many workgroups as pieces, so a
    // workgroup that finds every unfinished site held waits for one to come back; the holders never wait.
    int q_site, q_t0;
    if (piece_claim(a, smem, tid, q_site, q_t0) <= 0) {            // (no site for EPX_PIECE_WAIT_S seconds: reported, never seen)
        if (tid == 0) atomicOr(a.err, 4);
        return;
    }
    duo_piece<NV, DP, CPB, RW, STL, COLD, true>(kargs_p, tid, true, q_site, q_t0);
    __syncthreads();
    if (threadIdx.x == 0) piece_release(a, smem);    // the checkpoint records are out: the site goes back to the pool
}

#undef a

// The pieced launch with LOOPING workgroups (NutsArgs::persist, the default): as many workgroups as the device holds at
// a time, each claiming pieces until no site has anything left.  One workgroup per piece leaves CU-time to the in-order
// dispatcher (epx_pieces.h; scripts/probe/dispatch_gaps.hip): on one box 527.5 -> 534.0 site-updates/s at C3, whose
// pieces last about equally long, and 15.0 -> 15.8 at the C5 shard (nuts_stream.hip), whose pieces do not.
// The piece's body is a real CALL.  Inlined into the loop (round 2) it lost its register allocation: everything is then
// live around a loop that contains both roles' inner loops, 68 -> 500 B of scratch per lane and 20 % of the time.  As a
// function it is compiled like the kernel without the loop -- provided its arguments are made scalar again: arguments of
// a call travel in vector registers and count as divergent, so a pointer to the kernel arguments passed as it is turns
// every a.field into a vector load (876 B of scratch, 13 % slower), and __builtin_amdgcn_kernarg_segment_ptr() is null
// inside a called function.
template <int NV, int DP, int CPB, int RW, bool STL, bool COLD>
__device__ __attribute__((noinline)) void duo_piece_call(unsigned long long kargs_u, int q_site, int q_t0) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)kargs_u), hi = __builtin_amdgcn_readfirstlane((unsigned)(kargs_u >> 32));
    DuoArgsK *kargs_p = (DuoArgsK *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    duo_piece<NV, DP, CPB, RW, STL, COLD, true>(kargs_p, (int)threadIdx.x, true, __builtin_amdgcn_readfirstlane(q_site), __builtin_amdgcn_readfirstlane(q_t0));
}

template <int NV, int DP, int CPB, int RW, bool STL, bool COLD>
__global__ void __launch_bounds__(64 * (CPB == 4 && RW == 4 ? 8 : CPB * (1 + RW) + (CPB == 1 ? 2 : 0)))
k_nuts_duo_loop(NutsArgs a_by_value) {
    extern __shared__ __align__(16) unsigned char smem[];
    (void)a_by_value;
    DuoArgsK *kargs_p = (DuoArgsK *)__builtin_amdgcn_kernarg_segment_ptr();
    for (;;) {
        int q_site, q_t0;
        const int got = piece_claim(*kargs_p, smem, (int)threadIdx.x, q_site, q_t0);
        if (got <= 0) {
            if (got < 0 && threadIdx.x == 0) atomicOr(kargs_p->err, 4);
            return;
        }
        duo_piece_call<NV, DP, CPB, RW, STL, COLD>((unsigned long long)(uintptr_t)kargs_p, q_site, q_t0);
        __syncthreads();
        if (threadIdx.x == 0) piece_release(*kargs_p, smem);
        __syncthreads();                                 // (the next claim's scratch is the LDS this piece used)
    }
}

// ---------------------------------------------------------------------------
// host side: LDS layout + dispatch over the instantiated shapes
size_t nuts_duo_lds_layout(NutsArgs &a, int cpb, int rw, int dp, int n_max) {
    const int nv = (a.P + 63) / 64;
    const int ou = nv > 1 ? 4 : 8;                                                     // as the kernel (OU)
    const int d = a.d, dm = d < 64 ? d : 64, npad = ((dm + 1) / 2 + ou - 1) / ou * ou;
    const bool teamm = cpb == 4 && rw == 4;                                            // as the kernel (TEAM): whole 16-row tiles, no cavity precision in LDS
    size_t off = (size_t)(teamm ? team_rows(n_max) : n_max) * dp * 8;
    a.n_max = n_max; a.duo_rw = rw; a.cpb = cpb;
    a.off_Om = (int)off; off += teamm ? 0 : (size_t)npad * dm * 16;
    a.off_tail = (int)off; off += nv > 1 && !teamm ? (size_t)2 * (2 * npad + 2) * 8 : 0;
    off = (off + 15) & ~(size_t)15;
    const int vn = (teamm && 2 * dp + 8 < nv * 64) ? 2 * dp + 8 : nv * 64;                           // as the kernel (VN)
    a.slot_doubles = rw == 1 ? dp + 2 : (dp + 2) + vn + rw * (dp + 2) + vn;                         // as the kernel (VOFF, RESO, OVOFF)
    a.off_slot = (int)off; off += (size_t)cpb * a.slot_doubles * 8;
    const bool bkw = cpb == 1;                                                        // as the kernel (BKW)
    a.off_flag = (int)off; off += teamm ? 48 : (size_t)cpb * (1 + rw + (bkw ? 4 : 0)) * 4;
    off = (off + 15) & ~(size_t)15;
    a.off_spec = 0;
    if (bkw) { a.off_spec = (int)off; off += (size_t)2 * ((4 * nv * 64 + 4) + (4 * nv * 64 + 4)) * 8; }
    a.om_in_lds = 1;
    const size_t cap = 160 * 1024;
    const size_t stack = (size_t)cpb * a.max_depth * nuts_stack_record(nv) * 8;
    a.off_scr = 0;
    a.scr_doubles = (a.P + 7) & ~7;
    if (teamm) { a.off_scr = (int)off; off += (size_t)cpb * a.scr_doubles * 8; }
    a.stack_in_lds = 0; a.off_stack = (int)off; a.stack_lds_levels = 0;
    if (off + stack <= cap) { a.stack_in_lds = 1; off += stack; }
    else if (teamm) {
        // (TEAM: the cavity precision is in registers, so the LDS has room for the lowest stack levels)
        a.stack_ps = (a.P + 1) & ~1;                  // a packed vector: only the P live elements (the accesses beyond are masked)
        const size_t per_level = (size_t)cpb * 2 * a.stack_ps * 8;
        int lv = (int)((cap - 16 - off) / per_level);
        if (lv > 4) lv = 4;
        if (lv > a.max_depth) lv = a.max_depth;
        if (lv > 0) { a.stack_lds_levels = lv; off += (size_t)lv * per_level; }
    }
    a.off_piece = (int)off; off += 16;
    a.lds_bytes = (int)off;
    return off;
}

// doubles of global memory per chain of the resident layouts: tree stack + cold store (NutsArgs::stack)
size_t nuts_resident_chain_doubles(int nv, int max_depth) {
    return (size_t)max_depth * nuts_stack_record(nv) + (size_t)GV_COUNT * nv * 64;
}

template <int NV, int DP, int CPB, int RW>
static int launch_duo_one(const NutsArgs &a, int nblocks, hipStream_t stream) {
    auto go = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
        if (e != hipSuccess) return (int)e;
        constexpr int NT = 64 * (CPB == 4 && RW == 4 ? 8 : CPB * (1 + RW) + (CPB == 1 ? 2 : 0));
        const int nb = nblocks;
        (void)NT;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(NT), a.lds_bytes, stream, a);
        return (int)hipGetLastError();
    };
    constexpr bool COLD = NV >= 2 || CPB > 1;
    if constexpr (COLD && CPB > 1) {
        if (a.dyn_prog && a.persist) {
            // looping workgroups: as many as the device holds at a time (never more than there are pieces)
            auto loop = [&](auto kern) -> int {
                constexpr int NT = 64 * (CPB == 4 && RW == 4 ? 8 : CPB * (1 + RW));
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, a.lds_bytes);
                if (e != hipSuccess) return (int)e;
                int per_cu = 0, dev = 0, ncu = 0;
                e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), NT, (size_t)a.lds_bytes);
                if (e != hipSuccess) return (int)e;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
                const int hold = (per_cu > 0 ? (per_cu < 8 ? per_cu : 8) : 1) * (ncu > 0 ? ncu : 1);      // (the host sized the workgroups' private memory for at most 8 per CU)
                hipLaunchKernelGGL(kern, dim3(nblocks < hold ? nblocks : hold), dim3(NT), a.lds_bytes, stream, a);
                return (int)hipGetLastError();
            };
            return a.stack_in_lds ? loop(k_nuts_duo_loop<NV, DP, CPB, RW, true, COLD>) : loop(k_nuts_duo_loop<NV, DP, CPB, RW, false, COLD>);
        }
    }
    if constexpr (COLD && CPB > 1) {
        if (a.dyn_prog)
            return a.stack_in_lds ? go(k_nuts_duo<NV, DP, CPB, RW, true, COLD, true>) : go(k_nuts_duo<NV, DP, CPB, RW, false, COLD, true>);
    }
    return a.stack_in_lds ? go(k_nuts_duo<NV, DP, CPB, RW, true, COLD, false>) : go(k_nuts_duo<NV, DP, CPB, RW, false, COLD, false>);
}

template <int NV, int DP>
static int launch_duo_shape(const NutsArgs &a, int nblocks, int cpb, int rw, hipStream_t stream) {
    if (cpb == 4 && rw == 1) return launch_duo_one<NV, DP, 4, 1>(a, nblocks, stream);
    if (cpb == 4 && rw == 4) return launch_duo_one<NV, DP, 4, 4>(a, nblocks, stream);
    if (cpb == 1 && rw == 2) return launch_duo_one<NV, DP, 1, 2>(a, nblocks, stream);
    return -1;
}

bool nuts_duo_has(int cpb, int rw, int dp, int nv) {
    return ((cpb == 4 && (rw == 1 || rw == 4)) || (cpb == 1 && rw == 2)) && (dp == 16 || dp == 32) && (nv == 1 || nv == 2);
}

int launch_nuts_duo(const NutsArgs &a, int count, int cpb, int rw, int dp, int nv, hipStream_t stream) {
    if (!nuts_duo_has(cpb, rw, dp, nv)) return -1;
    const int nblocks = a.dyn_prog ? a.seg_nwg : count * ((a.chains + cpb - 1) / cpb);
    if (nv == 1) return dp == 16 ? launch_duo_shape<1, 16>(a, nblocks, cpb, rw, stream) : launch_duo_shape<1, 32>(a, nblocks, cpb, rw, stream);
    return dp == 16 ? launch_duo_shape<2, 16>(a, nblocks, cpb, rw, stream) : launch_duo_shape<2, 32>(a, nblocks, cpb, rw, stream);
}

}  // namespace epx
