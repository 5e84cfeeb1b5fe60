// Pieced sampler launches (epx_set_piece_queue): what the resident (nuts_duo.hip) and the streaming
// (nuts_stream.hip) kernels share -- the claim of a site, its release, and the checkpoint record's accessors.
//
// The launch has as many workgroups as the device holds at a time (NutsArgs::persist; at most one per piece), and a
// workgroup LOOPS over claims until no site has anything left.  (The first form had one workgroup per piece, "the
// hardware's dispatcher is the loop": workgroups are dealt to the 8 XCDs by blockIdx % 8 and IN ORDER, so when the next
// one's XCD has no CU free the free CUs of the other XCDs wait -- with pieces of unequal length 16-21 % of the CU-time,
// 3 % for looping workgroups: scripts/probe/dispatch_gaps.hip, profiles/r03_dispatch_gaps_probe.txt.)
// A workgroup claims -- compare-and-swap on the site's `busy` word -- the site with the largest predicted REMAINING
// work (transitions left x predicted leapfrogs per transition) among the sites nobody holds, runs its next piece from
// the checkpoint the piece before left, writes its own checkpoint, puts the site back and ends: longest remaining
// processing time first, the preemptive schedule that ends all sites at about the same time, and it adapts to what
// the sites really cost.  A workgroup that finds every unfinished site held waits for one to come back; the holders
// never wait, and every workgroup of the launch is resident, so nobody waits for a workgroup that has not started.
#pragma once
#include <hip/hip_runtime.h>

#ifndef EPX_PIECE_WAIT_S
#define EPX_PIECE_WAIT_S 60
#endif

namespace epx {

// what a chain carries from one piece to the next besides the sample, the Welford sums and the metric (lane, variable)
#define EPX_CK_LIST(X)                                                                                        \
    X(0, lps) X(1, eps) X(2, da_mu) X(3, s_bar) X(4, x_bar) X(5, da_count) X(6, va_n) X(7, eps_sum) X(8, acc_sum)  \
    X(9, depth_sum) X(10, nleap_tot) X(11, ngrad) X(12, t) X(13, va_counter) X(14, va_wsize) X(15, va_next)       \
    X(16, ndiv) X(17, npost) X(18, kept) X(19, failed)

// A checkpoint travels between CUs of DIFFERENT XCDs, whose L2s do not see each other's lines.  The protocol (and what
// each step leans on):
//   writer   1. every lane stores its part of the record with SYSTEM-scope atomic stores (ck_store: `global_store_dwordx2
//               ... sc0 sc1`).  On gfx942 / gfx950 the sc bits of a store ARE its coherence scope (LLVM AMDGPUUsage, "Memory
//               Model gfx942": store atomic monotonic, system scope = sc0=1 sc1=1): the XCD's L2 does not keep the line
//               dirty for a store of agent or system scope, it forwards the write to the fabric -- that is how a monotonic
//               atomic store becomes visible to another XCD without any fence;
//            2. every wave that stored waits `s_waitcnt vmcnt(0)` (piece_checkpoint_out).  vmcnt counts a store until it is
//               ACKNOWLEDGED, and a written-through store is acknowledged when the write has left the L2 -- the same
//               guarantee the model's own release sequence leans on (`buffer_wbl2 sc1` followed by `s_waitcnt vmcnt(0)`);
//            3. a workgroup barrier collects the waves (k_nuts_duo_loop / k_nuts_stream_loop: __syncthreads() behind the piece);
//            4. thread 0 stores the site's word, relaxed at agent scope (`global_store_dword ... sc1`, piece_release).
//   claimer  5. compare-and-swap on the site's word with ACQUIRE at agent scope, then an agent-scope acquire FENCE
//               (piece_claim: `buffer_inv sc1`): this XCD's L2 drops what it holds, so no older line of a record can be hit;
//            6. the record is read with system-scope atomic loads (ck_load: sc0 sc1, served past the L2).
// What is NOT used between 2 and 4 is the model's release FENCE: on these parts it is `buffer_wbl2 sc1`, a write-back of
// EVERY dirty line of the XCD's L2 -- the tree stacks and cold stores of all 32 workgroups of the XCD, which nobody else
// ever reads (6 of the 18 GB a C3 launch wrote to HBM, section 6 of HISTORY.md).  Steps 1-2 are a release of exactly the
// lines that travel; the form is the one /opt/skills/guides/MI355X_MICROARCH.md lists under "Valid forms" (`sc0 sc1`
// stores drained by every storing wave's vmcnt(0), a workgroup barrier, one lane's flag store; the consumer's acquire kept),
// measured there on gfx950 / ROCm 7.2 and marked "not an architectural guarantee".  This is an argument from the ISA's behaviour, not from the language's memory model (for which a
// relaxed store orders nothing): it is therefore (a) confined to this header, (b) switchable -- -DEPX_PIECE_FENCE puts the
// release fence and a release store back, build.sh ships that build as variants/libepx_fence.so -- and (c) tested on the
// device by a litmus run (tests/test_gpu_round4.py::test_piece_handoff_litmus_*: hundreds of sites in pieces of ONE
// transition, so that every site changes XCD dozens of times per launch, repeated, bit-equal to the uncut launch, under
// both builds).  Every piece boundary has its own record (piece_record), so no address is ever written twice in a launch.
__device__ inline void ck_store(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline double ck_load(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_SYSTEM));
}
template <typename T> __device__ inline void ck_assign(T &x, double v) { x = (T)v; }

// doubles of one chain's checkpoint record: sample, Welford mean and sum of squares, metric (nv x 64 each), scalars (64)
__host__ __device__ constexpr size_t piece_record_doubles(int nv) { return (size_t)(4 * nv + 1) * 64; }
// Transitions of the piece of `site` that starts at transition t0.  TWO lengths per site (round 5): the nominal one, L,
// for the first three quarters of the run, a quarter of it (at least 1) behind transition T1 = the last multiple of L at or
// below 3/4 of the run.  A pieced launch ends with the LAST piece of every workgroup, so on average half a piece of every
// CU is idle at the end (at C5 a nominal piece is ~1 s of a 31 s launch): the short pieces are handed out when the queue
// runs dry, and the claim by largest remaining work then evens the workgroups out four times finer.  (Nominal lengths are
// per site when the host gives them -- NutsArgs::dyn_lens -- else one for all.)
template <class Args>
__device__ __forceinline__ int piece_len_of(Args &a, int site) { return a.dyn_lens ? a.dyn_lens[site] : a.dyn_len; }
// (div: NutsArgs::dyn_tail_div -- 4 by default; 1 = one length throughout, the form of rounds 2-4: EPX_PIECE_TAIL_DIV for A/B)
__host__ __device__ inline int piece_short_len(int len, int div) { return len >= div ? len / div : 1; }
__host__ __device__ inline int piece_switch_at(int iter, int len) { return (iter - iter / 4) / len * len; }       // T1
__host__ __device__ inline int piece_boundaries(int iter, int len, int div) {        // boundaries behind the start: pieces of a site
    const int t1 = piece_switch_at(iter, len), ls = piece_short_len(len, div);
    return t1 / len + (iter - t1 + ls - 1) / ls;
}
template <class Args>
__device__ __forceinline__ int piece_len_at(Args &a, int site, int t0) {
    const int len = piece_len_of(a, site);
    return t0 < piece_switch_at(a.iter, len) ? len : piece_short_len(len, a.dyn_tail_div);
}
// Every piece BOUNDARY of a site has its own record: an address is written once per launch, by one workgroup, and read
// once, by another -- no XCD can hold an older version of it (the first form had one record per site, and a site that
// came back to an XCD it had been on could read a mix of that L2's older lines and fresh ones: a rare wrong trajectory,
// found by the EP parity test).  Boundary t_boundary is number t / L up to T1 and T1 / L + ceil((t - T1) / Ls) behind it
// (the last boundary is a.iter, whatever the lengths).
template <class Args>
__device__ __forceinline__ double *piece_record(Args &a, int site, int t_boundary, int chain, int nv) {
    const int len = piece_len_of(a, site);
    const int t1 = piece_switch_at(a.iter, len), ls = piece_short_len(len, a.dyn_tail_div);
    const int b = t_boundary <= t1 ? t_boundary / len : t1 / len + (t_boundary - t1 + ls - 1) / ls;
    return a.ckpt + (((size_t)site * a.dyn_nb + b) * a.chains + chain) * piece_record_doubles(nv);
}

// Claim a site (all threads of the workgroup; the LDS must not hold anything yet: smem[0..1100) is scratch, and
// (site, first transition) stay at smem + off_piece for piece_release).  Returns 1 with a site, 0 when every site of the
// launch has run all its transitions (a persistent workgroup's way out), -1 after EPX_PIECE_WAIT_S seconds without a site
// (never seen on a device of its own; the caller reports it and the host call fails with an error.  The limit is wall time:
// when several PROCESSES share one device -- the multi-rank tests on a one-GPU box -- the driver may park a workgroup that
// holds a site for as long as the other process's launch lasts, tens of seconds; such runs raise it, EPX_PIECE_WAIT_S in
// the environment -> NutsArgs::dyn_wait_s).
template <class Args>
__device__ __forceinline__ int piece_claim(Args &a, unsigned char *smem, int tid, int &q_site, int &q_t0) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    volatile double *sc = reinterpret_cast<volatile double *>(smem);
    volatile int *si = reinterpret_cast<volatile int *>(smem + 1024);
    q_site = -1; q_t0 = 0;
    // (bounded by wall time, not by a count of looks: s_memrealtime ticks at 100 MHz; a piece takes milliseconds to a few
    // seconds, so a workgroup that has found nothing for EPX_PIECE_WAIT_S seconds reports a lost piece instead of spinning on)
    const unsigned long long t_claim0 = __builtin_amdgcn_s_memrealtime();
    for (int attempt = 0; q_site < 0; ++attempt) {
        double best = -1.0; int arg = -1, unfinished = 0;
        for (int s = tid; s < a.dyn_count; s += blockDim.x) {
            // ONE word per site: 2 x (transitions done) + (held): progress and claim change together, atomically
            const int wd = __hip_atomic_load(a.dyn_prog + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int pr = wd >> 1, bz = wd & 1;
            unfinished |= pr < a.iter;
            if (bz == 0 && pr < a.iter) {
                // (a +-12 % jitter per (workgroup, site, attempt): 256 workgroups that all went for THE largest
                // remaining site would get it one at a time)
                unsigned hsh = (unsigned)s * 2654435761u ^ ((unsigned)blockIdx.x * 40503u + (unsigned)attempt * 97u) * 2246822519u;
                hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
                const double jit = 0.88 + 0.24 * (double)(hsh & 0xFFFF) * (1.0 / 65536.0);
                const double sc_s = (double)(a.iter - pr) * (a.dyn_rate ? a.dyn_rate[s] : 1.0) * jit;
                if (sc_s > best) { best = sc_s; arg = s; }
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64); const int oa = __shfl_xor(arg, off, 64);
            if (ob > best || (ob == best && oa >= 0 && (arg < 0 || oa < arg))) { best = ob; arg = oa; }
        }
        const int unf_w = __builtin_amdgcn_ballot_w64(unfinished != 0) != 0;
        if (lane == 0) { sc[wave] = best; si[wave] = arg; si[16 + wave] = unf_w; }
        __syncthreads();
        if (tid == 0) {
            double b = -1.0; int g = -1, unf = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { unf |= si[16 + w]; if (sc[w] > b) { b = sc[w]; g = si[w]; } }
            int got = unf ? -2 : -4;                       // -2: every unfinished site is held right now; -4: no site has anything left
            int t0_got = 0;
            if (g >= 0) {
                int expect = __hip_atomic_load(a.dyn_prog + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = (expect & 1) == 0 && (expect >> 1) < a.iter;
                if (ok) ok = __hip_atomic_compare_exchange_strong(a.dyn_prog + g, &expect, expect + 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT);
                got = ok ? g : -1;                         // -1: somebody was faster, look again
                t0_got = expect >> 1;                      // (the progress that was claimed: same word, same instant)
            }
            // (thread 0 decides for the workgroup when to give up: every thread must leave the loop in the same round)
            if (got < 0 && __builtin_amdgcn_s_memrealtime() - t_claim0 > (unsigned long long)(a.dyn_wait_s > 0 ? a.dyn_wait_s : EPX_PIECE_WAIT_S) * 100000000ull) got = -3;
            si[32] = got;
            if (got >= 0) {
                si[33] = t0_got;
                volatile int *pz = reinterpret_cast<volatile int *>(smem + a.off_piece);      // (kept for the release)
                pz[0] = got; pz[1] = si[33];
            }
        }
        __syncthreads();
        const int got = __builtin_amdgcn_readfirstlane(si[32]);         // (wave-uniform for the compiler, too)
        if (got >= 0) { q_site = got; q_t0 = __builtin_amdgcn_readfirstlane(si[33]); }
        __syncthreads();
        if (got == -3 || got == -4) { q_site = got; break; }
        if (got == -2) {
            // every unfinished site is held: a piece takes tens of milliseconds, so look again in ~0.2 ms (hundreds of
            // waiting workgroups polling the site words at full speed would be felt by the pieces that still run)
            for (int z = 0; z < 64; ++z) __builtin_amdgcn_s_sleep(127);
        }
    }
    // acquire at agent scope: this XCD's L2 may hold an older record of the site (from a piece that ran here earlier)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return q_site >= 0 ? 1 : (q_site == -4 ? 0 : -1);
}

// After a chain's checkpoint stores (every wave that wrote one): they are complete before anybody is told.
// The record is written with system-scope stores (ck_store: write-through, acknowledged when the write is out of this
// XCD's L2), so waiting for them IS the release.  The release FENCE that stood here writes back every dirty line of
// the XCD's L2 (`buffer_wbl2`) -- the tree stacks and cold stores of all 32 workgroups of the XCD, ~2 MB, at every one
// of the ~8 700 piece ends of a C3 launch: 15 of the 18 GB a launch wrote to HBM, for lines nobody else ever reads.
// EPX_PIECE_FENCE (compile time) brings the fences back.
__device__ __forceinline__ void piece_checkpoint_out() {
#ifdef EPX_PIECE_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// The site goes back to the pool (thread 0, after a workgroup barrier behind the chains' checkpoint stores and their
// s_waitcnt vmcnt(0)): progress first, then the claim word
template <class Args>
__device__ __forceinline__ void piece_release(Args &a, unsigned char *smem) {
    volatile int *pz = reinterpret_cast<volatile int *>(smem + a.off_piece);
    const int r_site = pz[0], r_t0 = pz[1];
    const int r_len = piece_len_at(a, r_site, r_t0);
    // (epx_sample_piece, dyn_hook: ONE transition per site -- the site goes back as finished, so a workgroup that starts
    // late, with more sites than the device holds workgroups, cannot take it for a second transition and leave another
    // site untouched; the record still lands at boundary t0 + 1, piece_record is keyed by the transition, not by this word)
    const int t1 = a.dyn_hook ? a.iter : (r_t0 + r_len < a.iter ? r_t0 + r_len : a.iter);
    // progress up, claim off: one store (agent scope: written through to where the other XCDs' claims read it; the
    // checkpoint stores of every wave are complete -- piece_checkpoint_out, then the workgroup barrier in front of this)
#ifdef EPX_PIECE_FENCE
    __hip_atomic_store(a.dyn_prog + r_site, 2 * t1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    __hip_atomic_store(a.dyn_prog + r_site, 2 * t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

}  // namespace epx
