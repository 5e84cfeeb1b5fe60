// Row streaming engine of the streaming sampler (nuts_stream.hip): one pass over a site's
// rows per leapfrog, shared by the (up to 4) chains of the workgroup.  HBM-bound by design:
// algorithmic traffic per pass = n*D*8 (X) + n*4 (y) bytes for 4 gradients.
//
//   * X travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging) into a ring
//     of NSL slots of 16 rows; three tiles stay in flight across the raw s_barriers (counted
//     s_waitcnt vmcnt, never 0 inside a pass), and the ring does not stop at leapfrog
//     boundaries: the first three tiles of the next pass are in flight while the tree
//     bookkeeping runs.
//   * One DMA instruction writes 1 KiB of LDS linearly (wave-uniform base + lane*16), so the
//     image of a row is unpadded; bank conflicts are avoided by permuting the 16-byte chunks
//     of row r on the SOURCE side (slot s holds chunk s ^ r).
//   * The two skinny products of the gradient, F = X B (n x D by D x 4 chains) and
//     G = X' g (D x n by n x 4), run on v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4: 16 rows or
//     columns x 4 chains per instruction, every lane useful), with B in registers: per 16-row
//     tile a wave issues DPB/8 + 4 ds_read_b64 and DPB/8 MFMA instead of several hundred
//     LDS-fed FMAs.  The matrix pipe is nowhere near saturated; the point is issue slots.
//   * Six waves with fixed roles and a software pipeline over tiles with ONE barrier per tile:
//     in phase p the four chain waves do the forward product of tile p (wave = column
//     quarter) and the backward product of tile p-2, wave 4 (loader) waits for tile p and
//     issues the DMA of tile p+3, wave 5 evaluates the logistic terms of tile p-1 for all 4
//     chains (lane = (row, chain)).
//   * stream_pass is a real function call (noinline): inlined into the sampler kernel its
//     loops inherit that kernel's register pressure and the compiler parks loop invariants in
//     scratch -- a scratch reload is a VMEM operation, and one inside the loader's loop drains
//     the whole DMA ring.  As a callee it is allocated on its own (about 100 VGPRs, no
//     scratch).  All LDS accesses go through address_space(3) pointers built from byte
//     offsets (generic pointers would turn into flat loads across the call boundary).
// Rows beyond the site (last tile) are clamped to the last row and masked in the logistic
// step; columns beyond D read whatever follows in memory (finite: the engine pads X with a
// zeroed KiB) and meet zero coefficients.
//
// v_mfma_f64_4x4x4f64 operand layout (measured, scripts/probe/mfma_layout.hip):
//   A[b][i][k] in lane 16k + 4b + i,  B[b][k][j] in lane 16k + 4b + j,  D[b][i][j] in lane 16i + 4b + j.
#pragma once
#include "epx_device.h"

namespace epx {

constexpr int TR = 16;          // rows per ring slot
constexpr int NSL = 6;          // ring slots: 3 resident (backward, logistic, forward) + 3 in flight
#ifndef EPX_RING_AHEAD
#define EPX_RING_AHEAD 3      // tiles in flight (diagnostic: 2 shows how far the launch follows the ring's depth; 3 x 17 DMA pieces is what one wave's vmcnt holds)
#endif
constexpr int NCH = 4;          // chain slots (waves 0..3) per workgroup
constexpr int STREAM_WAVES = 6; // + loader (wave 4) + logistic (wave 5)
constexpr int STREAM_THREADS = 64 * STREAM_WAVES;

template <int DPB> struct StreamGeom {
    static constexpr int CPR = DPB / 2;            // 16-byte chunks per row image
    static constexpr int ROWB = DPB * 8;           // bytes per row image
    static constexpr int RPI = 64 / CPR;           // rows per DMA instruction (1 KiB)
    static constexpr int NI = TR / RPI;            // DMA instructions per tile
    static constexpr int SLOTB = TR * ROWB;        // bytes per ring slot
    static constexpr int G = NI + 1;               // VMEM operations of the loader wave per tile (+ the y piece)
    static constexpr int DW = DPB / 4;             // columns per wave
    static constexpr int KS = DW / 4;              // forward k-steps per wave
    static constexpr int MB = DW / 16;             // backward 16-column groups per wave
};

// Cache policy of the row DMA: non-temporal (EPX_ROW_NT=0 brings the default policy back for A/B).  A site's rows are re-read
// every pass but never hit a cache in between (2 MB per CU and pass through a 4 MB L2 that 32 CUs share, 0.66 GB per pass
// through the 256 MB Infinity Cache): streamed with the default policy they only evict what the caches could keep -- the
// tree stacks, the cold store, the cavity precision -- and the DMA itself lands later (MI355X_MICROARCH.md, nt-weights).
// Measured on the C5 shard, one box, order default / nt / nt / default (bench.py --config c5shard --steps 1 --warmup 1):
// 12.84 / 15.28 / 15.44 / 12.85 site-updates/s, 152 -> 127 us per lock-step pass and CU, 54.4 -> 65.5 % of the HBM peak.
#ifndef EPX_ROW_NT
#define EPX_ROW_NT 1
#endif
#if EPX_ROW_NT
#define EPX_NT " nt"
#else
#define EPX_NT ""
#endif
// LDS-DMA pieces as inline asm: hipcc does not count them, so it inserts no vmcnt(0) before the
// LDS reads of the ring (it does for the builtin); completion is counted by hand (wait_vm).
// M0 = wave-uniform LDS destination, written in the statement that uses it.
// (the destination is made scalar behind an opaque vector copy: inside a called function -- the looping form of the pieced
// launch -- the compiler folded __builtin_amdgcn_readfirstlane of a value it knows to be uniform and then handed the
// inline assembly a VECTOR register for its "s" operand.  A v_readfirstlane_b32 written as inline assembly compiled and
// faulted on the device)
__device__ inline unsigned scalar_of(unsigned x) {
    asm volatile("" : "+v"(x));                   // (opaque: a vector value the compiler knows nothing about)
    return __builtin_amdgcn_readfirstlane(x);
}
__device__ inline void glds16(const void *src, unsigned lds_off) {
    unsigned keep;
    const unsigned dst = scalar_of(lds_off);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" EPX_NT "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__device__ inline void glds4(const void *src, unsigned lds_off) {
    unsigned keep;
    const unsigned dst = scalar_of(lds_off);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that leaves VMEM (the LDS-DMA ring) in flight
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------- LDS map
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
__device__ inline lds_f64 *lds_d(unsigned byte_off) { return reinterpret_cast<lds_f64 *>(byte_off); }
__device__ inline lds_i32 *lds_i(unsigned byte_off) { return reinterpret_cast<lds_i32 *>(byte_off); }

// Byte offsets from the start of the engine's LDS block (the start of the dynamic segment).  The
// block depends on the largest number of groups per site (K < J: several (eta, etb) blocks per
// site, each with its own coefficients / gradient) and of tiles per site of the launch.
struct StreamMap {
    unsigned ring;      // NSL slots of TR row images
    unsigned yring;     // NSL x TR responses: int32 (logistic family) or float64 (Gaussian family)
    unsigned is2;       // 4 chains: 1 / sigma^2 of the Gaussian likelihood (published with alpha, beta)
    unsigned part;      // 2 buffers x 4 waves x [chain][row] forward partials
    unsigned gs;        // 2 buffers x [chain][row] residuals
    unsigned red;       // 4 chains x {-, ll}
    unsigned tdesc;     // per tile: first row (int32), rows valid | group << 8 (int32)
    unsigned beta;      // groups x DPB x 4 coefficients, [group][column][chain]
    unsigned gsum;      // groups x DPB x 4 gradient wrt the coefficients
    unsigned alpha;     // groups x 4 intercepts
    unsigned da;        // groups x 4 sums of residuals
    unsigned end;
};
template <int DPB> __host__ __device__ inline StreamMap stream_map(int ngmax, int ntmax, int gauss) {
    using Gm = StreamGeom<DPB>;
    StreamMap m;
    unsigned o = 0;
    m.ring = o; o += NSL * Gm::SLOTB;
    m.yring = o; o += NSL * TR * (gauss ? 8 : 4);
    m.is2 = o; o += gauss ? NCH * 8 : 0;
    m.part = o; o += 2 * 4 * 64 * 8;
    m.gs = o; o += 2 * 64 * 8;
    m.red = o; o += NCH * 2 * 8;
    m.tdesc = o; o += ((unsigned)ntmax * 8 + 15) / 16 * 16;
    m.beta = o; o += (unsigned)ngmax * DPB * NCH * 8;
    m.gsum = o; o += (unsigned)ngmax * DPB * NCH * 8;
    m.alpha = o; o += (unsigned)ngmax * NCH * 8;
    m.da = o; o += (unsigned)ngmax * NCH * 8;
    m.end = o;
    return m;
}

// what the sampler kernel touches directly (generic pointers into the same block)
struct StreamLds {
    double *beta_s, *Gs, *alpha_s, *da_s, *is2_s;
    int *tdesc;
    template <int DPB> __device__ void carve(unsigned char *base, int ngmax, int ntmax, int gauss) {
        const StreamMap m = stream_map<DPB>(ngmax, ntmax, gauss);
        is2_s = reinterpret_cast<double *>(base + m.is2);
        beta_s = reinterpret_cast<double *>(base + m.beta);
        Gs = reinterpret_cast<double *>(base + m.gsum);
        alpha_s = reinterpret_cast<double *>(base + m.alpha);
        da_s = reinterpret_cast<double *>(base + m.da);
        tdesc = reinterpret_cast<int *>(base + m.tdesc);
    }
};

// tiles of a site: every group's rows are tiled separately (a tile never straddles two groups);
// returns the number of tiles, fills desc (if not NULL) with {first row, rows | group << 8}
__host__ __device__ inline int build_tiles(int ng, const long long *glim_rel, int *desc) {
    int t = 0;
    for (int g = 0; g < ng; ++g)
        for (long long r = glim_rel[g]; r < glim_rel[g + 1]; r += TR) {
            if (desc) {
                const long long left = glim_rel[g + 1] - r;
                desc[2 * t] = (int)r;
                desc[2 * t + 1] = (int)(left < TR ? left : TR) | (g << 8);
            }
            ++t;
        }
    return t;
}

// ---------------------------------------------------------------------------------- loader
// state of one pass (the sampler kernel keeps one per thread and hands its scalars to
// stream_pass in registers -- a struct argument would travel through scratch memory, and the
// callee's lazy loads of it put vmcnt(0) waits inside the loader's loop)
template <int DPB> struct PassArgs {
    const double *Xg;           // first row of the site
    const int *yg;              // its responses: int32 0/1, or (gauss) the float64 responses viewed as int32 pairs
    int gauss;                  // Gaussian likelihood: y real, residual (y - f) / sigma^2 with 1/sigma^2 per chain in LDS
    int n, D, ntile;
    int ngmax, ntmax;           // LDS map parameters of the launch
    unsigned lds0;              // LDS byte address of the engine's block
    int slot_f;                 // ring: slot of the next tile to consume
    int slot_i;                 // ring: slot the next DMA goes to       (loader)
    int t_i;                    // ring: site tile the next DMA fetches  (loader)
    int wave, lane;
    unsigned off[StreamGeom<DPB>::NI];   // loader: per-lane byte offsets of the pieces of a full tile
};
struct PassOut { double ll; int slot_f, slot_i, t_i; };

template <int DPB>
__device__ inline void loader_init(PassArgs<DPB> &pa, int lane) {
    using Gm = StreamGeom<DPB>;
    const int rl = lane / Gm::CPR, sl = lane % Gm::CPR;
#pragma unroll
    for (int q = 0; q < Gm::NI; ++q) {
        const int r = q * Gm::RPI + rl;
        pa.off[q] = (unsigned)(r * pa.D * 8 + (sl ^ (r & (Gm::CPR - 1))) * 16);
    }
}

// DMA of tile `tt` of the site into ring slot `slot` (executed by the loader wave only)
template <int DPB>
__device__ inline void ring_issue(const PassArgs<DPB> &s, const StreamMap &M, int tt, int slot, int lane) {
    using Gm = StreamGeom<DPB>;
    const unsigned dst = scalar_of(s.lds0 + M.ring + slot * Gm::SLOTB);
    const int row0 = __builtin_amdgcn_readfirstlane(*lds_i(s.lds0 + M.tdesc + tt * 8));
    const int nval = __builtin_amdgcn_readfirstlane(*lds_i(s.lds0 + M.tdesc + tt * 8 + 4)) & 255;
    if (nval == TR) {
        // full tile: one scalar base + the precomputed lane offsets
        // (the base comes out of v_readfirstlane: a VMEM read of an SGPR that a VALU instruction wrote needs 5 wait states,
        // and the compiler's hazard recogniser does not look into the statement -- s_mov, s_mov, s_nop 2 are those five)
        const unsigned long long base = (unsigned long long)(s.Xg + (size_t)row0 * s.D);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const unsigned long long sbase = ((unsigned long long)bhi << 32) | blo;
        unsigned keep;
        if constexpr (Gm::NI == 16) {
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %3, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %11, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %12, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %13, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %14, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %15, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %16, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %17, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %18, %2" EPX_NT "\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst), "s"(sbase), "v"(s.off[0]), "v"(s.off[1]), "v"(s.off[2]), "v"(s.off[3]), "v"(s.off[4]),
                  "v"(s.off[5]), "v"(s.off[6]), "v"(s.off[7]), "v"(s.off[8]), "v"(s.off[9]), "v"(s.off[10]),
                  "v"(s.off[11]), "v"(s.off[12]), "v"(s.off[13]), "v"(s.off[14]), "v"(s.off[15])
                : "memory", "scc");
        } else {
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %3, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %2" EPX_NT "\n\t"
                "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, %2" EPX_NT "\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst), "s"(sbase), "v"(s.off[0]), "v"(s.off[1]), "v"(s.off[2]), "v"(s.off[3]), "v"(s.off[4]),
                  "v"(s.off[5]), "v"(s.off[6]), "v"(s.off[7])
                : "memory", "scc");
        }
    } else {
        // ragged last tile of a group: rows beyond it are clamped to its last row
        const int rl = lane / Gm::CPR, sl = lane % Gm::CPR;
#pragma unroll
        for (int q = 0; q < Gm::NI; ++q) {
            const int r = q * Gm::RPI + rl;
            const int c = sl ^ (r & (Gm::CPR - 1));
            const int row = row0 + (r < nval ? r : nval - 1);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(s.Xg) + ((size_t)row * s.D + 2 * c) * 8;
            glds16(src, dst + q * 1024);
        }
    }
    if (s.gauss) {
        // 16 doubles = 32 dwords: lane -> (row lane / 2, half lane % 2); the DMA writes consecutive dwords
        const int r = (lane >> 1) & (TR - 1);
        const int row = row0 + (r < nval ? r : nval - 1);
        if (lane < 2 * TR) glds4(s.yg + 2 * (size_t)row + (lane & 1), s.lds0 + M.yring + slot * TR * 8);
    } else {
        const int r = lane & (TR - 1);
        const int row = row0 + (r < nval ? r : nval - 1);
        if (lane < TR) glds4(s.yg + row, s.lds0 + M.yring + slot * TR * 4);
    }
}

// Prime the ring: the first three tiles (loader wave; the site must have at least one row).
template <int DPB>
__device__ inline void ring_prime(PassArgs<DPB> &s, int lane) {
    const StreamMap M = stream_map<DPB>(s.ngmax, s.ntmax, s.gauss);
    for (int i = 0; i < EPX_RING_AHEAD; ++i) {
        ring_issue<DPB>(s, M, s.t_i, s.slot_i, lane);
        s.t_i = s.t_i + 1 == s.ntile ? 0 : s.t_i + 1;
        s.slot_i = s.slot_i + 1 == NSL ? 0 : s.slot_i + 1;
    }
}

// ---------------------------------------------------------------------------------- one pass
// All six waves call stream_pass once per leapfrog (same number of barriers on every path).
// In: beta, alpha per group (LDS, published by a barrier before the call), the tile table.
// Out (LDS; valid on return, the pass ends with a barrier): gsum[group][column][chain],
// da[group][chain]; returned: ll of this wave's chain (chain waves) and the new ring position.
template <int DPB>
__device__ __attribute__((noinline)) PassOut stream_pass_impl(const double *Xg, const int *yg, int n, int D, int ntile,
                                                             int ngmax, int ntmax, unsigned lds0, int slot_f0,
                                                             int slot_i0, int t_i0, int wave_, int lane, int gauss_) {
    using Gm = StreamGeom<DPB>;
    PassArgs<DPB> s;
    s.Xg = Xg; s.yg = yg; s.n = n; s.D = D; s.ntile = ntile; s.lds0 = lds0;
    s.gauss = __builtin_amdgcn_readfirstlane(gauss_);
    s.ngmax = __builtin_amdgcn_readfirstlane(ngmax); s.ntmax = __builtin_amdgcn_readfirstlane(ntmax);
    s.slot_f = slot_f0; s.slot_i = slot_i0; s.t_i = t_i0; s.wave = wave_; s.lane = lane;
    const int wave = __builtin_amdgcn_readfirstlane(s.wave);
    if (wave == NCH) loader_init<DPB>(s, lane);
    const StreamMap M = stream_map<DPB>(s.ngmax, s.ntmax, s.gauss);
    const int nt = __builtin_amdgcn_readfirstlane(s.ntile);
    const unsigned B0 = __builtin_amdgcn_readfirstlane(s.lds0);
    const int l15 = lane & 15, lg = lane >> 4, l3 = lane & 3;
    int slot_f = __builtin_amdgcn_readfirstlane(s.slot_f);      // slot of tile p
    int slot_l = slot_f, slot_b = slot_f;                       // slots of tiles p-1 and p-2 (set as the pipe fills)
    PassOut out;
    out.ll = 0.0; out.slot_i = s.slot_i; out.t_i = s.t_i;
    auto tile_group = [&](int t) { return __builtin_amdgcn_readfirstlane(*lds_i(B0 + M.tdesc + t * 8 + 4)) >> 8; };
    if (wave < NCH) {
        // ------------------------------------------------ chain waves: the two products
        double bq[Gm::KS];                      // forward B operand: beta[group][k0 + lg][chain l3]
        double acc[Gm::MB];
#pragma unroll
        for (int mb = 0; mb < Gm::MB; ++mb) acc[mb] = 0.0;
        int g_bq = -1, g_acc = -1;              // group whose coefficients are in bq / whose gradient is in acc
        int g_l = 0, g_b = 0;                   // groups of tiles p-1, p-2
        int g_n = tile_group(0);                // group of the tile of the coming phase (fetched one phase ahead)
        const int frow = 4 * (l15 >> 2) + lg;   // D lane of the forward product -> (row frow, chain l3)
        auto flush = [&]() {
#pragma unroll
            for (int mb = 0; mb < Gm::MB; ++mb) {
                *lds_d(B0 + M.gsum + ((g_acc * DPB + wave * Gm::DW + 16 * mb + 4 * (l15 >> 2) + lg) * NCH + l3) * 8) = acc[mb];
                acc[mb] = 0.0;
            }
        };
        for (int p = 0; p < nt + 2; ++p) {
            lds_barrier();
            const bool do_f = p < nt, do_b = p >= 2;
            int g_f = 0;
            double a[Gm::KS], bb[4], aa[4 * Gm::MB];
            const int g_next = *lds_i(B0 + M.tdesc + (p + 1 < nt ? p + 1 : nt - 1) * 8 + 4);
            if (do_f) {
                g_f = g_n;
                if (g_f != g_bq) {
#pragma unroll
                    for (int ks = 0; ks < Gm::KS; ++ks)
                        bq[ks] = *lds_d(B0 + M.beta + ((g_f * DPB + wave * Gm::DW + 4 * ks + lg) * NCH + l3) * 8);
                    g_bq = g_f;
                }
                const unsigned tile = B0 + M.ring + slot_f * Gm::SLOTB + l15 * Gm::ROWB;
#pragma unroll
                for (int ks = 0; ks < Gm::KS; ++ks) {
                    const int col = wave * Gm::DW + 4 * ks + lg;
                    a[ks] = *lds_d(tile + (((col >> 1) ^ l15) & (Gm::CPR - 1)) * 16 + (col & 1) * 8);
                }
            }
            if (do_b) {
                if (g_b != g_acc) {
                    if (g_acc >= 0) flush();
                    g_acc = g_b;
                }
                const unsigned tile = B0 + M.ring + slot_b * Gm::SLOTB;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int rk = 2 * ks + (lg >> 1) + 8 * (lg & 1);
                    bb[ks] = *lds_d(B0 + M.gs + ((p & 1) * 64 + l3 * 16 + rk) * 8);
#pragma unroll
                    for (int mb = 0; mb < Gm::MB; ++mb) {
                        const int col = wave * Gm::DW + 16 * mb + l15;
                        aa[ks * Gm::MB + mb] =
                            *lds_d(tile + rk * Gm::ROWB + (((col >> 1) ^ rk) & (Gm::CPR - 1)) * 16 + (col & 1) * 8);
                    }
                }
            }
            if (do_f) {
                double f0 = 0.0, f1 = 0.0;
#pragma unroll
                for (int ks = 0; ks < Gm::KS; ks += 2) {
                    f0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ks], bq[ks], f0, 0, 0, 0);
                    f1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ks + 1], bq[ks + 1], f1, 0, 0, 0);
                }
                *lds_d(B0 + M.part + (((p & 1) * 4 + wave) * 64 + l3 * 16 + frow) * 8) = f0 + f1;
            }
            if (do_b) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int mb = 0; mb < Gm::MB; ++mb)
                        acc[mb] = __builtin_amdgcn_mfma_f64_4x4x4f64(aa[ks * Gm::MB + mb], bb[ks], acc[mb], 0, 0, 0);
            }
            slot_b = slot_l; slot_l = slot_f;
            g_b = g_l; g_l = g_f;
            g_n = __builtin_amdgcn_readfirstlane(g_next) >> 8;
            if (p < nt) slot_f = slot_f + 1 == NSL ? 0 : slot_f + 1;
        }
        if (g_acc >= 0) flush();                // publish G[group][column][chain] of the last group
    } else if (wave == NCH) {
        // ------------------------------------------------ loader
        int slot_i = __builtin_amdgcn_readfirstlane(s.slot_i), t_i = __builtin_amdgcn_readfirstlane(s.t_i);
        for (int p = 0; p < nt + 2; ++p) {
            if (p < nt) wait_vm<(EPX_RING_AHEAD - 1) * Gm::G>();   // tile p has landed; tiles p+1, p+2 stay in flight
            lds_barrier();
            if (p < nt) {
                ring_issue<DPB>(s, M, t_i, slot_i, lane);       // tile p+3 -> the slot of tile p-3
                t_i = t_i + 1 == nt ? 0 : t_i + 1;
                slot_i = slot_i + 1 == NSL ? 0 : slot_i + 1;
                slot_f = slot_f + 1 == NSL ? 0 : slot_f + 1;
            }
        }
        out.slot_i = slot_i; out.t_i = t_i;
    } else {
        // ------------------------------------------------ logistic terms: lane = (row l15, chain lg)
        // The tile descriptor is fetched one phase ahead and alpha / the residual sum stay in
        // registers while the group does not change: a phase is as long as its slowest wave, and
        // two dependent LDS round trips plus a read-modify-write per tile made it this one.
        double ll = 0.0, da_acc = 0.0, alpha_l = 0.0;
        double wprod = 1.0, wlog = 0.0;         // log-likelihood = sum(lin) - log(prod(w)), see logistic_split
        const bool gauss = s.gauss != 0;
        const double is2 = gauss ? *lds_d(B0 + M.is2 + lg * 8) : 0.0;     // 1 / sigma^2 of this lane's chain
        for (int i = lane; i < s.ngmax * NCH; i += 64) *lds_d(B0 + M.da + i * 8) = 0.0;
        int pk_n = __builtin_amdgcn_readfirstlane(*lds_i(B0 + M.tdesc + 4));       // tile 0
        int g_cur = -1;
        auto flush_da = [&]() {                 // residual sum of the group's tiles -> da[group][chain]
            double dsum = da_acc;
            dsum += dpp_d<DPP_QUAD_XOR1>(dsum); dsum += dpp_d<DPP_QUAD_XOR2>(dsum);
            dsum += dpp_d<DPP_ROW_HALF_MIRROR>(dsum); dsum += dpp_d<DPP_ROW_MIRROR>(dsum);
            if (l15 == 0) *lds_d(B0 + M.da + (g_cur * NCH + lg) * 8) = dsum;
            da_acc = 0.0;
        };
        for (int p = 0; p < nt + 2; ++p) {
            lds_barrier();
            if (p >= 1 && p - 1 < nt) {
                const int pb = (p - 1) & 1;
                const int pk = pk_n;
                const int pk_next = *lds_i(B0 + M.tdesc + (p < nt ? p : nt - 1) * 8 + 4);      // used next phase
                const int nval = pk & 255, grp = pk >> 8;
                if (grp != g_cur) {
                    if (g_cur >= 0) flush_da();
                    g_cur = grp;
                    alpha_l = *lds_d(B0 + M.alpha + (grp * NCH + lg) * 8);
                }
                double f = alpha_l;
#pragma unroll
                for (int w = 0; w < 4; ++w) f += *lds_d(B0 + M.part + ((pb * 4 + w) * 64 + lane) * 8);
                double l = 0.0, w = 1.0, g = 0.0;
                if (gauss) {            // -(y - f)^2 / (2 sigma^2); the -n log sigma term is the caller's (m*a_sg.stan)
                    const double res = *lds_d(B0 + M.yring + (slot_l * TR + l15) * 8) - f;
                    g = res * is2; l = -0.5 * res * g;
                } else logistic_split(f, (double)*lds_i(B0 + M.yring + (slot_l * TR + l15) * 4), l, w, g);
                const bool ok = l15 < nval;
                g = ok ? g : 0.0;
                *lds_d(B0 + M.gs + (pb * 64 + lane) * 8) = g;
                ll += ok ? l : 0.0; wprod *= ok ? w : 1.0; da_acc += g;
                if ((p & 255) == 0) { wlog += log(wprod); wprod = 1.0; }       // w <= 2: no overflow in 256 factors
                pk_n = __builtin_amdgcn_readfirstlane(pk_next);
            }
            slot_b = slot_l; slot_l = slot_f;
            if (p < nt) slot_f = slot_f + 1 == NSL ? 0 : slot_f + 1;
        }
        if (g_cur >= 0) flush_da();
        ll -= wlog + log(wprod);
        // ll: sum over the 16 rows of each lane group -> red[chain]
        ll += dpp_d<DPP_QUAD_XOR1>(ll); ll += dpp_d<DPP_QUAD_XOR2>(ll);
        ll += dpp_d<DPP_ROW_HALF_MIRROR>(ll); ll += dpp_d<DPP_ROW_MIRROR>(ll);
        if (l15 == 0) *lds_d(B0 + M.red + lg * 16 + 8) = ll;
    }
    out.slot_f = slot_f;
    lds_barrier();
    if (wave < NCH) out.ll = *lds_d(B0 + M.red + wave * 16 + 8);
    return out;
}
template <int DPB>
__device__ inline PassOut stream_pass(const PassArgs<DPB> &s) {
    return stream_pass_impl<DPB>(s.Xg, s.yg, s.n, s.D, s.ntile, s.ngmax, s.ntmax, s.lds0, s.slot_f, s.slot_i, s.t_i,
                                 s.wave, s.lane, s.gauss);
}

// ====================================================================== resident variant
// The same lock-step gradient for sites whose rows FIT the LDS (D <= 32; BASELINE configs
// C3 / C4: n_j = 500, D = 32 -> 128 KB): the row images are loaded once per site update and
// stay; there is no DMA ring, no loader / logistic wave and no barrier inside the pass.  Every
// chain wave owns every 4th tile and runs it alone: forward product -> the tile's logistic
// terms directly on the MFMA result (D layout: one (row, chain) per lane) -> residuals through
// a wave-private LDS line -> backward product.  Per-wave partial gradients / residual sums are
// combined after one barrier.
template <int DPB> struct ResGeom {
    static constexpr int CPR = DPB / 2;            // 16-byte chunks per row image
    static constexpr int ROWB = DPB * 8;           // bytes per row image
    static constexpr int KS = DPB / 4;             // forward k-steps per tile (all columns)
    static constexpr int MB = DPB / 16;            // backward 16-column blocks
};
struct ResMap {
    unsigned ximg;      // npad row images (site rows, zero padded to a multiple of 16)
    unsigned yimg;      // npad responses (int32)
    unsigned tdesc;     // per tile: first row, rows valid | group << 8
    unsigned beta;      // groups x DPB x 4, [group][column][chain]
    unsigned gsum;      // groups x DPB x 4 gradient wrt the coefficients (combined)
    unsigned gpart;     // 4 waves x groups x DPB x 4 partial gradients
    unsigned alpha;     // groups x 4 intercepts
    unsigned da;        // groups x 4 residual sums (combined)
    unsigned dapart;    // 4 waves x groups x 4
    unsigned gsw;       // 4 waves x [chain][row] residuals of the tile in hand
    unsigned llpart;    // 4 waves x 4 chains
    unsigned end;
};
template <int DPB> __host__ __device__ inline ResMap res_map(int nmax, int ngmax, int ntmax) {
    using Gm = ResGeom<DPB>;
    // with several groups a tile may start on any row: its 16-row window can run 15 rows past the site
    const unsigned npad = (((unsigned)nmax + 15u) & ~15u) + (ngmax > 1 ? 16u : 0u);
    ResMap m;
    unsigned o = 0;
    m.ximg = o; o += npad * Gm::ROWB;
    m.yimg = o; o += npad * 4;
    m.tdesc = o; o += ((unsigned)ntmax * 8 + 15) / 16 * 16;
    m.beta = o; o += (unsigned)ngmax * DPB * NCH * 8;
    m.gsum = o; o += (unsigned)ngmax * DPB * NCH * 8;
    m.gpart = o; o += 4u * ngmax * DPB * NCH * 8;
    m.alpha = o; o += (unsigned)ngmax * NCH * 8;
    m.da = o; o += (unsigned)ngmax * NCH * 8;
    m.dapart = o; o += 4u * ngmax * NCH * 8;
    m.gsw = o; o += 4 * 64 * 8;
    m.llpart = o; o += 4 * NCH * 8;
    m.end = o;
    return m;
}

// byte offset of element (row, col) in the swizzled row image
template <int DPB> __device__ inline unsigned ximg_off(int row, int col) {
    using Gm = ResGeom<DPB>;
    return (unsigned)row * Gm::ROWB + ((((unsigned)col >> 1) ^ (unsigned)row) & (Gm::CPR - 1)) * 16 + (col & 1) * 8;
}

// Copy the site's rows into the LDS image (all threads of the workgroup; zero padding).
template <int DPB>
__device__ inline void res_load_site(const double *Xg, const int *yg, int n, int D, int ngmax, unsigned B0,
                                     const ResMap &M, int tid, int nthreads) {
    using Gm = ResGeom<DPB>;
    const int npad = ((n + 15) & ~15) + (ngmax > 1 ? 16 : 0);
    for (int idx = tid; idx < npad * Gm::CPR; idx += nthreads) {
        const int r = idx / Gm::CPR, c = idx % Gm::CPR;
        double x0 = 0.0, x1 = 0.0;
        if (r < n) {
            if (2 * c < D) x0 = Xg[(size_t)r * D + 2 * c];
            if (2 * c + 1 < D) x1 = Xg[(size_t)r * D + 2 * c + 1];
        }
        const unsigned o = B0 + M.ximg + (unsigned)r * Gm::ROWB + (((unsigned)c ^ (unsigned)r) & (Gm::CPR - 1)) * 16;
        *lds_d(o) = x0; *lds_d(o + 8) = x1;
    }
    for (int r = tid; r < npad; r += nthreads) *lds_i(B0 + M.yimg + r * 4) = r < n ? yg[r] : 0;
}

// One pass of the resident engine, executed by the 4 chain waves.  In: beta, alpha per group
// (LDS, published by a barrier before the call), the tile table.  Out (valid on return: the
// pass ends with barriers): gsum, da per group; returns ll of this wave's chain.
// A wave works on UT of its tiles at a time: one tile is a chain of dependent steps (LDS reads
// -> MFMA chain -> ~90 dependent logistic instructions -> lane exchange -> MFMA chain), and a
// lone wave per SIMD has nothing else to issue meanwhile; UT independent tiles fill those slots.
template <int DPB, int UT>
__device__ inline void resident_tiles(unsigned B0, const ResMap &M, const int (&tt)[UT], int grp, int lane,
                                      const double (&bq)[ResGeom<DPB>::KS], double (&acc)[ResGeom<DPB>::MB],
                                      double &ll, double &da) {
    using Gm = ResGeom<DPB>;
    const int l15 = lane & 15, lg = lane >> 4, l3 = lane & 3;
    const int frow = 4 * (l15 >> 2) + lg;       // D lane of the forward product -> (row frow, chain l3)
    int row0[UT], nval[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        row0[u] = __builtin_amdgcn_readfirstlane(*lds_i(B0 + M.tdesc + tt[u] * 8));
        nval[u] = __builtin_amdgcn_readfirstlane(*lds_i(B0 + M.tdesc + tt[u] * 8 + 4)) & 255;
    }
    // ---- forward: rows row0 + (0..15), all columns
    double a[UT][Gm::KS];
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int ks = 0; ks < Gm::KS; ++ks) a[u][ks] = *lds_d(B0 + M.ximg + ximg_off<DPB>(row0[u] + l15, 4 * ks + lg));
    double f0[UT], f1[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) { f0[u] = 0.0; f1[u] = 0.0; }
#pragma unroll
    for (int ks = 0; ks < Gm::KS; ks += 2)
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            f0[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u][ks], bq[ks], f0[u], 0, 0, 0);
            f1[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u][ks + 1], bq[ks + 1], f1[u], 0, 0, 0);
        }
    // ---- logistic terms on the product's own lanes: (row frow, chain l3)
    const double alpha = *lds_d(B0 + M.alpha + ((unsigned)grp * NCH + l3) * 8);
    double g[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        const double f = (f0[u] + f1[u]) + alpha;
        const double yy = (double)*lds_i(B0 + M.yimg + (row0[u] + frow) * 4);
        double l = 0.0, gg = 0.0;
        logistic_terms(f, yy, l, gg);
        const bool ok = frow < nval[u];
        l = ok ? l : 0.0; gg = ok ? gg : 0.0;
        ll += l; da += gg;
        g[u] = gg;
    }
    // ---- backward: (all columns) x (rows of the tile) x chains.  The B operand of k-step ks in
    // lane (k = lg, j = l3) is the residual of (row rk, chain l3), which sits in the D lane
    // 16 (rk & 3) + 4 (rk >> 2) + l3: a lane gather, no LDS storage
    double bb[UT][4], aa[UT][4 * Gm::MB];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int rk = 2 * ks + (lg >> 1) + 8 * (lg & 1);
        const int src = 16 * (rk & 3) + 4 * (rk >> 2) + l3;
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            bb[u][ks] = __shfl(g[u], src, 64);
#pragma unroll
            for (int mb = 0; mb < Gm::MB; ++mb)
                aa[u][ks * Gm::MB + mb] = *lds_d(B0 + M.ximg + ximg_off<DPB>(row0[u] + rk, 16 * mb + l15));
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int mb = 0; mb < Gm::MB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f64_4x4x4f64(aa[u][ks * Gm::MB + mb], bb[u][ks], acc[mb], 0, 0, 0);
}

template <int DPB>
__device__ inline double resident_pass(unsigned B0, const ResMap &M, int nt, int ngmax, int wave, int lane,
                                       int nthreads, int tid) {
    using Gm = ResGeom<DPB>;
    constexpr int UT = 4;
    const int l15 = lane & 15, lg = lane >> 4, l3 = lane & 3;
    // wave-private partials start at zero (a wave may own no tile of some group)
    for (int i = lane; i < ngmax * DPB * NCH; i += 64) *lds_d(B0 + M.gpart + ((unsigned)wave * ngmax * DPB * NCH + i) * 8) = 0.0;
    for (int i = lane; i < ngmax * NCH; i += 64) *lds_d(B0 + M.dapart + ((unsigned)wave * ngmax * NCH + i) * 8) = 0.0;
    double bq[Gm::KS];
    double acc[Gm::MB];
#pragma unroll
    for (int mb = 0; mb < Gm::MB; ++mb) acc[mb] = 0.0;
    int g_cur = -1;
    double ll = 0.0, da = 0.0;
    auto flush = [&]() {
#pragma unroll
        for (int mb = 0; mb < Gm::MB; ++mb) {
            // D lane (i = lg, b = l15 >> 2, j = l3) -> G[column 16 mb + 4 b + i][chain j]
            *lds_d(B0 + M.gpart + (((unsigned)(wave * ngmax + g_cur) * DPB + 16 * mb + 4 * (l15 >> 2) + lg) * NCH + l3) * 8) = acc[mb];
            acc[mb] = 0.0;
        }
        // residual sum of the group: over the lanes of the same chain (all b, all i)
        double t = da;
        t += dpp_d<0x124>(t); t += dpp_d<0x128>(t);            // row_ror 4, 8: the four b of a row of 16
        t += partner_d<4>(t, lane); t += partner_d<5>(t, lane);  // the four i
        if (lane < NCH) *lds_d(B0 + M.dapart + ((unsigned)(wave * ngmax + g_cur) * NCH + lane) * 8) = t;
        da = 0.0;
    };
    auto tile_group = [&](int t) { return __builtin_amdgcn_readfirstlane(*lds_i(B0 + M.tdesc + t * 8 + 4)) >> 8; };
    auto enter_group = [&](int grp) {
        if (grp == g_cur) return;
        if (g_cur >= 0) flush();
        g_cur = grp;
#pragma unroll
        for (int ks = 0; ks < Gm::KS; ++ks) bq[ks] = *lds_d(B0 + M.beta + (((unsigned)grp * DPB + 4 * ks + lg) * NCH + l3) * 8);
    };
    int t = wave;
    while (t < nt) {
        // UT of this wave's tiles together when they exist and belong to one group, else one
        const int tlast = t + NCH * (UT - 1);
        bool batch = tlast < nt;
        const int grp = tile_group(t);
        if (batch) batch = tile_group(tlast) == grp;     // groups are contiguous: first == last is enough
        enter_group(grp);
        if (batch) {
            int tt[UT];
#pragma unroll
            for (int u = 0; u < UT; ++u) tt[u] = t + NCH * u;
            resident_tiles<DPB, UT>(B0, M, tt, grp, lane, bq, acc, ll, da);
            t += NCH * UT;
        } else {
            const int tt[1] = {t};
            resident_tiles<DPB, 1>(B0, M, tt, grp, lane, bq, acc, ll, da);
            t += NCH;
        }
    }
    if (g_cur >= 0) flush();
    {
        double tl = ll;
        tl += dpp_d<0x124>(tl); tl += dpp_d<0x128>(tl);
        tl += partner_d<4>(tl, lane); tl += partner_d<5>(tl, lane);
        if (lane < NCH) *lds_d(B0 + M.llpart + ((unsigned)wave * NCH + lane) * 8) = tl;
    }
    lds_barrier();
    // combine the four waves' partials
    for (int i = tid; i < ngmax * DPB * NCH; i += nthreads) {
        double tsum = 0.0;
#pragma unroll
        for (int w = 0; w < NCH; ++w) tsum += *lds_d(B0 + M.gpart + ((unsigned)w * ngmax * DPB * NCH + i) * 8);
        *lds_d(B0 + M.gsum + i * 8) = tsum;
    }
    for (int i = tid; i < ngmax * NCH; i += nthreads) {
        double tsum = 0.0;
#pragma unroll
        for (int w = 0; w < NCH; ++w) tsum += *lds_d(B0 + M.dapart + ((unsigned)w * ngmax * NCH + i) * 8);
        *lds_d(B0 + M.da + i * 8) = tsum;
    }
    double llc = 0.0;
#pragma unroll
    for (int w = 0; w < NCH; ++w) llc += *lds_d(B0 + M.llpart + ((unsigned)w * NCH + wave) * 8);
    lds_barrier();
    return llc;
}

}  // namespace epx
