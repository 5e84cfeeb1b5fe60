// Kernel argument blocks and launch entry points shared by the .hip files and
// the C-ABI implementation (epx_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace epx {

// Where a dense kernel keeps its two d x ld work matrices + 4 vectors.
struct DenseWs {
    int use_lds;        // 1: dynamic LDS, 0: global workspace
    double *global;     // per-block slots of 2*d*ld + 4*ld doubles
};

struct CavityArgs {
    int k0, d, ld;
    DenseWs ws;
    const double *Q, *r;               // global approximation (device)
    const double *Qsite, *rsite;       // site array base
    const double *dQsite, *drsite;     // optional update (proposal = site + df * update)
    size_t site_stride, rsite_stride;
    double df;
    double *cav_Om, *cav_mu;           // K x d x d, K x d
    uint8_t *flags;                    // K
};

struct MomentArgs {
    int k0, d, ld, S, prec_estim;
    DenseWs ws;
    const double *draws;               // element (site, s, i) at site*stride_site + s*stride_s + i*stride_i
    long draws_site0;
    long stride_site, stride_s, stride_i;
    const double *Q, *r;               // global approximation subtracted at method.py:457-458
    double *dQi, *dri;                 // K x d x d, K x d
    double *tilt_mean, *tilt_scatter;  // K x d, K x d x d
    uint8_t *flags;
};

struct SumArgs {
    int K, d, len, nslice;
    const double *Qi, *ri, *dQi, *dri;
    double *partial;                   // nslice x len
    double *out;                       // len
};

struct GlobalArgs {
    int d, ld, want_moments;
    DenseWs ws;
    const double *packed;              // [sum Qi, sum ri, sum dQi, sum dri] or NULL (use Q,r as they are)
    const double *Q0, *r0;
    double df;
    double *Q, *r, *S, *m;
    int *flag;
    // damping sweep (find_damp.py:146-173): selection criteria of the trial against a target
    // tgt = [m_t (d), S_t (d*d), xbar (d), Sc (d*d), half_logdet_St, n_samp] ; crit = [global_pd, cav_pd, mse, kl, ll]
    const double *tgt;
    double *crit;
};

struct InvertArgs {
    int d, ld, cho_form;
    DenseWs ws;
    double *A, *b;
    int32_t *info;
};

struct OlseArgs {
    int d, ld, n;
    DenseWs ws;
    double *S;
    const double *P;
    int32_t *info;
};

struct ForceArgs {
    int d, ld;
    DenseWs ws;
    double *Qi;
    const double *dQi;
    double df, thresh, target;
    uint8_t *forced;
    double *min_eig;
};

__global__ void k_cavity(CavityArgs a);
__global__ void k_moments(MomentArgs a);
__global__ void k_site_sums_partial(SumArgs a);
__global__ void k_site_sums_final(SumArgs a);
__global__ void k_global(GlobalArgs a);
__global__ void k_axpy(double *out, const double *x, const double *dx, double df, size_t n);
__global__ void k_mix_partial(SumArgs a);
__global__ void k_sweep_flag(const int *all_flag, double *crit);
__global__ void k_all_flags(const uint8_t *flags, int k0, int count, int *out);
__global__ void k_trial_flags(const uint8_t *flags, int count, int site_base, const int *global_pd, double *out);
__global__ void k_invert(InvertArgs a);
__global__ void k_olse(OlseArgs a);
__global__ void k_force_pd(ForceArgs a);

// ------------------------------------------------------------------ sampler
enum { ST_STEPSIZE_MEAN = 0, ST_STEPSIZE_FINAL, ST_NLEAP, ST_NGRAD, ST_NDIV, ST_ACCEPT_MEAN,
       ST_DEPTH_MEAN, ST_FAIL, ST_COUNT };
enum { K_INIT = 0, K_MOM = 1, K_DIR = 2, K_TOP = 3, K_MERGE = 4, K_SSMOM = 5 };
enum { MAX_DEPTH_CAP = 12 };
// seconds a workgroup of a pieced launch looks for a site before it reports a lost piece (epx_pieces.h); the environment
// variable of the same name overrides it at run time (NutsArgs::dyn_wait_s)
#ifndef EPX_PIECE_WAIT_S
#define EPX_PIECE_WAIT_S 60
#endif

// columns of zeros behind the last site's cavity precision (epx_api.hip allocates them): the streaming sampler's register
// ring reads up to OM_UNROLL columns past a site's Omega (nuts_stream.hip)
enum { EPX_OM_PAD_COLS = 64 };

// Default of NutsArgs::yield_cycles: cycles of s_memtime since the state wave's last job went out PLUS the estimate of the
// bookkeeping still ahead (EPX_SM_YIELD's argument) beyond which the wave lets a pass go by.  A team's pass lasts ~6 500
// cycles; a lost pass of one chain costs the team a quarter of a pass (~2 000 cycles): HISTORY.md section 3.1g (round 4)
#ifndef EPX_YIELD_DEFAULT
#define EPX_YIELD_DEFAULT 8500
#endif
struct NutsArgs {
    int gauss;                  // Gaussian-likelihood family (m*a_sg.stan): phi[0] = log sigma, real responses in yd
    const double *yd;
    int model, D, d, P;
    int k0;                       // first site of the batch
    int chains, iter, warmup, thin, nkeep, max_depth, init_mode;
    int cpb;                      // chains per block (waves per block = cpb * WPC)
    const int64_t *k_lim;         // K+1 row limits
    const double *X;              // N x D row-major
    const uint8_t *y;             // N
    const int *y32;               // N, the same responses as int32 (DMA granule of the streaming sampler)
    const double *cav_Om;         // K x d x d
    const double *cav_mu;         // K x d
    const int64_t *seeds;         // per site of the batch (index k - k0)
    const int *order;             // optional: workgroup i works on site order[i] of the batch (longest first), or NULL
    // multi-group sites (K < J; streaming layout only): groups site_g0[k]..site_g0[k+1] with the
    // absolute row limits g_lim[g]..g_lim[g+1]; NULL = one group per site.  P is then the record
    // stride (largest coordinate count over the sites).
    const int *site_g0;
    const int64_t *g_lim;
    int ngmax, ntmax;             // streaming layout: most groups / tiles of a site (LDS map)
    double *draws;                // K x chains x nkeep x P
    double *last;                 // K x chains x P (read when init_mode == PREV, always written)
    double *chain_stats;          // K x chains x ST_COUNT
    const double *eps_in;         // test hook: per (site of batch, chain) fixed step size -> no step-size search
    const double *inv_e_in;       // test hook: per (site of batch, chain) x P diagonal inverse metric
    int t_offset;                 // test hook: transition index offset of the random stream
    // opt-in `adapt = carry` (not the reference's behaviour, see include/epx.h): start from the step size the
    // chain ended the previous call with and from the site's pooled sample variances of that call as diagonal
    // metric; warm-up then adapts the step size only.  Indexed by ABSOLUTE site; NULL = adapt from scratch;
    // a non-positive step size = no history for that chain
    const double *carry_eps;      // K x chains
    const double *carry_metric;   // K x P
    unsigned long long *stamps;   // diagnostic build (-DEPX_STAMPS): per block 8 cycle sums
    int stamps_nrec;              // ... and the host's count of records per kind (the launched grid may be smaller: looping workgroups add up their pieces)
    double *dbg;                  // test hook: if set, write lp and grad of the initial point (1+P) and stop
    double *trace;                // test hook (epx_set_trace): per (site of the batch < trace_sites, chain, transition) a record of 8 + P doubles
    int trace_sites;
    double *team_passes;          // K (absolute site): passes the row team of layout 7 made over the site's rows, the ones a chain yielded
                                  // included (added up over the pieces of a queued launch); NULL = not counted
    double *stack;                // global memory of the chains of the resident layouts, indexed by (site of the batch,
                                  // chain): stack_stride doubles each -- tree stack (max_depth * (4 NV 64 + 2)) first,
                                  // then the state wave's cold store (nuts_duo.hip); one stride for every kernel of a
                                  // call, so the two kernels of a split launch can share the buffer.  Streaming layout:
                                  // nuts_stream_chain_doubles() per chain
    size_t stack_stride;
    // dynamic LDS layout (byte offsets), computed on the host
    int n_max;                    // rows reserved for X in LDS
    int off_y, off_Om, off_mu, off_xch, off_stack, lds_bytes;
    int stack_in_lds;
    int scr_doubles;              // ... its length (the P live elements, rounded up)
    int off_scr;                  // nuts_duo.hip TEAM form: per chain one vector of LDS scratch (view -> vector order)
    int stack_ps;                 // ... and the doubles one vector of such a record takes (packed: >= P)
    int stack_lds_levels;         // nuts_duo.hip, stack not (all) in LDS: the lowest levels of every chain's tree stack that are (0: none)
    int om_in_lds;
    int off_spec;                 // > 0: LDS offset of the speculative kernel's mailbox / control records (layout 2)
    int no_spec;                  // 1: keep the bookkeeping on the gradient waves (k_nuts) even when off_spec > 0
    int yield_cycles;             // row team (layout 7): a state wave whose bookkeeping of a finished subtree / transition would
                                  // end more than this many cycles after its job went out lets the team's pass go by WITHOUT a
                                  // new job of its chain (one lost pass of one chain) instead of keeping the four chains
                                  // waiting; 0 = never
    int grp;                      // 1: sites with several groups on layout 2 (k_nuts_spec<..., GRP>, nuts_gradient_groups.inc)
    int off_gl;                   // LDS offset of the site's group row limits (grp)
    // k_nuts_duo (nuts_duo.hip): row waves + state waves with LDS hand-offs
    int duo_rw;                   // row waves per chain
    int off_tail;                 // cavity precision rows 64..d-1 (row-major, stride d rounded up to even)
    int off_slot;                 // per chain: job / result slots
    int off_flag;                 // per chain: hand-off sequence numbers
    int slot_doubles;             // doubles of one chain's slots
    int off_piece;                // 16 B: (site, first transition) of the piece a persistent workgroup is running
    int *err;                     // device word: set when a hand-off spin gives up (never in a healthy run)
    // pieced launch (layout 5; epx_set_piece_queue): seg_nwg workgroups, one per piece
    int seg_nwg;
    int persist;                          // workgroups loop over claims (seg_nwg = what the device holds at a time) instead of one per piece
    double *ckpt;                 // pieced launches: per (site, chain) a record of (4 NV + 1) x 64 doubles (sample, Welford sums, metric, scalars)
    // piece queue (epx_set_piece_queue): every workgroup claims a site by largest remaining predicted work and runs
    // dyn_len transitions of it; dyn_prog[site] = 2 x transitions done + (claimed): one word, changed atomically
    int *dyn_prog, *dyn_busy;
    const double *dyn_rate;       // predicted work per transition of every site of the batch, or NULL (all equal)
    int dyn_len, dyn_count;
    const int *dyn_lens;          // per site: transitions of one of ITS pieces (pieces of equal predicted work), or NULL: dyn_len for all
    int dyn_nb;                   // checkpoint records (piece boundaries) reserved per site
    int dyn_tail_div;             // the pieces behind 3/4 of a site's run are 1/dyn_tail_div of the nominal length (epx_pieces.h)
    int dyn_hook;                 // epx_sample_piece: a site is released as FINISHED behind its one transition (never claimable twice)
    int dyn_wait_s;               // seconds a workgroup looks for a site before it reports a lost piece (EPX_PIECE_WAIT_S; epx_pieces.h)
};

// launch wrapper implemented in nuts.hip; returns hipError_t as int
int launch_nuts(const NutsArgs &a, int count, int wpc, int dp, int nv, hipStream_t stream);
size_t nuts_lds_layout(NutsArgs &a, int wpc, int dp, int n_max);

// row-wave / state-wave variant (nuts_duo.hip): cpb chains of a site per workgroup, every chain one
// state wave + rw row waves that talk through LDS slots (no workgroup barrier in the loop)
int launch_nuts_duo(const NutsArgs &a, int count, int cpb, int rw, int dp, int nv, hipStream_t stream);
size_t nuts_duo_lds_layout(NutsArgs &a, int cpb, int rw, int dp, int n_max);
bool nuts_duo_has(int cpb, int rw, int dp, int nv);
// doubles per level of a chain's tree stack: (rho, p_sharp of the left end) of a pending left sibling
constexpr int nuts_stack_record(int nv) { return 2 * nv * 64; }
size_t nuts_resident_chain_doubles(int nv, int max_depth);

// streaming variant (nuts_stream.hip): one workgroup per site, chains in lock step, X through
// an LDS-DMA ring; dpb in {64, 128}, nv = ceil(P/64) <= 7.  a.stack holds, per (site of the
// batch, chain), nuts_stream_chain_doubles() doubles (tree stack + cold store).
int launch_nuts_stream(const NutsArgs &a, int count, int dpb, int nv, hipStream_t stream);
size_t nuts_stream_lds_bytes(int nv, int dpb, int d, int ngmax, int ntmax, int nmax_res, int gauss);
size_t nuts_stream_chain_doubles(int nv, int max_depth);

struct RhatArgs {
    int k0, chains, nkeep, P;
    const int *site_g0;           // multi-group sites: coordinates of site k = d + groups * pg (<= P); or NULL
    int d, pg;
    const double *draws;          // K x chains x nkeep x P
    const double *chain_stats;    // K x chains x ST_COUNT
    double *site_stats;           // count x EPX_ST_COUNT (batch-relative)
};
__global__ void k_site_stats(RhatArgs a);
struct CarryArgs {
    int k0, chains, nkeep, P;
    const double *draws;          // K x chains x nkeep x P
    const double *chain_stats;    // K x chains x ST_COUNT
    double *carry_eps, *carry_metric;
};
__global__ void k_carry_update(CarryArgs a);
__global__ void k_rng_probe(uint64_t seed, int chain, uint32_t t, uint32_t kind, uint32_t a,
                            uint32_t b, double *out4);

}  // namespace epx
