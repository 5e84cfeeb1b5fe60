// Private to libepx.so: the context behind the opaque epx_ctx of include/epx.h, shared by
// epx_api.hip (entry points) and epx_comm.hip (the in-library RCCL binding).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <vector>

#include "../../include/epx.h"

// error message kept per thread, returned by epx_last_error(); always returns -1
int epx_fail(const char *fmt, ...);
#define fail epx_fail

#define HIPCHK(x)                                                                         \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define CTX(c)                                      \
    if (!(c)) return fail("null context");          \
    HIPCHK(hipSetDevice((c)->device));

template <typename T>
static hipError_t dalloc(T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    return hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T));
}

static const size_t LDS_CAP = 160 * 1024;

struct epx_ctx {
    int device, model, K, D, d, P;
    int64_t N;
    hipStream_t stream;
    std::vector<int64_t> k_lim;
    // multi-group sites (K < J): groups per site, device copies of the prefix sums / row limits
    std::vector<int> g_cnt;
    int *site_g0_d;
    int64_t *g_lim_d;
    int multi, ng_max, nt_max, pg;
    int n_max;
    // device buffers
    int64_t *k_lim_d;
    double *X;
    uint8_t *y;
    int *y32;
    double *yd;                     // real responses (Gaussian-likelihood family), else NULL
    int gauss;
    double *Q0, *r0, *Q, *r, *S, *m;
    double *Qi, *ri, *Qi2, *ri2, *dQi, *dri;
    double *cav_Om, *cav_mu;
    double *tilt_mean, *tilt_scatter;
    uint8_t *flags;
    int *iflags;              // [4]
    double *packed, *partial; // sums
    int nslice;
    double *dense_ws;         // global workspace for dense kernels (lazily sized)
    size_t dense_ws_slots;
    // sampler buffers (lazily sized)
    int s_chains, s_nkeep;
    double *draws, *last, *chain_stats, *site_stats, *stack;
    double *team_passes;            // K: row-team passes of the last sampling call per site (layout 7; 0 elsewhere)
    size_t stack_elems;
    int64_t *seeds_d;
    double *dbg;              // [1+P] lp, grad ; [P] theta (test hook)
    int64_t *dbg_seed;
    double *inj;              // injected samples (test hook)
    size_t inj_elems;
    int has_last;
    int nsamp;                // draws per site of the last tilted/moments call
    double last_df;
    hipEvent_t ev0, ev1;
    hipStream_t stream2;            // second queue of a split sampling launch (epx_set_site_split)
    hipEvent_t ev_fork, ev_join;
    int split_n, last_split, n_cu;
    unsigned long long *stamps;
    size_t stamps_n, stamps_last;
    int last_layout;
    int *order_d;
    int order_n;
    int last_segments;
    double *ckpt;
    size_t ckpt_n;
    // piece queue (epx_set_piece_queue): transitions per claim (0: off), predicted work per transition of the sites
    int dyn_len, dyn_has_rate;
    int *dyn_lens_d;          // per-site piece lengths of the queue (device), their host copy below
    std::vector<double> *dyn_rate_h;
    double *dyn_rate;
    int *dyn_words;           // [progress (K) | busy (K)]
    double *sweep_buf;        // damping sweep: target block + ndf x 5 criteria
    size_t sweep_elems;
    double *carry_eps, *carry_metric;   // adapt = carry: K x chains step sizes, K x P diagonal metrics (lazily sized)
    int carry_chains;                   // chains the history was recorded with (0: none)
    double *min_eig;          // force-pd fallback: smallest eigenvalue per site (K)
    // in-library RCCL binding (epx_comm.hip); comm == nullptr: single rank
    void *comm;               // ncclComm_t
    int comm_rank, comm_size;
    int (*comm_ext)(double *, long long, int, void *);   // host transport given by the caller (epx_comm_init_host) ...
    void *comm_ext_user;                                 // ... used instead of RCCL when set
    double *comm_stage;       // device staging of the small host-side collectives
    size_t comm_stage_n;
    int *err_flag;            // device word the sampler kernels set when a hand-off spin gives up
    // epx_sample_piece (test hook): one piece of ONE transition from injected checkpoint records
    int hook_t0;              // > 0 while such a call runs
    const double *hook_in;    // host: K x chains records (csrc/epx_pieces.h layout)
    double *hook_out;         // host: the records the piece leaves at boundary hook_t0 + 1
    // per-transition trace of the sampler (epx_set_trace, test hook): the first trace_sites sites of a sampling call
    int trace_sites, trace_chains, trace_iter, trace_last_sites;     // (trace_last_sites: what the last sampling call recorded)
    double *trace;
    size_t trace_n;
};

