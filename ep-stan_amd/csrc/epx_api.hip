// C-ABI of libepx.so (include/epx.h): context management, host<->device
// copies and kernel launches.  No CPU fallback exists: without a HIP device
// every entry point fails with an error message.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "../../include/epx.h"
#include "epx_kernels.h"
#include "epx_pieces.h"
#include "epx_ctx.h"

using namespace epx;

static thread_local std::string g_err;

int epx_fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

const char *epx_last_error(void) { return g_err.c_str(); }

int epx_device_count(int *count) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail("hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = c;
    return 0;
}

int epx_runtime_info(int device, int *hip_runtime_version, char *arch, int arch_len) {
    int v = 0;
    hipError_t e = hipRuntimeGetVersion(&v);
    if (e != hipSuccess) return fail("hipRuntimeGetVersion: %s", hipGetErrorString(e));
    if (hip_runtime_version) *hip_runtime_version = v;
    if (arch && arch_len > 0) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) return fail("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
        snprintf(arch, (size_t)arch_len, "%s", prop.gcnArchName);
    }
    return 0;
}

int epx_device_synchronize(int device) {
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail("epx_device_synchronize(%d): %s", device, hipGetErrorString(e));
    return 0;
}

int epx_model_dims(int model, int D, int *dphi, int *npar) {
    int d, P;
    const int o = (model >= EPX_M1A_SG && model <= EPX_M5A_SG) ? 1 : 0;      // log sigma in front
    switch (model - (o ? EPX_M1A_SG : 0)) {
    case EPX_M1B_SG: d = D + 1; P = D + 2; break;
    case EPX_M2B_SG: d = 2; P = D + 3; break;
    case EPX_M3B_SG: d = D + 1; P = 2 * D + 2; break;
    case EPX_M4B_SG: case EPX_M5B_SG: d = 2 * D + 2; P = 3 * D + 3; break;
    default: return fail("unknown model id %d", model);
    }
    if (dphi) *dphi = d + o;
    if (npar) *npar = P + o;
    return 0;
}

// room behind the packed site sums for the statistics that ride on the same all-reduce
// (epx_update_trial: 8 sums + 8 maxima x up to 120 ranks) and the three trial flags
enum { STAT_CAP = 8, PACKED_EXTRA = 1024 };
int epx_comm_allreduce_dev(epx_ctx *c, double *buf, size_t n, int op);      // epx_comm.hip

static inline int ld_of(int d) { return d | 1; }
static inline size_t dense_slot_doubles(int d) { return 2 * (size_t)d * ld_of(d) + 4 * (size_t)ld_of(d); }

// LDS or global workspace for `nblocks` concurrent dense blocks
static int dense_ws(epx_ctx *c, int d, int nblocks, DenseWs *ws, size_t *lds_bytes) {
    const size_t bytes = dense_slot_doubles(d) * 8;
    if (bytes + 1024 <= LDS_CAP) { ws->use_lds = 1; ws->global = nullptr; *lds_bytes = bytes; return 0; }
    ws->use_lds = 0; *lds_bytes = 0;
    if (c->dense_ws_slots < (size_t)nblocks) {
        if (c->dense_ws) (void)hipFree(c->dense_ws);
        HIPCHK(dalloc(&c->dense_ws, dense_slot_doubles(d) * nblocks));
        c->dense_ws_slots = nblocks;
    }
    ws->global = c->dense_ws;
    return 0;
}

template <typename K>
static int set_lds(K kern, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail("hipFuncSetAttribute(%zu): %s", bytes, hipGetErrorString(e));
    }
    return 0;
}

static int ctx_create(int device, int model, int K_local, int D, const int64_t *k_lim, const int32_t *g_cnt,
                      const int64_t *g_lim, const double *X, const int32_t *y, const double *yd, epx_ctx **out);

int epx_ctx_create(int device, int model, int K_local, int D, const int64_t *k_lim, const double *X,
                   const int32_t *y, epx_ctx **out) {
    return epx_ctx_create_groups(device, model, K_local, D, k_lim, nullptr, nullptr, X, y, out);
}

int epx_ctx_create_groups(int device, int model, int K_local, int D, const int64_t *k_lim, const int32_t *g_cnt,
                          const int64_t *g_lim, const double *X, const int32_t *y, epx_ctx **out) {
    if (out) *out = nullptr;
    if (model >= EPX_M1A_SG && model <= EPX_M5A_SG)
        return fail("model %d has real responses: use epx_ctx_create_real", model);
    return ctx_create(device, model, K_local, D, k_lim, g_cnt, g_lim, X, y, nullptr, out);
}

int epx_ctx_create_real(int device, int model, int K_local, int D, const int64_t *k_lim, const double *X,
                        const double *y, epx_ctx **out) {
    return epx_ctx_create_real_groups(device, model, K_local, D, k_lim, nullptr, nullptr, X, y, out);
}

int epx_ctx_create_real_groups(int device, int model, int K_local, int D, const int64_t *k_lim, const int32_t *g_cnt,
                               const int64_t *g_lim, const double *X, const double *y, epx_ctx **out) {
    if (out) *out = nullptr;
    if (model < EPX_M1A_SG || model > EPX_M5A_SG)
        return fail("model %d has 0/1 responses: use epx_ctx_create", model);
    return ctx_create(device, model, K_local, D, k_lim, g_cnt, g_lim, X, nullptr, y, out);
}

static int ctx_create(int device, int model, int K_local, int D, const int64_t *k_lim, const int32_t *g_cnt,
                      const int64_t *g_lim, const double *X, const int32_t *y, const double *yd, epx_ctx **out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("no HIP device available: libepx has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("device %d out of range (%d devices)", device, ndev);
    int d, P;
    if (epx_model_dims(model, D, &d, &P)) return -1;
    if (K_local < 1) return fail("K_local must be >= 1");
    if (D < 1) return fail("D must be >= 1");
    HIPCHK(hipSetDevice(device));
    epx_ctx *c = new epx_ctx();
    memset((void *)c, 0, sizeof(int) * 6);
    c->device = device; c->K = K_local; c->D = D; c->d = d; c->P = P;
    c->gauss = yd != nullptr;
    c->model = c->gauss ? model - EPX_M1A_SG : model;       // the kernels take the b-model id plus the family flag
    model = c->model;
    c->k_lim.assign(k_lim, k_lim + K_local + 1);
    c->N = k_lim[K_local] - k_lim[0];
    c->n_max = 0;
    for (int k = 0; k < K_local; ++k) {
        const int64_t n = k_lim[k + 1] - k_lim[k];
        if (n < 1) { delete c; return fail("site %d is empty", k); }
        if (n > c->n_max) c->n_max = (int)n;
    }
    if (k_lim[0] != 0) { delete c; return fail("k_lim[0] must be 0 (rows are rank-local)"); }
    // groups: tiles never straddle two groups, so the tile count of a site depends on them
    c->pg = model == EPX_M1B_SG ? 1 : 1 + D;
    c->multi = g_cnt != nullptr;
    c->ng_max = 1; c->nt_max = 1;
    std::vector<int> g0((size_t)K_local + 1, 0);
    {
        int64_t gi = 0;
        for (int k = 0; k < K_local; ++k) {
            const int ng = g_cnt ? g_cnt[k] : 1;
            if (ng < 1) { delete c; return fail("site %d has %d groups", k, ng); }
            if (ng > c->ng_max) c->ng_max = ng;
            int nt = 0;
            for (int g = 0; g < ng; ++g) {
                const int64_t lo = g_cnt ? g_lim[gi + g] : k_lim[k], hi = g_cnt ? g_lim[gi + g + 1] : k_lim[k + 1];
                if (hi <= lo) { delete c; return fail("group %d of site %d is empty", g, k); }
                nt += (int)((hi - lo + 15) / 16);
            }
            if (g_cnt && (g_lim[gi] != k_lim[k] || g_lim[gi + ng] != k_lim[k + 1])) {
                delete c;
                return fail("the groups of site %d do not cover its rows", k);
            }
            if (nt > c->nt_max) c->nt_max = nt;
            gi += ng;
            g0[k + 1] = (int)gi;
        }
        if (g_cnt) {
            c->g_cnt.assign(g_cnt, g_cnt + K_local);
            c->P = d + c->ng_max * c->pg;           // record stride: the largest site
        }
    }
    c->dense_ws = nullptr; c->dense_ws_slots = 0;
    c->draws = c->last = c->chain_stats = c->site_stats = c->stack = nullptr; c->team_passes = nullptr;
    c->seeds_d = nullptr; c->inj = nullptr; c->inj_elems = 0; c->stack_elems = 0;
    c->s_chains = 0; c->s_nkeep = 0; c->has_last = 0; c->nsamp = 0; c->last_df = 0.0;
    c->stamps = nullptr; c->stamps_n = 0; c->stamps_last = 0;
    HIPCHK(hipStreamCreate(&c->stream));
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    {
        // own priority level = own hardware queue: normal-priority streams share 4 queues round-robin
        // by first use, and two streams on one queue run their kernels one after the other (measured:
        // scripts/probe/concurrent.hip)
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi));
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    c->split_n = 0; c->last_split = 0;
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        c->n_cu = prop.multiProcessorCount;
    }
    const size_t K = K_local, d2 = (size_t)d * d;
    HIPCHK(dalloc(&c->k_lim_d, K + 1));
    if (c->multi) {
        HIPCHK(dalloc(&c->site_g0_d, K + 1));
        HIPCHK(dalloc(&c->g_lim_d, (size_t)g0[K] + 1));
        HIPCHK(hipMemcpy(c->site_g0_d, g0.data(), (K + 1) * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->g_lim_d, g_lim, ((size_t)g0[K] + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    // X carries a zeroed KiB behind the last row: the streaming sampler's row DMA reads full
    // 128-column images
    HIPCHK(dalloc(&c->X, (size_t)c->N * D + 128));
    HIPCHK(hipMemset(c->X + (size_t)c->N * D, 0, 128 * sizeof(double)));
    HIPCHK(dalloc(&c->y, (size_t)c->N));
    HIPCHK(dalloc(&c->y32, (size_t)c->N));
    HIPCHK(dalloc(&c->Q0, d2)); HIPCHK(dalloc(&c->r0, d));
    HIPCHK(dalloc(&c->Q, d2)); HIPCHK(dalloc(&c->r, d));
    HIPCHK(dalloc(&c->S, d2)); HIPCHK(dalloc(&c->m, d));
    HIPCHK(dalloc(&c->Qi, K * d2)); HIPCHK(dalloc(&c->ri, K * d));
    HIPCHK(dalloc(&c->Qi2, K * d2)); HIPCHK(dalloc(&c->ri2, K * d));
    HIPCHK(dalloc(&c->dQi, K * d2)); HIPCHK(dalloc(&c->dri, K * d));
    // (64 columns of zeros behind the last site's cavity precision: the streaming sampler requests its columns a round
    // ahead, nuts_stream.hip)
    HIPCHK(dalloc(&c->cav_Om, K * d2 + (size_t)EPX_OM_PAD_COLS * d)); HIPCHK(dalloc(&c->cav_mu, K * d));
    HIPCHK(hipMemset(c->cav_Om + K * d2, 0, (size_t)EPX_OM_PAD_COLS * d * sizeof(double)));
    HIPCHK(dalloc(&c->tilt_mean, K * d)); HIPCHK(dalloc(&c->tilt_scatter, K * d2));
    HIPCHK(dalloc(&c->flags, K));
    HIPCHK(dalloc(&c->iflags, 4));
    HIPCHK(dalloc(&c->min_eig, K));
    HIPCHK(dalloc(&c->err_flag, 4));
    HIPCHK(hipMemset(c->err_flag, 0, 4 * sizeof(int)));
    HIPCHK(dalloc(&c->dbg, 2 * (size_t)c->P + 1));
    HIPCHK(dalloc(&c->dbg_seed, 1));
    const int len = 2 * (int)(d2 + d);
    c->nslice = K_local >= 64 ? 32 : 1;
    HIPCHK(dalloc(&c->packed, (size_t)len + PACKED_EXTRA));
    HIPCHK(dalloc(&c->partial, (size_t)len * c->nslice));
    HIPCHK(hipMemcpy(c->k_lim_d, k_lim, (K + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->X, X, (size_t)c->N * D * sizeof(double), hipMemcpyHostToDevice));
    c->yd = nullptr;
    if (c->gauss) {
        for (int64_t i = 0; i < c->N; ++i)
            if (!std::isfinite(yd[i])) { epx_ctx_destroy(c); return fail("y[%lld] is not finite", (long long)i); }
        HIPCHK(dalloc(&c->yd, (size_t)c->N));
        HIPCHK(hipMemcpy(c->yd, yd, (size_t)c->N * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(c->y, 0, (size_t)c->N));
        HIPCHK(hipMemset(c->y32, 0, (size_t)c->N * sizeof(int)));
    } else {
        std::vector<uint8_t> yb((size_t)c->N);
        for (int64_t i = 0; i < c->N; ++i) {
            if (y[i] != 0 && y[i] != 1) { epx_ctx_destroy(c); return fail("y[%lld] = %d is not 0/1", (long long)i, y[i]); }
            yb[i] = (uint8_t)y[i];
        }
        HIPCHK(hipMemcpy(c->y, yb.data(), yb.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->y32, y, (size_t)c->N * sizeof(int), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemset(c->Qi, 0, K * d2 * 8)); HIPCHK(hipMemset(c->ri, 0, K * d * 8));
    HIPCHK(hipMemset(c->Qi2, 0, K * d2 * 8)); HIPCHK(hipMemset(c->ri2, 0, K * d * 8));
    HIPCHK(hipMemset(c->dQi, 0, K * d2 * 8)); HIPCHK(hipMemset(c->dri, 0, K * d * 8));
    HIPCHK(hipMemset(c->Q0, 0, d2 * 8)); HIPCHK(hipMemset(c->r0, 0, d * 8));
    HIPCHK(hipMemset(c->Q, 0, d2 * 8)); HIPCHK(hipMemset(c->r, 0, d * 8));
    HIPCHK(hipMemset(c->flags, 0, K));
    *out = c;
    return 0;
}

int epx_ctx_destroy(epx_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->comm || c->comm_ext) (void)epx_comm_destroy(c);
    delete c->dyn_rate_h; c->dyn_rate_h = nullptr;
    void *ptrs[] = {c->dyn_lens_d, c->ckpt, c->dyn_rate, c->dyn_words, c->carry_eps, c->carry_metric, c->min_eig, c->err_flag, c->comm_stage, c->yd, c->site_g0_d, c->g_lim_d, c->sweep_buf, c->order_d, c->k_lim_d, c->X, c->y, c->y32, c->Q0, c->r0, c->Q, c->r, c->S, c->m, c->Qi, c->ri, c->Qi2,
                    c->ri2, c->dQi, c->dri, c->cav_Om, c->cav_mu, c->tilt_mean, c->tilt_scatter,
                    c->flags, c->iflags, c->packed, c->partial, c->dense_ws, c->draws, c->last,
                    c->chain_stats, c->site_stats, c->stack, c->seeds_d, c->dbg, c->dbg_seed, c->inj, c->trace, c->team_passes};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int epx_set_prior(epx_ctx *c, const double *Q0, const double *r0) {
    CTX(c);
    HIPCHK(hipMemcpy(c->Q0, Q0, (size_t)c->d * c->d * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->r0, r0, (size_t)c->d * 8, hipMemcpyHostToDevice));
    return 0;
}

static int check_range(epx_ctx *c, int k0, int count) {
    if (k0 < 0 || count < 1 || k0 + count > c->K) return fail("site range [%d,%d) outside [0,%d)", k0, k0 + count, c->K);
    return 0;
}

static int site_ptrs(epx_ctx *c, int which, double **Q, double **r) {
    switch (which) {
    case EPX_QI: *Q = c->Qi; *r = c->ri; return 0;
    case EPX_QI2: *Q = c->Qi2; *r = c->ri2; return 0;
    case EPX_DQI: *Q = c->dQi; *r = c->dri; return 0;
    }
    return fail("unknown site array %d", which);
}

int epx_set_sites(epx_ctx *c, int which, const double *QF, const double *rF) {
    CTX(c);
    double *Q, *r;
    if (site_ptrs(c, which, &Q, &r)) return -1;
    const size_t K = c->K, d = c->d;
    if (QF) HIPCHK(hipMemcpy(Q, QF, K * d * d * 8, hipMemcpyHostToDevice));
    if (rF) HIPCHK(hipMemcpy(r, rF, K * d * 8, hipMemcpyHostToDevice));
    return 0;
}

int epx_get_sites(epx_ctx *c, int which, double *QF, double *rF) {
    CTX(c);
    double *Q, *r;
    if (site_ptrs(c, which, &Q, &r)) return -1;
    const size_t K = c->K, d = c->d;
    if (which == EPX_QI2) {
        // materialise the proposal Qi + df*dQi of the last trial (method.py:1071-1072)
        hipLaunchKernelGGL(k_axpy, dim3(1024), dim3(256), 0, c->stream, c->Qi2, c->Qi, c->dQi, c->last_df, K * d * d);
        hipLaunchKernelGGL(k_axpy, dim3(64), dim3(256), 0, c->stream, c->ri2, c->ri, c->dri, c->last_df, K * d);
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (QF) HIPCHK(hipMemcpy(QF, Q, K * d * d * 8, hipMemcpyDeviceToHost));
    if (rF) HIPCHK(hipMemcpy(rF, r, K * d * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_set_site(epx_ctx *c, int which, int k, const double *Qh, const double *rh) {
    CTX(c);
    if (check_range(c, k, 1)) return -1;
    double *Q, *r;
    if (site_ptrs(c, which, &Q, &r)) return -1;
    const size_t d = c->d;
    if (Qh) HIPCHK(hipMemcpy(Q + (size_t)k * d * d, Qh, d * d * 8, hipMemcpyHostToDevice));
    if (rh) HIPCHK(hipMemcpy(r + (size_t)k * d, rh, d * 8, hipMemcpyHostToDevice));
    return 0;
}

int epx_get_site(epx_ctx *c, int which, int k, double *Qh, double *rh) {
    CTX(c);
    if (check_range(c, k, 1)) return -1;
    double *Q, *r;
    if (site_ptrs(c, which, &Q, &r)) return -1;
    if (which == EPX_QI2) return fail("epx_get_site: use epx_get_sites for the proposal array");
    const size_t d = c->d;
    if (Qh) HIPCHK(hipMemcpy(Qh, Q + (size_t)k * d * d, d * d * 8, hipMemcpyDeviceToHost));
    if (rh) HIPCHK(hipMemcpy(rh, r + (size_t)k * d, d * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_set_global(epx_ctx *c, const double *Q, const double *r) {
    CTX(c);
    if (Q) HIPCHK(hipMemcpy(c->Q, Q, (size_t)c->d * c->d * 8, hipMemcpyHostToDevice));
    if (r) HIPCHK(hipMemcpy(c->r, r, (size_t)c->d * 8, hipMemcpyHostToDevice));
    return 0;
}
int epx_get_global(epx_ctx *c, double *Q, double *r) {
    CTX(c);
    if (Q) HIPCHK(hipMemcpy(Q, c->Q, (size_t)c->d * c->d * 8, hipMemcpyDeviceToHost));
    if (r) HIPCHK(hipMemcpy(r, c->r, (size_t)c->d * 8, hipMemcpyDeviceToHost));
    return 0;
}

static int launch_cavity(epx_ctx *c, const double *Qs, const double *rs, const double *dQs,
                         const double *drs, double df, int k0, int count) {
    CavityArgs a;
    a.k0 = k0; a.d = c->d; a.ld = ld_of(c->d);
    size_t lds;
    if (dense_ws(c, c->d, count, &a.ws, &lds)) return -1;
    a.Q = c->Q; a.r = c->r; a.Qsite = Qs; a.rsite = rs; a.dQsite = dQs; a.drsite = drs;
    a.site_stride = (size_t)c->d * c->d; a.rsite_stride = c->d; a.df = df;
    a.cav_Om = c->cav_Om; a.cav_mu = c->cav_mu; a.flags = c->flags;
    if (set_lds(k_cavity, lds)) return -1;
    hipLaunchKernelGGL(k_cavity, dim3(count), dim3(256), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

int epx_cavity_batch(epx_ctx *c, int which, int k0, int count, uint8_t *posdef) {
    CTX(c);
    if (check_range(c, k0, count)) return -1;
    int rc;
    if (which == EPX_QI) rc = launch_cavity(c, c->Qi, c->ri, nullptr, nullptr, 0.0, k0, count);
    else if (which == EPX_QI2) rc = launch_cavity(c, c->Qi, c->ri, c->dQi, c->dri, c->last_df, k0, count);
    else return fail("cavity needs EPX_QI or EPX_QI2");
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (posdef) HIPCHK(hipMemcpy(posdef, c->flags + k0, count, hipMemcpyDeviceToHost));
    return 0;
}

int epx_cavity_site(epx_ctx *c, int k, const double *Q, const double *r, const double *Qi,
                    const double *ri, uint8_t *posdef) {
    CTX(c);
    if (check_range(c, k, 1)) return -1;
    const size_t d = c->d;
    // the caller's arrays become the context's state for this site (Worker.cavity
    // aliases Q, r: method.py:286-287)
    HIPCHK(hipMemcpy(c->Q, Q, d * d * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->r, r, d * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->Qi2 + (size_t)k * d * d, Qi, d * d * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->ri2 + (size_t)k * d, ri, d * 8, hipMemcpyHostToDevice));
    if (launch_cavity(c, c->Qi2, c->ri2, nullptr, nullptr, 0.0, k, 1)) return -1;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (posdef) HIPCHK(hipMemcpy(posdef, c->flags + k, 1, hipMemcpyDeviceToHost));
    return 0;
}

int epx_get_cavity(epx_ctx *c, int k, double *Mat, double *vec) {
    CTX(c);
    if (check_range(c, k, 1)) return -1;
    const size_t d = c->d;
    if (Mat) HIPCHK(hipMemcpy(Mat, c->cav_Om + (size_t)k * d * d, d * d * 8, hipMemcpyDeviceToHost));
    if (vec) HIPCHK(hipMemcpy(vec, c->cav_mu + (size_t)k * d, d * 8, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------ sampler
static int norm_opts(const epx_sampler_opts *o, epx_sampler_opts *n) {
    if (!o) return fail("null sampler options");
    *n = *o;
    if (n->chains < 1 || n->chains > 16) return fail("chains must be in 1..16 (got %d)", n->chains);
    if (n->iter < 1) return fail("iter must be >= 1");
    if (n->warmup < 0) n->warmup = n->iter / 2;                 // method.py:157,567-569
    if (n->warmup >= n->iter) return fail("warmup (%d) must be smaller than iter (%d)", n->warmup, n->iter);
    if (n->thin < 1) return fail("thin must be >= 1");
    if (n->max_depth <= 0) n->max_depth = 10;
    if (n->max_depth > MAX_DEPTH_CAP) return fail("max_depth > %d", MAX_DEPTH_CAP);
    if (n->init < 0 || n->init > 2) return fail("bad init mode");
    return 0;
}

static int ensure_sampler_buffers(epx_ctx *c, int chains, int nkeep) {
    const size_t K = c->K, P = c->P;
    if (c->s_chains != chains || c->s_nkeep != nkeep) {
        if (c->draws) (void)hipFree(c->draws);
        if (c->last) (void)hipFree(c->last);
        if (c->chain_stats) (void)hipFree(c->chain_stats);
        if (c->site_stats) (void)hipFree(c->site_stats);
        if (c->seeds_d) (void)hipFree(c->seeds_d);
        if (c->team_passes) (void)hipFree(c->team_passes);
        c->draws = c->last = c->chain_stats = c->site_stats = nullptr; c->seeds_d = nullptr; c->team_passes = nullptr;
        HIPCHK(dalloc(&c->draws, K * chains * nkeep * P));
        HIPCHK(dalloc(&c->last, K * chains * P));
        HIPCHK(dalloc(&c->chain_stats, K * chains * ST_COUNT));
        HIPCHK(dalloc(&c->site_stats, K * 8));
        HIPCHK(dalloc(&c->team_passes, K));
        HIPCHK(hipMemset(c->team_passes, 0, K * 8));
        HIPCHK(dalloc(&c->seeds_d, K));
        if (c->carry_eps) (void)hipFree(c->carry_eps);
        if (c->carry_metric) (void)hipFree(c->carry_metric);
        c->carry_eps = c->carry_metric = nullptr; c->carry_chains = 0;
        HIPCHK(dalloc(&c->carry_eps, K * chains));
        HIPCHK(dalloc(&c->carry_metric, K * P));
        {
            std::vector<double> neg(K * chains, -1.0);       // no history yet
            HIPCHK(hipMemcpy(c->carry_eps, neg.data(), neg.size() * 8, hipMemcpyHostToDevice));
        }
        HIPCHK(hipMemset(c->last, 0, K * chains * P * 8));
        HIPCHK(hipMemset(c->chain_stats, 0, K * chains * ST_COUNT * 8));
        c->s_chains = chains; c->s_nkeep = nkeep; c->has_last = 0;
    }
    return 0;
}

static int pad_dp(int D) { return D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : -1; }

// layout (out): 1 = one block per site (wave = chain, X resident in LDS), 2 = one block per
// (site, chain) with 4 cooperating waves, 3 = streaming (chains in lock step, X through an LDS tile)
static int build_nuts_args(epx_ctx *c, int k0, int count, const epx_sampler_opts &o, NutsArgs &a,
                           int *wpc_out, int *dp_out, int *nv_out, int *layout_out, int stack_sites = 0) {
    if (stack_sites < count) stack_sites = count;       // the HBM tree stack is indexed by site: a split launch sizes it for all
    const int nkeep = (o.iter - o.warmup + o.thin - 1) / o.thin;
    memset(&a, 0, sizeof a);
    a.dyn_tail_div = 1;                                 // (a divisor: never 0, also for launches that do not come from the piece queue)
    a.model = c->model; a.D = c->D; a.d = c->d; a.P = c->P; a.k0 = k0;
    a.gauss = c->gauss; a.yd = c->yd;
    a.chains = o.chains; a.iter = o.iter; a.warmup = o.warmup; a.thin = o.thin; a.nkeep = nkeep;
    a.max_depth = o.max_depth; a.init_mode = o.init;
    a.k_lim = c->k_lim_d; a.X = c->X; a.y = c->y; a.y32 = c->y32; a.cav_Om = c->cav_Om; a.cav_mu = c->cav_mu;
    a.site_g0 = c->site_g0_d; a.g_lim = c->g_lim_d; a.ngmax = c->ng_max; a.ntmax = c->nt_max;
    const int no_spec = o.reserved & 1;           // flag: bookkeeping on the gradient waves (A/B and tests)
    int nv = (c->P + 63) / 64;
    int dp = pad_dp(c->D);
    // resident layouts: enough sites to fill the 256 CUs -> one block per site, else one block
    // per (site, chain) with 4 cooperating waves
    int layout = o.layout;
    // one workgroup per (site, chain) schedules chain by chain (no waiting for the slowest of a site's
    // chains) and has the shorter leapfrog; one workgroup per site packs 4 chains on a CU.  Measured
    // cross-over at the C2 site size, where two layout-2 workgroups share a CU (scripts/tick_occupancy.py:
    // layout 2 ahead by 24 % at 192 sites, 5 % at 256, behind by 4 % at 512); 192 when only one fits.
    bool many = count >= 192;
    if (layout == 0 && !c->multi && dp > 0 && nv <= 2) {
        NutsArgs t = a;
        t.cpb = 1;
        const size_t lds2 = nuts_lds_layout(t, 4, dp, c->n_max);
        if (2 * lds2 <= LDS_CAP) many = count >= 320;
    }
    // layout 4: one block per site, chains in lock step, rows resident in LDS, MFMA products: the
    // home of multi-group sites with small D (measured 1.05-1.4x faster than streaming them);
    // for single-group sites layout 1 is still faster at the C3 site size (132 vs 157 ms per
    // launch), so it is chosen on request only
    bool lock = false;
    // several groups per site: one workgroup per chain, the four gradient waves share the
    // site's groups (nuts_gradient_groups.inc); needs rows, Omega, tree stack and mailbox in LDS
    bool grp = false;
    if (c->multi && dp > 0 && nv <= 2 && !no_spec &&
        (layout == 2 || layout == 0)) {       // measured ahead of the lock-step layouts from 32 to 1024 sites (scripts/ab_kj.py)
        a.grp = 1; a.cpb = 1;
        const size_t lds = nuts_lds_layout(a, 4, dp, c->n_max);
        if (lds <= LDS_CAP && a.om_in_lds && a.stack_in_lds && a.off_spec > 0) grp = true;
        else a.grp = 0;
    }
    if (grp) {
        *wpc_out = 4; *dp_out = dp; *nv_out = nv; *layout_out = 2;
        a.no_spec = 0;
        return 0;
    }
    // (Gaussian-likelihood sites that do not fit the resident forms above are streamed: layout 3)
    if (layout == 2 && c->multi) layout = 0;
    if ((layout == 4 || (layout == 0 && c->multi)) && c->D <= 32 && nv <= 7 && !c->gauss) {
        const int dpl = c->D <= 16 ? 16 : 32;
        size_t lds = nuts_stream_lds_bytes(nv, dpl, c->d, c->ng_max, c->nt_max, c->n_max, 0);
        if (lds <= LDS_CAP) {
            lock = true; layout = 4; dp = dpl;
            a.cpb = 4; a.n_max = c->n_max; a.stack_in_lds = 0; a.om_in_lds = 0;
            const size_t om = (size_t)c->d * c->d * 8;
            if (lds + om <= LDS_CAP) { a.om_in_lds = 1; lds += om; }       // small sites: Omega next to the rows
            a.lds_bytes = (int)lds;
        }
    }
    if (layout == 4 && !lock) layout = 0;
    // layouts 5 / 6 (nuts_duo.hip): row waves + a state wave per chain, LDS hand-offs.  5 = one workgroup per
    // site (up to 4 chains), the default for batches that fill the chip; 6 = one workgroup per chain with 4 row waves
    // Few sites, models with per-coefficient scales (m4b / m5b): layout 6 by default -- its state wave runs from the
    // "view" alone (nuts_duo.hip), 374 against 413 ms per C2 iteration of layout 2; the other models stay on layout 2
    // (only when every chain has a CU of its own: the regime where a launch is as long as its slowest chain)
    const bool auto6 = layout == 0 && !many && c->model >= EPX_M4B_SG && count * o.chains <= c->n_cu;
    if (!lock && !c->multi && !c->gauss && dp > 0 && nv <= 2 && !no_spec &&
        (layout == 5 || layout == 6 || layout == 7 || (layout == 0 && many) || auto6)) {
        // layout 7: the row TEAM of nuts_duo.hip -- four row waves serve the four chains of a site in lock step on the
        // matrix pipe, the state waves are layout 5's; the default for batches that fill the chip since round 3 (layout 5
        // when the TEAM form does not fit the LDS: it pads the rows to whole 16-row tiles)
        int cand[2], ncand = 0;
        if (layout == 6 || auto6) cand[ncand++] = 6;
        else if (layout == 5) cand[ncand++] = 5;
        else { cand[ncand++] = 7; cand[ncand++] = 5; }         // (a request for 7 that does not fit is served by 5)
        for (int ic = 0; ic < ncand; ++ic) {
            const bool six = cand[ic] == 6, seven = cand[ic] == 7;
            const int cpb = six ? 1 : 4, rw = six ? 2 : (seven ? 4 : 1);
            NutsArgs t = a;
            const size_t lds = nuts_duo_lds_layout(t, cpb, rw, dp, c->n_max);
            // (layout 6: one chain per workgroup, the bookkeeping wave's stack lives in LDS or the layout is not used)
            const bool fits = nuts_duo_has(cpb, rw, dp, nv) && lds <= LDS_CAP &&
                              (seven ? (c->n_max + 63) / 64 <= 32 : (c->n_max + 64 * rw - 1) / (64 * rw) <= 64) &&
                              (!six || t.stack_in_lds);
            if (!fits) continue;
            a = t;
            a.err = c->err_flag;
            layout = cand[ic];
            {
                a.stack_stride = nuts_resident_chain_doubles(nv, o.max_depth);
                const size_t need = (size_t)stack_sites * o.chains * a.stack_stride;
                if (c->stack_elems < need) {
                    if (c->stack) (void)hipFree(c->stack);
                    HIPCHK(dalloc(&c->stack, need));
                    c->stack_elems = need;
                }
                a.stack = c->stack;
            }
            a.no_spec = seven && getenv("EPX_NO_LEAN") ? 1 : 0;       // (diagnostic: layout 7 with the full chain rule in every round)
            { const char *y = getenv("EPX_YIELD"); a.yield_cycles = seven ? (y ? atoi(y) : EPX_YIELD_DEFAULT) : 0; }      // (A/B: EPX_YIELD=0 turns it off)
            *wpc_out = rw; *dp_out = dp; *nv_out = nv; *layout_out = layout;
            return 0;
        }
        if (layout == 5 || layout == 6 || layout == 7) layout = 0;
    }
    if (layout == 0) layout = many ? 1 : 2;
    int wpc = 1;
    bool resident = !lock && dp > 0 && nv <= 2 && layout != 3 && !c->multi;      // several groups per site: layouts 3 / 4 only
    if (resident) {
        if (layout == 1) { wpc = 1; a.cpb = o.chains < 4 ? o.chains : 4; }
        else { wpc = 4; a.cpb = 1; }
        const size_t lds = nuts_lds_layout(a, wpc, dp, c->n_max);
        if (lds > LDS_CAP) resident = false;
    }
    if (c->gauss) {
        // Gaussian-likelihood family: the resident kernels are built for their everything-in-LDS forms; any other
        // shape (rows beyond the LDS, D > 32, several groups with many coordinates) is streamed (layout 3)
        const bool ok = resident && a.om_in_lds && (wpc == 1 || (a.stack_in_lds && a.off_spec > 0 && !no_spec));
        if (!ok) resident = false;
    }
    if (!resident && !lock) {
        // rows (or parameters) do not fit the resident kernel: stream X through an LDS tile
        layout = 3;
        if (c->D > 128) return fail("D = %d > 128 is not supported by the streaming sampler", c->D);
        if (nv > 7) return fail("P = %d > 448 sampled coordinates not supported", c->P);
        dp = c->D <= 64 ? 64 : 128;
        a.cpb = 4; wpc = 1;
        a.stack_in_lds = 0; a.om_in_lds = 0;
        a.lds_bytes = (int)nuts_stream_lds_bytes(nv, dp, c->d, c->ng_max, c->nt_max, 0, c->gauss);
        a.off_piece = a.lds_bytes; a.lds_bytes += 16;       // (site, first transition) of a pieced launch's workgroup
        a.err = c->err_flag;
        if ((size_t)a.lds_bytes > LDS_CAP) return fail("streaming sampler needs %d B of LDS", a.lds_bytes);
    }
    if (!a.stack_in_lds) {
        a.stack_stride = layout >= 3 ? nuts_stream_chain_doubles(nv, o.max_depth) : nuts_resident_chain_doubles(nv, o.max_depth);
        const size_t need = (size_t)stack_sites * o.chains * a.stack_stride;
        if (c->stack_elems < need) {
            if (c->stack) (void)hipFree(c->stack);
            HIPCHK(dalloc(&c->stack, need));
            c->stack_elems = need;
        }
        a.stack = c->stack;
    }
    a.no_spec = no_spec;
    *wpc_out = wpc; *dp_out = dp; *nv_out = nv; *layout_out = layout;
    return 0;
}

static int launch_sampler(const NutsArgs &a, int count, int wpc, int dp, int nv, int layout, hipStream_t stream) {
    if (layout == 5 || layout == 6 || layout == 7) return launch_nuts_duo(a, count, a.cpb, a.duo_rw, dp, nv, stream);
    if (layout >= 3) return launch_nuts_stream(a, count, dp, nv, stream);
    return launch_nuts(a, count, wpc, dp, nv, stream);
}

static int run_sampler(epx_ctx *c, int k0, int count, const int64_t *seeds, const epx_sampler_opts &o,
                       double *elapsed_ms, const double *eps_dev = nullptr,
                       const double *inv_e_dev = nullptr, int t_offset = 0) {
    const int nkeep = (o.iter - o.warmup + o.thin - 1) / o.thin;
    if (ensure_sampler_buffers(c, o.chains, nkeep)) return -1;
    if (o.init == EPX_INIT_PREV && !c->has_last) return fail("init=PREV before any sampling call");
    NutsArgs a;
    int wpc, dp, nv, layout;
    if (build_nuts_args(c, k0, count, o, a, &wpc, &dp, &nv, &layout)) return -1;
    a.seeds = c->seeds_d; a.draws = c->draws; a.last = c->last; a.chain_stats = c->chain_stats;
    a.eps_in = eps_dev; a.inv_e_in = inv_e_dev; a.t_offset = t_offset;
    a.team_passes = c->team_passes;
    HIPCHK(hipMemsetAsync(c->team_passes + k0, 0, (size_t)count * 8, c->stream));
    const bool want_carry = (o.reserved & 2) != 0 && !eps_dev;
    if (want_carry) { a.carry_eps = c->carry_eps; a.carry_metric = c->carry_metric; }
    a.order = (c->order_d && c->order_n == count && k0 == 0) ? c->order_d : nullptr;
    c->last_segments = 0;
    c->trace_chains = 0;
    if (c->trace_sites > 0 && !eps_dev) {
        // test hook (epx_set_trace): a record of every transition of the first sites' chains, warm-up included
        const int ts = c->trace_sites < count ? c->trace_sites : count;
        const size_t need = (size_t)ts * o.chains * o.iter * (size_t)(8 + c->P);
        if (c->trace_n < need) {
            if (c->trace) (void)hipFree(c->trace);
            c->trace = nullptr; c->trace_n = 0;
            HIPCHK(dalloc(&c->trace, need));
            c->trace_n = need;
        }
        HIPCHK(hipMemsetAsync(c->trace, 0, need * 8, c->stream));
        a.trace = c->trace; a.trace_sites = ts;
        c->trace_chains = o.chains; c->trace_iter = o.iter; c->trace_last_sites = ts;
    }
    // Piece queue (epx_set_piece_queue): one workgroup per piece, sites claimed by largest remaining predicted work
    const bool hook = c->hook_t0 > 0;          // epx_sample_piece: ONE transition per site from injected checkpoint records
    bool use_queue = (c->dyn_len > 0 || hook) && (layout == 5 || layout == 7 || layout == 3) && k0 == 0 && count == c->K &&
                     o.chains <= a.cpb && !eps_dev && !a.dbg && (o.layout == 0 || o.layout == layout);
    if (hook && (!use_queue || c->hook_t0 >= o.iter))
        return fail("epx_sample_piece: needs a piece-capable layout (5, 7, 3; got %d) over all sites and 0 < t0 < iter", layout);
    // Piece lengths are per site.  Default: piece_len transitions for every site.  With EPX_EQUAL_WORK_PIECES set (A/B only)
    // a site gets pieces of  piece_len x (mean rate / its rate)  transitions, within [piece_len / 4, 4 piece_len], so that
    // the workgroups of the launch last about equally long -- tried against the 10 % of idle CUs that the piece timeline of
    // a C5-shard launch shows (profiles/r03_stream_piece_timeline.json) and measured SLOWER (54.6 % against 58.1 % of the
    // HBM peak: more pieces re-prime more often and the light sites' long pieces coarsen the end of the launch); kept off.
    std::vector<int> lens_h;
    // Shorter pieces behind 3/4 of a site's run (epx_pieces.h): a quarter of the nominal length for the STREAMING sampler
    // (layout 3; same-box A/B at the C5 shard: +0.7 %, profiles/r05_piece_tail_ab.txt), one length throughout for the resident
    // layouts -- there a piece start re-stages the site's 128 KB of rows, the A/B showed no gain (546.6 / 545.9 / 545.6 /
    // 545.8 site-updates/s) and the extra pieces cost 7 GB of HBM traffic per C3 launch.  EPX_PIECE_TAIL_DIV overrides (A/B).
    int tail_div = layout == 3 ? 4 : 1;
    if (const char *tde = getenv("EPX_PIECE_TAIL_DIV")) { const int v = atoi(tde); tail_div = v > 1 ? v : 1; }
    size_t total_pieces = 0;
    int nb_site = 0;
    if (use_queue) {
        lens_h.assign((size_t)count, hook ? 1 : c->dyn_len);
        if (!hook && getenv("EPX_EQUAL_WORK_PIECES") && c->dyn_has_rate && c->dyn_rate_h && (int)c->dyn_rate_h->size() >= count) {
            double mean = 0.0;
            for (int k = 0; k < count; ++k) mean += (*c->dyn_rate_h)[k];
            mean /= count;
            const int lo = c->dyn_len / 4 > 1 ? c->dyn_len / 4 : 1, hi = 4 * c->dyn_len;
            for (int k = 0; k < count; ++k) {
                int l = (int)std::lround(c->dyn_len * mean / (*c->dyn_rate_h)[k]);
                l = l < lo ? lo : (l > hi ? hi : l);
                lens_h[k] = l > o.iter ? o.iter : l;
            }
        }
        for (int k = 0; k < count; ++k) {
            const int np = piece_boundaries(o.iter, lens_h[k], tail_div);     // (nominal pieces, then shorter ones behind 3/4 of the run: epx_pieces.h)
            total_pieces += np;
            nb_site = np + 1 > nb_site ? np + 1 : nb_site;
        }
    }
    if (use_queue) {
        // A pieced launch keeps tree stack + cold store per WORKGROUP (1 GB at the C5 shard; 4 GB with one region per piece)
        // and a checkpoint record per piece boundary.  If the device cannot give that memory, the launch runs unpieced -- same
        // draws, one workgroup per site -- instead of failing the sampling call.
        // (looping workgroups -- the default -- keep theirs per RESIDENT workgroup: never more than 8 per CU; with one
        // workgroup per piece, EPX_PIECE_GRID, every piece has its own)
        bool looping = !getenv("EPX_PIECE_GRID") && !hook;
#ifdef EPX_STAMPS
        if (!getenv("EPX_PIECE_LOOP")) looping = false;     // (the diagnostic build's records are per workgroup: one piece each unless asked otherwise)
#endif
        const size_t regions = (looping && total_pieces > (size_t)c->n_cu * 8) ? (size_t)c->n_cu * 8 : total_pieces;
        const size_t need_stack = regions * o.chains * a.stack_stride;
        const size_t need_ckpt = (size_t)count * nb_site * o.chains * (size_t)(4 * nv + 1) * 64;
        if (c->stack_elems < need_stack) {
            double *p = nullptr;
            if (dalloc(&p, need_stack) == hipSuccess) {
                if (c->stack) (void)hipFree(c->stack);
                c->stack = p; c->stack_elems = need_stack;
            } else { (void)hipGetLastError(); use_queue = false; }
        }
        if (use_queue && c->ckpt_n < need_ckpt) {
            double *p = nullptr;
            // (EPX_TEST_FAIL_CKPT: the test hook of this fallback -- the allocation counts as refused)
            if (!getenv("EPX_TEST_FAIL_CKPT") && dalloc(&p, need_ckpt) == hipSuccess) {
                if (c->ckpt) (void)hipFree(c->ckpt);
                c->ckpt = p; c->ckpt_n = need_ckpt;
            } else { (void)hipGetLastError(); use_queue = false; }
        }
        // (build_nuts_args took c->stack before the re-allocation above: the launch -- pieced or, when the checkpoint records
        // could not be had, unpieced -- must see the buffer that exists now.  The larger buffer also serves the unpieced form.)
        a.stack = c->stack;
    }
    // (the hook has no unpieced form: if the records could not be had, the call fails instead of running a whole update
    // and copying out of a checkpoint buffer that is missing or too small)
    if (hook && !use_queue) return fail("epx_sample_piece: the checkpoint records / tree stacks of the pieced launch could not be allocated");
    if (use_queue) {
        if (!c->dyn_words) HIPCHK(dalloc(&c->dyn_words, 2 * (size_t)c->K));
        HIPCHK(hipMemsetAsync(c->dyn_words, 0, 2 * (size_t)count * sizeof(int), c->stream));      // (stream-ordered in front of the launch)
        a.dyn_prog = c->dyn_words; a.dyn_busy = c->dyn_words + count;
        a.dyn_rate = (c->dyn_has_rate && !hook) ? c->dyn_rate : nullptr;
        a.dyn_len = hook ? 1 : c->dyn_len; a.dyn_count = count;
        a.dyn_hook = hook ? 1 : 0;
        a.dyn_wait_s = EPX_PIECE_WAIT_S;
        if (const char *we = getenv("EPX_PIECE_WAIT_S")) { const int v = atoi(we); if (v > 0) a.dyn_wait_s = v; }
        if (!c->dyn_lens_d) HIPCHK(dalloc(&c->dyn_lens_d, (size_t)c->K));
        HIPCHK(hipMemcpyAsync(c->dyn_lens_d, lens_h.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));            // (lens_h is a local)
        a.dyn_lens = c->dyn_lens_d; a.dyn_nb = nb_site; a.dyn_tail_div = tail_div;
        a.seg_nwg = (int)total_pieces;                      // at most one workgroup per piece ...
        a.persist = getenv("EPX_PIECE_GRID") ? 0 : 1;       // ... looping ones, as many as the device holds (the launcher cuts seg_nwg down); EPX_PIECE_GRID: the first form, for A/B
        if (hook) {
            // every site stands at transition t0 with the caller's records at that boundary; `count` workgroups of the
            // one-piece-per-workgroup form claim one site each, run ONE transition, leave the record of boundary t0 + 1
            a.seg_nwg = count; a.persist = 0;
            std::vector<int> prog((size_t)count, 2 * c->hook_t0);
            HIPCHK(hipStreamSynchronize(c->stream));        // (the memset above is stream-ordered, this copy is not)
            HIPCHK(hipMemcpy(c->dyn_words, prog.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice));
            const size_t rec = (size_t)(4 * nv + 1) * 64;
            for (int k = 0; k < count; ++k)
                HIPCHK(hipMemcpy(c->ckpt + (((size_t)k * nb_site + c->hook_t0) * o.chains) * rec,
                                 c->hook_in + (size_t)k * o.chains * rec, (size_t)o.chains * rec * 8, hipMemcpyHostToDevice));
        }
#ifdef EPX_STAMPS
        if (!getenv("EPX_PIECE_LOOP")) a.persist = 0;       // (the diagnostic build's records are per workgroup: one piece each unless asked otherwise)
#endif
        a.stack = c->stack;                                  // (tree stack + cold store: one region per piece, sized above)
        a.ckpt = c->ckpt;                                    // (one record per piece boundary of a site)
        a.order = nullptr;
        { const int nominal = hook ? 1 : c->dyn_len; c->last_segments = -((o.iter + nominal - 1) / nominal); }      // (negative: pieces per site of a queued launch, at the nominal length)
    }
    // Split launch (epx_set_site_split): the leading sites of the order -- the ones expected to
    // need the most leapfrogs -- run one workgroup per chain (layout 2, shorter leapfrog) on a
    // second queue while the rest run one workgroup per site (layout 1, more chains per CU).
    // The launch ends with its slowest chain; this takes that chain at the faster tick.
    NutsArgs a2;
    int wpc2 = 0, dp2 = 0, nv2 = 0, n_lead = 0;
    if (!use_queue && o.layout == 0 && (layout == 1 || layout == 5 || layout == 7) && a.order && c->split_n > 0 && !eps_dev) {
        n_lead = c->split_n < count ? c->split_n : count - 1;
        const int cap = c->n_cu / (2 * o.chains);          // at most half of the CUs for the lead sites
        if (n_lead > cap) n_lead = cap;
        if (n_lead > 0) {
            epx_sampler_opts o2 = o;
            o2.layout = 2;
            int layout2;
            if (build_nuts_args(c, k0, n_lead, o2, a2, &wpc2, &dp2, &nv2, &layout2, count)) return -1;
            if (layout2 != 2) n_lead = 0;
            else {
                a2.seeds = c->seeds_d; a2.draws = c->draws; a2.last = c->last; a2.chain_stats = c->chain_stats;
                a2.eps_in = nullptr; a2.inv_e_in = nullptr; a2.t_offset = t_offset;
                a2.carry_eps = a.carry_eps; a2.carry_metric = a.carry_metric;
                a2.order = a.order;
                a2.trace = a.trace; a2.trace_sites = a.trace_sites;      // (records are keyed by the REAL site, through `order`, in both launches)
                a.order = a.order + n_lead;
            }
        }
    }
    c->last_split = n_lead;
#ifdef EPX_STAMPS
    n_lead = 0; a.order = (c->order_d && c->order_n == count && k0 == 0) ? c->order_d : nullptr; c->last_split = 0;
    {
        const int nblk = use_queue ? a.seg_nwg : count * ((o.chains + a.cpb - 1) / a.cpb);      // (a pieced launch: one workgroup per piece)
        // (two records of 8 sums per workgroup: the roles' shares, then the phases of the row team's pass)
        // (behind the three records of every workgroup: two records of the first diagnostic form, then six of histograms)
        if (c->stamps_n < (size_t)3 * nblk + 8) {
            if (c->stamps) (void)hipFree(c->stamps);
            HIPCHK(dalloc(&c->stamps, (size_t)(3 * nblk + 8) * 8));
            c->stamps_n = 3 * nblk + 8;
        }
        HIPCHK(hipMemset(c->stamps, 0, (size_t)(3 * nblk + 8) * 64));
        a.stamps = c->stamps; a.stamps_nrec = nblk;
        c->stamps_last = 3 * nblk + 8;
    }
#endif
    HIPCHK(hipMemcpyAsync(c->seeds_d, seeds, (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    int rc = 0;
    if (n_lead > 0) {
        HIPCHK(hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        rc = launch_sampler(a2, n_lead, wpc2, dp2, nv2, 2, c->stream2);
        HIPCHK(hipEventRecord(c->ev_join, c->stream2));
    }
    if (rc == 0) rc = launch_sampler(a, count - n_lead, wpc, dp, nv, layout, c->stream);
    if (n_lead > 0) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    if (rc != 0) return fail("NUTS kernel launch failed (%d: %s)", rc, rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported shape");
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    RhatArgs ra;
    ra.k0 = k0; ra.chains = o.chains; ra.nkeep = nkeep; ra.P = c->P;
    ra.site_g0 = c->site_g0_d; ra.d = c->d; ra.pg = c->pg;
    ra.draws = c->draws; ra.chain_stats = c->chain_stats; ra.site_stats = c->site_stats;
    hipLaunchKernelGGL(k_site_stats, dim3(count), dim3(128), 0, c->stream, ra);
    HIPCHK(hipGetLastError());
    if (!eps_dev && !a.dbg) {
        // history for `adapt = carry` (cheap: one pass over the kept draws); written after every real sampling call
        CarryArgs ca;
        ca.k0 = k0; ca.chains = o.chains; ca.nkeep = nkeep; ca.P = c->P;
        ca.draws = c->draws; ca.chain_stats = c->chain_stats; ca.carry_eps = c->carry_eps; ca.carry_metric = c->carry_metric;
        hipLaunchKernelGGL(k_carry_update, dim3(count), dim3(128), 0, c->stream, ca);
        HIPCHK(hipGetLastError());
        c->carry_chains = o.chains;
    }
    c->has_last = 1;
    c->last_layout = layout;
    c->nsamp = o.chains * nkeep;
    int herr = 0;
    {
        // (no return between a queued copy into this frame and the synchronisation)
        hipError_t e = hipSuccess;
        if (layout == 5 || layout == 6 || layout == 7 || use_queue) e = hipMemcpyAsync(&herr, c->err_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        const hipError_t es = hipStreamSynchronize(c->stream);
        HIPCHK(e);
        HIPCHK(es);
    }
    if (hook && c->hook_out && !herr) {
        const size_t rec = (size_t)(4 * nv + 1) * 64;
        for (int k = 0; k < count; ++k)
            HIPCHK(hipMemcpy(c->hook_out + (size_t)k * o.chains * rec,
                             c->ckpt + (((size_t)k * nb_site + c->hook_t0 + 1) * o.chains) * rec, (size_t)o.chains * rec * 8, hipMemcpyDeviceToHost));
    }
    if (herr) {
        HIPCHK(hipMemset(c->err_flag, 0, sizeof(int)));
        return fail("sampler: a hand-off between the waves of a chain timed out (code %d); the draws of this call are void", herr);
    }
    if (elapsed_ms) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        *elapsed_ms = ms;
    }
    return 0;
}

static int launch_moments(epx_ctx *c, int k0, int count, const double *draws, long site0,
                          long stride_site, long stride_s, long stride_i, int S, int prec_estim) {
    if (prec_estim != EPX_PREC_SAMPLE && prec_estim != EPX_PREC_OLSE) return fail("bad prec_estim %d", prec_estim);
    if (S < c->d) return fail("fewer draws (%d) than dimensions (%d)", S, c->d);   // method.py:427 raises ValueError
    MomentArgs a;
    a.k0 = k0; a.d = c->d; a.ld = ld_of(c->d); a.S = S; a.prec_estim = prec_estim;
    size_t lds;
    if (dense_ws(c, c->d, count, &a.ws, &lds)) return -1;
    a.draws = draws; a.draws_site0 = site0; a.stride_site = stride_site; a.stride_s = stride_s; a.stride_i = stride_i;
    a.Q = c->Q; a.r = c->r; a.dQi = c->dQi; a.dri = c->dri;
    a.tilt_mean = c->tilt_mean; a.tilt_scatter = c->tilt_scatter; a.flags = c->flags;
    if (set_lds(k_moments, lds)) return -1;
    hipLaunchKernelGGL(k_moments, dim3(count), dim3(256), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

int epx_sample_batch(epx_ctx *c, int k0, int count, const int64_t *seeds, const epx_sampler_opts *opts,
                     double *stats, double *elapsed_ms) {
    CTX(c);
    if (check_range(c, k0, count)) return -1;
    epx_sampler_opts o;
    if (norm_opts(opts, &o)) return -1;
    if (run_sampler(c, k0, count, seeds, o, elapsed_ms)) return -1;
    if (stats) HIPCHK(hipMemcpy(stats, c->site_stats, (size_t)count * 8 * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_sample_piece(epx_ctx *c, const int64_t *seeds, const epx_sampler_opts *opts, int t0,
                     const double *records_in, double *records_out) {
    CTX(c);
    epx_sampler_opts o;
    if (norm_opts(opts, &o)) return -1;
    if (t0 <= 0 || !records_in) return fail("epx_sample_piece: t0 > 0 and the records of that boundary are required");
    c->hook_t0 = t0; c->hook_in = records_in; c->hook_out = records_out;
    const int rc = run_sampler(c, 0, c->K, seeds, o, nullptr);
    c->hook_t0 = 0; c->hook_in = nullptr; c->hook_out = nullptr;
    return rc ? -1 : 0;
}

int epx_tilted_batch(epx_ctx *c, int k0, int count, const int64_t *seeds, const epx_sampler_opts *opts,
                     int prec_estim, uint8_t *posdef, double *stats, double *elapsed_ms) {
    CTX(c);
    if (check_range(c, k0, count)) return -1;
    epx_sampler_opts o;
    if (norm_opts(opts, &o)) return -1;
    if (run_sampler(c, k0, count, seeds, o, elapsed_ms)) return -1;
    const int nkeep = c->s_nkeep;
    // native draw layout: (site, chain, keep, P) -> draw s = chain*nkeep + keep is contiguous
    if (launch_moments(c, k0, count, c->draws, k0, (long)o.chains * nkeep * c->P, c->P, 1,
                       o.chains * nkeep, prec_estim)) return -1;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (posdef) HIPCHK(hipMemcpy(posdef, c->flags + k0, count, hipMemcpyDeviceToHost));
    if (stats) HIPCHK(hipMemcpy(stats, c->site_stats, (size_t)count * 8 * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_moments_batch(epx_ctx *c, int k0, int count, const double *samples, int S, int prec_estim,
                      uint8_t *posdef) {
    CTX(c);
    if (check_range(c, k0, count)) return -1;
    const size_t need = (size_t)count * S * c->d;
    if (c->inj_elems < need) {
        if (c->inj) (void)hipFree(c->inj);
        HIPCHK(dalloc(&c->inj, need));
        c->inj_elems = need;
    }
    HIPCHK(hipMemcpy(c->inj, samples, need * 8, hipMemcpyHostToDevice));
    // (S, d) F-order per site: element (s, i) at i*S + s
    if (launch_moments(c, k0, count, c->inj, 0, (long)S * c->d, 1, S, S, prec_estim)) return -1;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->nsamp = S;
    if (posdef) HIPCHK(hipMemcpy(posdef, c->flags + k0, count, hipMemcpyDeviceToHost));
    return 0;
}

int epx_get_tilted(epx_ctx *c, int k, double *Mat, double *vec, int *nsamp) {
    CTX(c);
    if (check_range(c, k, 1)) return -1;
    const size_t d = c->d;
    if (Mat) HIPCHK(hipMemcpy(Mat, c->tilt_scatter + (size_t)k * d * d, d * d * 8, hipMemcpyDeviceToHost));
    if (vec) HIPCHK(hipMemcpy(vec, c->tilt_mean + (size_t)k * d, d * 8, hipMemcpyDeviceToHost));
    if (nsamp) *nsamp = c->nsamp;
    return 0;
}

int epx_num_draws(epx_ctx *c, int *S) {
    if (!c) return fail("null context");
    *S = c->s_chains * c->s_nkeep;
    return 0;
}

int epx_get_draws(epx_ctx *c, int k, int all_params, double *out) {
    CTX(c);
    if (check_range(c, k, 1)) return -1;
    if (!c->draws) return fail("no draws yet");
    const size_t S = (size_t)c->s_chains * c->s_nkeep, P = c->P;
    std::vector<double> tmp(S * P);
    HIPCHK(hipMemcpy(tmp.data(), c->draws + (size_t)k * S * P, S * P * 8, hipMemcpyDeviceToHost));
    const size_t ncol = all_params ? P : (size_t)c->d;
    for (size_t j = 0; j < ncol; ++j)
        for (size_t s = 0; s < S; ++s) out[j * S + s] = tmp[s * P + j];      // (S, ncol) F-order
    return 0;
}

int epx_nuts_transitions(epx_ctx *c, int k0, int count, const int64_t *seeds, int chains, int nt,
                         int t_offset, int layout, const double *q0, const double *eps,
                         const double *inv_e, double *q_out, double *chain_stats) {
    CTX(c);
    if (check_range(c, k0, count)) return -1;
    epx_sampler_opts o;
    memset(&o, 0, sizeof o);
    o.chains = chains; o.iter = nt; o.warmup = 0; o.thin = 1; o.init = EPX_INIT_PREV; o.max_depth = 10;
    o.layout = layout;
    epx_sampler_opts on;
    if (norm_opts(&o, &on)) return -1;
    if (ensure_sampler_buffers(c, chains, nt)) return -1;
    const size_t P = c->P, nc = (size_t)count * chains;
    HIPCHK(hipMemcpy(c->last + (size_t)k0 * chains * P, q0, nc * P * 8, hipMemcpyHostToDevice));
    c->has_last = 1;
    double *eps_d, *inv_d;
    HIPCHK(dalloc(&eps_d, nc));
    HIPCHK(dalloc(&inv_d, nc * P));
    HIPCHK(hipMemcpy(eps_d, eps, nc * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(inv_d, inv_e, nc * P * 8, hipMemcpyHostToDevice));
    int rc = run_sampler(c, k0, count, seeds, on, nullptr, eps_d, inv_d, t_offset);
    (void)hipFree(eps_d); (void)hipFree(inv_d);
    if (rc) return rc;
    if (q_out) HIPCHK(hipMemcpy(q_out, c->draws + (size_t)k0 * chains * nt * P, nc * nt * P * 8, hipMemcpyDeviceToHost));
    if (chain_stats) HIPCHK(hipMemcpy(chain_stats, c->chain_stats + (size_t)k0 * chains * ST_COUNT,
                                      nc * ST_COUNT * 8, hipMemcpyDeviceToHost));
    return 0;
}

#ifdef EPX_STAMPS
// diagnostic build only: cycle sums of the last sampler launch, (nblocks, 8)
extern "C" int epx_dbg_get_stamps(epx_ctx *c, unsigned long long *out, int max_blocks) {
    CTX(c);
    size_t nb = c->stamps_last < (size_t)max_blocks ? c->stamps_last : (size_t)max_blocks;
    HIPCHK(hipMemcpy(out, c->stamps, nb * 64, hipMemcpyDeviceToHost));
    return (int)nb;
}
#endif

int epx_set_site_order(epx_ctx *c, const int32_t *order, int count) {
    CTX(c);
    if (!order || count <= 0) { c->order_n = 0; return 0; }
    if (count > c->K) return fail("site order has %d entries for %d sites", count, c->K);
    std::vector<char> seen((size_t)count, 0);
    for (int i = 0; i < count; ++i) {
        if (order[i] < 0 || order[i] >= count || seen[order[i]]) return fail("site order is not a permutation of 0..%d", count - 1);
        seen[order[i]] = 1;
    }
    if (!c->order_d) HIPCHK(dalloc(&c->order_d, (size_t)c->K));
    HIPCHK(hipMemcpy(c->order_d, order, (size_t)count * sizeof(int), hipMemcpyHostToDevice));
    c->order_n = count;
    return 0;
}

int epx_set_piece_queue(epx_ctx *c, int piece_len, const double *rate) {
    CTX(c);
    if (piece_len <= 0) { c->dyn_len = 0; return 0; }
    if (!c->dyn_words) HIPCHK(dalloc(&c->dyn_words, 2 * (size_t)c->K));
    c->dyn_has_rate = rate != nullptr;
    if (rate) {
        for (int k = 0; k < c->K; ++k)
            if (!(rate[k] > 0.0) || !std::isfinite(rate[k])) return fail("predicted work of site %d is not positive", k);
        if (!c->dyn_rate) HIPCHK(dalloc(&c->dyn_rate, (size_t)c->K));
        HIPCHK(hipMemcpy(c->dyn_rate, rate, (size_t)c->K * 8, hipMemcpyHostToDevice));
        if (!c->dyn_rate_h) c->dyn_rate_h = new std::vector<double>();
        c->dyn_rate_h->assign(rate, rate + c->K);
    }
    c->dyn_len = piece_len;
    return 0;
}

int epx_set_trace(epx_ctx *c, int sites) {
    CTX(c);
    if (sites < 0 || sites > c->K) return fail("trace of %d sites outside 0..%d", sites, c->K);
    c->trace_sites = sites;
    return 0;
}

int epx_get_team_passes(epx_ctx *c, int k0, int count, double *out) {
    CTX(c);
    if (!out) return fail("null argument");
    if (k0 < 0 || count < 0 || k0 + count > c->K) return fail("sites %d..%d outside 0..%d", k0, k0 + count, c->K);
    if (!c->team_passes) return fail("no sampling call yet");
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->team_passes + k0, (size_t)count * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_get_trace(epx_ctx *c, double *out, long long n_out) {
    CTX(c);
    if (!c->trace || c->trace_chains <= 0) return fail("no trace: epx_set_trace before the sampling call");
    const int ts = c->trace_last_sites;
    const size_t n = (size_t)ts * c->trace_chains * c->trace_iter * (size_t)(8 + c->P);
    if ((long long)n != n_out) return fail("trace has %zu doubles, the caller expects %lld", n, n_out);
    HIPCHK(hipMemcpy(out, c->trace, n * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_last_segments(epx_ctx *c) {
    if (!c) return fail("null context");
    return c->last_segments;
}

int epx_set_site_split(epx_ctx *c, int n_lead) {
    CTX(c);
    if (n_lead < 0 || n_lead > c->K) return fail("site split %d outside 0..%d", n_lead, c->K);
    c->split_n = n_lead;
    return 0;
}

int epx_last_split(epx_ctx *c) {
    if (!c) return fail("null context");
    return c->last_split;
}

int epx_cu_count(epx_ctx *c) {
    if (!c) return fail("null context");
    return c->n_cu;
}

int epx_last_layout(epx_ctx *c) {
    if (!c) return fail("null context");
    return c->last_layout;
}

int epx_get_chain_stats(epx_ctx *c, int k0, int count, double *out) {
    CTX(c);
    if (check_range(c, k0, count)) return -1;
    if (!c->chain_stats) return fail("no sampling call yet");
    HIPCHK(hipMemcpy(out, c->chain_stats + (size_t)k0 * c->s_chains * ST_COUNT,
                     (size_t)count * c->s_chains * ST_COUNT * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_get_adapt(epx_ctx *c, int k, double *eps, double *metric) {
    CTX(c);
    if (check_range(c, k, 1)) return -1;
    if (!c->carry_chains) return fail("no sampling call yet");
    if (eps) HIPCHK(hipMemcpy(eps, c->carry_eps + (size_t)k * c->s_chains, (size_t)c->s_chains * 8, hipMemcpyDeviceToHost));
    if (metric) HIPCHK(hipMemcpy(metric, c->carry_metric + (size_t)k * c->P, (size_t)c->P * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_logdensity_grad(epx_ctx *c, int k, const double *theta, double *lp, double *grad) {
    return epx_logdensity_grad_layout(c, k, theta, 2, lp, grad);
}

int epx_logdensity_grad_layout(epx_ctx *c, int k, const double *theta, int layout_req, double *lp, double *grad) {
    CTX(c);
    if (check_range(c, k, 1)) return -1;
    // the sampler kernel itself evaluates the initial point and stops (NutsArgs::dbg)
    epx_sampler_opts o;
    memset(&o, 0, sizeof o);
    o.chains = 1; o.iter = 2; o.warmup = 1; o.thin = 1; o.init = EPX_INIT_PREV; o.max_depth = 10; o.layout = layout_req;
    NutsArgs a;
    int wpc, dp, nv, layout;
    if (build_nuts_args(c, k, 1, o, a, &wpc, &dp, &nv, &layout)) return -1;
    const size_t P = c->P;
    HIPCHK(hipMemcpy(c->dbg + P + 1, theta, P * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(c->dbg_seed, 0, sizeof(int64_t)));
    a.seeds = c->dbg_seed;
    a.last = c->dbg + P + 1 - (size_t)k * P;      // kernel reads last[(k*chains + chain)*P], chains = 1
    a.dbg = c->dbg;
    int rc = launch_sampler(a, 1, wpc, dp, nv, layout, c->stream);
    if (rc != 0) return fail("NUTS kernel launch failed (%d)", rc);
    c->last_layout = layout;
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<double> outv(P + 1);
    HIPCHK(hipMemcpy(outv.data(), c->dbg, (P + 1) * 8, hipMemcpyDeviceToHost));
    *lp = outv[0];
    memcpy(grad, outv.data() + 1, P * 8);
    return 0;
}

// ------------------------------------------------------------- global update
int epx_packed_len(epx_ctx *c, int *len) {
    if (!c) return fail("null context");
    *len = 2 * (c->d * c->d + c->d);
    return 0;
}

int epx_site_sums(epx_ctx *c, double *packed_host, double *packed_dev) {
    CTX(c);
    SumArgs a;
    a.K = c->K; a.d = c->d; a.len = 2 * (c->d * c->d + c->d); a.nslice = c->nslice;
    a.Qi = c->Qi; a.ri = c->ri; a.dQi = c->dQi; a.dri = c->dri;
    a.partial = c->partial; a.out = packed_dev ? packed_dev : c->packed;
    const int nb = (a.len + 255) / 256;
    hipLaunchKernelGGL(k_site_sums_partial, dim3(nb, a.nslice), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_site_sums_final, dim3(nb), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (packed_host) HIPCHK(hipMemcpy(packed_host, a.out, (size_t)a.len * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_mix_sums(epx_ctx *c, double *out) {
    CTX(c);
    if (!c->nsamp) return fail("no tilted moments yet");
    SumArgs a;
    a.K = c->K; a.d = c->d; a.len = 2 * c->d * c->d + c->d; a.nslice = c->nslice;
    a.Qi = c->tilt_scatter; a.ri = c->tilt_mean; a.dQi = nullptr; a.dri = nullptr;
    a.partial = c->partial; a.out = c->packed;
    const int nb = (a.len + 255) / 256;
    hipLaunchKernelGGL(k_mix_partial, dim3(nb, a.nslice), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_site_sums_final, dim3(nb), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, a.out, (size_t)a.len * 8, hipMemcpyDeviceToHost));
    return 0;
}

static int launch_global(epx_ctx *c, const double *packed_dev, double df, int want_moments,
                         const double *tgt = nullptr, double *crit = nullptr) {
    GlobalArgs a;
    a.tgt = tgt; a.crit = crit;
    a.d = c->d; a.ld = ld_of(c->d); a.want_moments = want_moments;
    size_t lds;
    if (dense_ws(c, c->d, c->K, &a.ws, &lds)) return -1;
    a.packed = packed_dev; a.Q0 = c->Q0; a.r0 = c->r0; a.df = df;
    a.Q = c->Q; a.r = c->r; a.S = c->S; a.m = c->m; a.flag = c->iflags;
    if (set_lds(k_global, lds)) return -1;
    hipLaunchKernelGGL(k_global, dim3(1), dim3(256), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

int epx_damped_trial(epx_ctx *c, double df, const double *packed_host, const double *packed_dev,
                     int *global_pd, int *cav_pd, int *first_bad) {
    CTX(c);
    const double *pk = packed_dev;
    if (packed_host) {
        HIPCHK(hipMemcpyAsync(c->packed, packed_host, (size_t)2 * (c->d * c->d + c->d) * 8,
                              hipMemcpyHostToDevice, c->stream));
        pk = c->packed;
    }
    if (!pk) pk = c->packed;       // sums of the last epx_site_sums on this rank
    c->last_df = df;
    if (launch_global(c, pk, df, 0)) return -1;
    int h[4] = {0, 0, -1, 0};
    HIPCHK(hipMemcpyAsync(h, c->iflags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *global_pd = h[0];
    *cav_pd = 0;
    if (first_bad) *first_bad = -1;
    if (!h[0]) return 0;
    if (launch_cavity(c, c->Qi, c->ri, c->dQi, c->dri, df, 0, c->K)) return -1;
    hipLaunchKernelGGL(k_all_flags, dim3(1), dim3(256), 0, c->stream, c->flags, 0, c->K, c->iflags + 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h + 1, c->iflags + 1, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *cav_pd = h[1];
    if (first_bad) *first_bad = h[2];
    return 0;
}

int epx_update_trial(epx_ctx *c, double df, int reduce_sums, int site_base, double *stat_sum, int n_sum,
                     double *stat_max, int n_max, int want_moments, int *global_pd, int *cav_pd,
                     int64_t *first_bad, double *S, double *m) {
    CTX(c);
    if (n_sum < 0 || n_sum > STAT_CAP || n_max < 0 || n_max > STAT_CAP) return fail("at most %d statistics of a kind", STAT_CAP);
    const bool bound = c->comm || c->comm_ext;
    const int nr = bound ? c->comm_size : 1, rank = bound ? c->comm_rank : 0;
    const int len = 2 * (c->d * c->d + c->d);
    const int next = n_sum + n_max * nr;
    if (next + 3 > PACKED_EXTRA) return fail("too many ranks (%d) for the statistics block", nr);
    double *ext_d = c->packed + len, *trial_d = c->packed + len + PACKED_EXTRA - 3;
    std::vector<double> ext((size_t)next + 1, 0.0);
    if (reduce_sums) {
        // statistics of this rank: sums as they are, maxima in the rank's own slots (the others stay 0,
        // so the sum over ranks is an all-gather and the maximum is taken on the host)
        for (int i = 0; i < n_sum; ++i) ext[i] = stat_sum[i];
        for (int i = 0; i < n_max; ++i) ext[n_sum + (size_t)rank * n_max + i] = stat_max[i];
        if (next) HIPCHK(hipMemcpyAsync(ext_d, ext.data(), (size_t)next * 8, hipMemcpyHostToDevice, c->stream));
        SumArgs a;
        a.K = c->K; a.d = c->d; a.len = len; a.nslice = c->nslice;
        a.Qi = c->Qi; a.ri = c->ri; a.dQi = c->dQi; a.dri = c->dri;
        a.partial = c->partial; a.out = c->packed;
        const int nb = (a.len + 255) / 256;
        hipLaunchKernelGGL(k_site_sums_partial, dim3(nb, a.nslice), dim3(256), 0, c->stream, a);
        hipLaunchKernelGGL(k_site_sums_final, dim3(nb), dim3(256), 0, c->stream, a);
        HIPCHK(hipGetLastError());
        // the ONE reduction of the iteration (method.py:1073-1074; affine in df, so it serves every trial)
        if (epx_comm_allreduce_dev(c, c->packed, (size_t)len + next, EPX_OP_SUM)) return -1;
    }
    c->last_df = df;
    if (launch_global(c, c->packed, df, want_moments)) return -1;
    if (launch_cavity(c, c->Qi, c->ri, c->dQi, c->dri, df, 0, c->K)) return -1;
    hipLaunchKernelGGL(k_trial_flags, dim3(1), dim3(256), 0, c->stream, c->flags, c->K, site_base, c->iflags, trial_d);
    HIPCHK(hipGetLastError());
    if (epx_comm_allreduce_dev(c, trial_d, 3, EPX_OP_MIN)) return -1;
    double tf[3] = {0, 0, 0};
    {
        // Once a device-to-host copy into this frame's buffers is queued, nothing returns before the stream has been
        // synchronised: a copy that lands after an early return would write into a dead stack frame
        hipError_t e = hipMemcpyAsync(tf, trial_d, sizeof tf, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && reduce_sums && next) e = hipMemcpyAsync(ext.data(), ext_d, (size_t)next * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && want_moments && S) e = hipMemcpyAsync(S, c->S, (size_t)c->d * c->d * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && want_moments && m) e = hipMemcpyAsync(m, c->m, (size_t)c->d * 8, hipMemcpyDeviceToHost, c->stream);
        const hipError_t es = hipStreamSynchronize(c->stream);           // the one synchronisation of the trial
        HIPCHK(e);
        HIPCHK(es);
    }
    *global_pd = tf[0] != 0.0;
    *cav_pd = tf[1] != 0.0;
    if (first_bad) *first_bad = tf[2] >= 1e17 ? -1 : (int64_t)tf[2];
    if (reduce_sums) {
        for (int i = 0; i < n_sum; ++i) stat_sum[i] = ext[i];
        for (int i = 0; i < n_max; ++i) {
            double v = ext[n_sum + i];
            for (int r = 1; r < nr; ++r) v = std::fmax(v, ext[n_sum + (size_t)r * n_max + i]);
            stat_max[i] = v;
        }
    }
    return 0;
}

int epx_damp_sweep(epx_ctx *c, int ndf, const double *dfs, const double *packed_host, const double *packed_dev,
                   const double *m_target, const double *S_target, double half_logdet_S_target,
                   const double *samp_mean, const double *samp_scatter, int n_samp, double *out) {
    CTX(c);
    if (ndf < 1 || !dfs || !m_target || !S_target || !out) return fail("damp sweep: missing argument");
    const size_t d = c->d, d2 = d * d, ntgt = 2 * (d + d2) + 2;
    const size_t need = ntgt + (size_t)ndf * 5;
    if (c->sweep_elems < need) {
        if (c->sweep_buf) (void)hipFree(c->sweep_buf);
        c->sweep_buf = nullptr; c->sweep_elems = 0;
        HIPCHK(dalloc(&c->sweep_buf, need));
        c->sweep_elems = need;
    }
    std::vector<double> h(ntgt, 0.0);
    memcpy(h.data(), m_target, d * 8);
    memcpy(h.data() + d, S_target, d2 * 8);
    const bool have_samp = samp_mean && samp_scatter && n_samp > 0;
    if (have_samp) { memcpy(h.data() + d + d2, samp_mean, d * 8); memcpy(h.data() + 2 * d + d2, samp_scatter, d2 * 8); }
    h[2 * (d + d2)] = half_logdet_S_target;
    h[2 * (d + d2) + 1] = have_samp ? (double)n_samp : 0.0;
    HIPCHK(hipMemcpyAsync(c->sweep_buf, h.data(), ntgt * 8, hipMemcpyHostToDevice, c->stream));
    const double *pk = packed_dev;
    if (packed_host) {
        HIPCHK(hipMemcpyAsync(c->packed, packed_host, (size_t)2 * (d2 + d) * 8, hipMemcpyHostToDevice, c->stream));
        pk = c->packed;
    }
    if (!pk) pk = c->packed;
    double *crit = c->sweep_buf + ntgt;
    // every trial back to back on the stream, one synchronisation at the end
    for (int i = 0; i < ndf; ++i) {
        if (launch_global(c, pk, dfs[i], 1, c->sweep_buf, crit + (size_t)i * 5)) return -1;
        if (launch_cavity(c, c->Qi, c->ri, c->dQi, c->dri, dfs[i], 0, c->K)) return -1;
        hipLaunchKernelGGL(k_all_flags, dim3(1), dim3(256), 0, c->stream, c->flags, 0, c->K, c->iflags + 1);
        hipLaunchKernelGGL(k_sweep_flag, dim3(1), dim3(64), 0, c->stream, c->iflags + 1, crit + (size_t)i * 5);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(out, crit, (size_t)ndf * 5 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->last_df = dfs[ndf - 1];
    return 0;
}

int epx_accept(epx_ctx *c, double df) {
    CTX(c);
    const size_t K = c->K, d = c->d;
    hipLaunchKernelGGL(k_axpy, dim3(1024), dim3(256), 0, c->stream, c->Qi, c->Qi, c->dQi, df, K * d * d);
    hipLaunchKernelGGL(k_axpy, dim3(64), dim3(256), 0, c->stream, c->ri, c->ri, c->dri, df, K * d);
    HIPCHK(hipGetLastError());
    // no synchronisation: every later use is ordered behind it on the context's stream (the
    // synchronous copies of the accessors run on the legacy default stream, which waits for it)
    return 0;
}

int epx_global_moments(epx_ctx *c, double *S, double *m) {
    CTX(c);
    if (launch_global(c, nullptr, 0.0, 1)) return -1;
    int ok = 0;
    HIPCHK(hipMemcpyAsync(&ok, c->iflags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!ok) return fail("global precision is not positive definite");
    if (S) HIPCHK(hipMemcpy(S, c->S, (size_t)c->d * c->d * 8, hipMemcpyDeviceToHost));
    if (m) HIPCHK(hipMemcpy(m, c->m, (size_t)c->d * 8, hipMemcpyDeviceToHost));
    return 0;
}

int epx_force_pd(epx_ctx *c, double df, double thresh, double min_eig_target, uint8_t *forced) {
    CTX(c);
    ForceArgs a;
    a.d = c->d; a.ld = ld_of(c->d);
    size_t lds;
    if (dense_ws(c, c->d, c->K, &a.ws, &lds)) return -1;
    a.Qi = c->Qi; a.dQi = c->dQi; a.df = df; a.thresh = thresh; a.target = min_eig_target;
    a.forced = c->flags; a.min_eig = c->min_eig;
    if (set_lds(k_force_pd, lds)) return -1;
    hipLaunchKernelGGL(k_force_pd, dim3(c->K), dim3(256), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (forced) HIPCHK(hipMemcpy(forced, c->flags, c->K, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------ stand-alone util ops
static int util_dense_ws(int d, int nb, DenseWs *ws, size_t *lds, double **owned) {
    *owned = nullptr;
    const size_t bytes = dense_slot_doubles(d) * 8;
    if (bytes + 1024 <= LDS_CAP) { ws->use_lds = 1; ws->global = nullptr; *lds = bytes; return 0; }
    ws->use_lds = 0; *lds = 0;
    HIPCHK(dalloc(owned, dense_slot_doubles(d) * nb));
    ws->global = *owned;
    return 0;
}

static int need_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("no HIP device available: libepx has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    return 0;
}

// the stand-alone entry points run on the device they are given and leave the calling thread's
// current device as they found it (a rank of a multi-GPU job keeps its own GPU current)
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int epx_invert_normal_params(int device, int d, int nb, double *A, double *b, int cho_form, int32_t *info) {
    DeviceGuard guard;
    if (need_device(device)) return -1;
    if (d < 1 || nb < 1) return fail("bad sizes");
    InvertArgs a;
    a.d = d; a.ld = ld_of(d); a.cho_form = cho_form;
    size_t lds; double *owned;
    if (util_dense_ws(d, nb, &a.ws, &lds, &owned)) return -1;
    double *Ad, *bd = nullptr; int32_t *infod;
    HIPCHK(dalloc(&Ad, (size_t)nb * d * d));
    HIPCHK(dalloc(&infod, nb));
    HIPCHK(hipMemcpy(Ad, A, (size_t)nb * d * d * 8, hipMemcpyHostToDevice));
    if (b) { HIPCHK(dalloc(&bd, (size_t)nb * d)); HIPCHK(hipMemcpy(bd, b, (size_t)nb * d * 8, hipMemcpyHostToDevice)); }
    a.A = Ad; a.b = bd; a.info = infod;
    if (set_lds(k_invert, lds)) return -1;
    hipLaunchKernelGGL(k_invert, dim3(nb), dim3(256), lds, 0, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(A, Ad, (size_t)nb * d * d * 8, hipMemcpyDeviceToHost));
    if (b) HIPCHK(hipMemcpy(b, bd, (size_t)nb * d * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(info, infod, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    (void)hipFree(Ad); (void)hipFree(infod); if (bd) (void)hipFree(bd); if (owned) (void)hipFree(owned);
    return 0;
}

int epx_olse(int device, int d, int nb, double *S, int n, const double *P, int32_t *info) {
    DeviceGuard guard;
    if (need_device(device)) return -1;
    if (d < 1 || nb < 1) return fail("bad sizes");
    OlseArgs a;
    a.d = d; a.ld = ld_of(d); a.n = n;
    size_t lds; double *owned;
    if (util_dense_ws(d, nb, &a.ws, &lds, &owned)) return -1;
    double *Sd, *Pd = nullptr; int32_t *infod;
    HIPCHK(dalloc(&Sd, (size_t)nb * d * d));
    HIPCHK(dalloc(&infod, nb));
    HIPCHK(hipMemcpy(Sd, S, (size_t)nb * d * d * 8, hipMemcpyHostToDevice));
    if (P) { HIPCHK(dalloc(&Pd, (size_t)nb * d * d)); HIPCHK(hipMemcpy(Pd, P, (size_t)nb * d * d * 8, hipMemcpyHostToDevice)); }
    a.S = Sd; a.P = Pd; a.info = infod;
    if (set_lds(k_olse, lds)) return -1;
    hipLaunchKernelGGL(k_olse, dim3(nb), dim3(256), lds, 0, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(S, Sd, (size_t)nb * d * d * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(info, infod, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    (void)hipFree(Sd); (void)hipFree(infod); if (Pd) (void)hipFree(Pd); if (owned) (void)hipFree(owned);
    return 0;
}

int epx_rng_probe(int device, uint64_t seed, int chain, uint32_t t, uint32_t kind, uint32_t a, uint32_t b,
                  double *out4) {
    DeviceGuard guard;
    if (need_device(device)) return -1;
    double *od;
    HIPCHK(dalloc(&od, 4));
    hipLaunchKernelGGL(k_rng_probe, dim3(1), dim3(64), 0, 0, seed, chain, t, kind, a, b, od);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out4, od, 32, hipMemcpyDeviceToHost));
    (void)hipFree(od);
    return 0;
}
