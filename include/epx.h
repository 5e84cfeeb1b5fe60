/*
 * epx.h -- C ABI of libepx.so: the MI355X (gfx950) engine behind ep-stan's
 * data-parallel EP inner loop.
 *
 * Every entry point replaces one piece of /root/reference/epstan (cited per
 * function as file:line) and is what a ctypes binding in the reference's
 * epstan/method.py would call instead of PyStan + SciPy/LAPACK (INTEGRATION.md
 * shows that binding).  Conventions:
 *   - extern "C", plain pointers and sizes, no C++/torch types;
 *   - every function returns 0 on success, <0 on error; the message is kept
 *     per thread and read with epx_last_error();
 *   - the caller owns host buffers, the library owns device buffers behind an
 *     opaque epx_ctx bound to ONE HIP device; a ctx is not thread-safe; calls
 *     are synchronous on return (the stream is drained);
 *   - all reals are float64; matrices are column-major ("F order") exactly as
 *     the reference lays them out: site arrays (d,d,K) / (d,K) with the site
 *     index slowest (method.py:838-851), so site k is one contiguous d*d block;
 *   - pointers named *_dev are DEVICE addresses (a caller that runs the reduction between
 *     epx_site_sums() and epx_damped_trial() itself, on device memory of its own);
 *   - several GPUs: one context per rank, bound to the others by epx_comm_init() (RCCL inside
 *     the library); epx_update_trial() then performs the iteration's one all-reduce in stream
 *     order.
 */
#ifndef EPX_H
#define EPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct epx_ctx epx_ctx;

/* Site log-density families: the reference's single-group Stan programs
 * experiment/models/m{1,2,3,4,5}b_sg.stan (selected by the basename of the
 * `site_model` path given to Master, method.py:647,672). */
enum epx_model { EPX_M1B_SG = 0, EPX_M2B_SG = 1, EPX_M3B_SG = 2, EPX_M4B_SG = 3, EPX_M5B_SG = 4,
                 /* Gaussian-likelihood family, experiment/models/m{1..5}a_sg.stan: phi = [log sigma | the
                  * b-model's phi], y ~ normal(alpha + X beta, sigma), real responses (epx_ctx_create_real) */
                 EPX_M1A_SG = 5, EPX_M2A_SG = 6, EPX_M3A_SG = 7, EPX_M4A_SG = 8, EPX_M5A_SG = 9 };

/* Worker.PREC_ESTIM_OPTIONS, method.py:163 */
enum epx_prec_estim { EPX_PREC_SAMPLE = 0, EPX_PREC_OLSE = 1 };

/* which site array an accessor addresses (method.py:844-851) */
enum epx_which { EPX_QI = 0, EPX_QI2 = 1, EPX_DQI = 2 };

/* init modes of the sampler: Stan's init='random' / '0' (method.py:159, 579-583)
 * or the last draw of each chain of the previous call (init_prev, :404-406). */
enum epx_init { EPX_INIT_RANDOM = 0, EPX_INIT_ZERO = 1, EPX_INIT_PREV = 2 };

/* Sampler settings = Worker.DEFAULT_STAN_PARAMS (method.py:154-160) plus the
 * Stan 2.17 control defaults the reference leaves untouched. */
typedef struct epx_sampler_opts {
    int32_t chains;      /* default 4 */
    int32_t iter;        /* default 1000; fit.py uses 200 */
    int32_t warmup;      /* <0 means iter/2 (warmup=None, method.py:157,567-569) */
    int32_t thin;        /* default 1 */
    int32_t init;        /* enum epx_init */
    int32_t max_depth;   /* Stan max_treedepth, default 10 */
    int32_t layout;      /* 0 auto, 1 one block per site (rows resident in LDS), 2 one block per
                            (site, chain), 3 streaming (rows through an LDS-DMA ring, chains in lock step;
                            chosen automatically when the rows do not fit LDS or D > 32), 4 lock step with
                            the rows resident in LDS (D <= 32; default for multi-group sites), 5 one block per
                            site with a state wave + a row wave per chain (same draws as 1), 6 one block per
                            chain: state wave, two row waves and a bookkeeping wave, 7 one block per site: a state
                            wave per chain + four row waves that serve the site's four chains in lock step on the
                            matrix pipe (v_mfma_f64_4x4x4; the default for batches that fill the chip since
                            round 3; served by 5 when its padded rows do not fit the LDS) */
    int32_t reserved;    /* flags; bit 0: layout 2 without the speculative bookkeeping wave (same draws,
                            used for A/B measurements and tests);
                            bit 1: `adapt = carry` -- NOT the reference's behaviour (a fresh model.sampling per site
                            update re-adapts from scratch, util.py:716), an opt-in the survey sanctions when it is
                            reported: every chain starts from the step size its previous call ended with and from
                            the site's pooled sample variances of that call as diagonal metric (regularised like
                            Stan's estimate), and warm-up adapts the step size only.  Same target distribution;
                            sites without history (first call, a failed chain) adapt from scratch. */
} epx_sampler_opts;

/* per-site sampler statistics written by epx_tilted_batch (doubles) */
enum epx_site_stat {
    EPX_ST_STEPSIZE = 0,   /* mean over chains of mean stepsize__ (method.py:99-102) */
    EPX_ST_RHAT = 1,       /* max split-Rhat over sampled coordinates (method.py:104) */
    EPX_ST_NLEAP = 2,      /* leapfrogs in trees, all chains */
    EPX_ST_NGRAD = 3,      /* gradient evaluations, all chains */
    EPX_ST_NDIV = 4,       /* divergent post-warm-up transitions */
    EPX_ST_ACCEPT = 5,     /* mean post-warm-up accept_stat */
    EPX_ST_DEPTH = 6,      /* mean post-warm-up tree depth */
    EPX_ST_FAIL = 7,       /* >0: a chain started at a non-finite density */
    EPX_ST_COUNT = 8
};

const char *epx_last_error(void);
int epx_device_count(int *count);
/* HIP runtime version (hipRuntimeGetVersion: major * 10^7 + minor * 10^5 + patch) and the device's gcnArchName.  The
 * loader (ep-stan_amd/_lib.py) keys ONE optimisation on them: the piece hand-off without the L2 write-back fence
 * (csrc/epx_pieces.h) is measured behaviour of gfx950 under ROCm 7.2, not an architectural guarantee -- on any other
 * architecture or runtime the loader takes variants/libepx_fence.so (the build with the fence) when it is there.
 * No counterpart in the reference. */
int epx_runtime_info(int device, int *hip_runtime_version, char *arch, int arch_len);
/* Blocks until every stream of `device` is idle (hipDeviceSynchronize).  Every entry point of a context already returns
 * with its own work finished; this is the bracket a measurement harness puts around a timed region (bench.py) without
 * bringing a second HIP runtime -- PyTorch's bundled one -- into the process. */
int epx_device_synchronize(int device);

/* dphi and number of sampled coordinates P of a model (m*b_sg.stan parameter blocks). */
int epx_model_dims(int model, int D, int *dphi, int *npar);

/*
 * Context = the K_local sites of one rank: replaces Master.__init__'s worker
 * construction and array allocation (method.py:817-851).  X is the rank's
 * (N_local, D) row-major block (C-contiguous, method.py:733), y its 0/1
 * responses, k_lim[K_local+1] the row limits of the sites (method.py:700).
 * Sites are ordered; site k owns rows k_lim[k]..k_lim[k+1].
 */
int epx_ctx_create(int device, int model, int K_local, int D, const int64_t *k_lim,
                   const double *X, const int32_t *y, epx_ctx **out);

/* The same for sites that hold SEVERAL groups of the hierarchical model (K < J, experiment/fit.py:310-324:
 * `Master(..., A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k+1})` with the multi-group programs
 * experiment/models/m{1..5}b.stan).  g_cnt[k] = groups in site k; g_lim = row limits of all groups in site
 * order (sum(g_cnt)+1 entries, rank-local rows): the rows of a group are contiguous and the groups of a site
 * tile its rows -- what util.distribute_groups (util.py:582-608) produces.  Site k then samples
 * dphi + g_cnt[k] * (1 [+ D]) coordinates [phi | eta_1.. | etb_1 .. ]; records of draws / last states use the
 * stride of the largest site.  Runs on the streaming sampler layout. */
int epx_ctx_create_groups(int device, int model, int K_local, int D, const int64_t *k_lim, const int32_t *g_cnt,
                          const int64_t *g_lim, const double *X, const int32_t *y, epx_ctx **out);

/* The same for the Gaussian-likelihood models (EPX_M1A_SG..EPX_M5A_SG; `real y[N]` in
 * experiment/models/m1a_sg.stan:16, simulated in models/m1a.py:150-176): y holds real responses.
 * One group per site.  Sites whose rows, cavity precision (and, for one workgroup per chain, the tree stack) fit
 * the LDS run on the resident kernels, all others (D > 32, rows beyond the LDS) on the streaming layout. */
int epx_ctx_create_real(int device, int model, int K_local, int D, const int64_t *k_lim, const double *X,
                        const double *y, epx_ctx **out);
/* ... with several groups per site (experiment/models/m1a.stan:11-45, `j_ind`; g_cnt / g_lim as in
 * epx_ctx_create_groups): the one-workgroup-per-chain layout (D <= 32, <= 128 coordinates), else streaming. */
int epx_ctx_create_real_groups(int device, int model, int K_local, int D, const int64_t *k_lim, const int32_t *g_cnt,
                               const int64_t *g_lim, const double *X, const double *y, epx_ctx **out);
int epx_ctx_destroy(epx_ctx *ctx);

/* prior natural parameters Q0 (d,d) F-order, r0 (d): method.py:772-797 */
int epx_set_prior(epx_ctx *ctx, const double *Q0, const double *r0);

/* host <-> device copies of the site arrays, F-order (d,d,K_local) / (d,K_local)
 * (method.py:844-851); either pointer may be NULL. */
int epx_set_sites(epx_ctx *ctx, int which, const double *QF, const double *rF);
int epx_get_sites(epx_ctx *ctx, int which, double *QF, double *rF);

/* one site's (d,d) / (d) block of a site array: the NumPy views dQi[:,:,k], dri[:,k]
 * that Worker.tilted / Worker.cavity take (method.py:1012-1023, find_damp.py:139,160) */
int epx_set_site(epx_ctx *ctx, int which, int k, const double *Q, const double *r);
int epx_get_site(epx_ctx *ctx, int which, int k, double *Q, double *r);

/* global approximation Q (d,d), r (d) held by the context (method.py:841-842) */
int epx_set_global(epx_ctx *ctx, const double *Q, const double *r);
int epx_get_global(epx_ctx *ctx, double *Q, double *r);

/*
 * Worker.cavity for sites k0..k0+count (method.py:267-302), batched:
 * Mat_k = Q - A_k, vec_k = Mat_k^-1 (r - a_k), posdef[k] = Cholesky succeeded,
 * where (A_k, a_k) = site array `which`; for which == EPX_QI2 the proposal
 * Qi + df*dQi is formed on the fly (method.py:1071-1072) with the df of the
 * last epx_damped_trial.  The results stay on the device as the sampler's
 * Omega_phi / mu_phi (method.py:221-222).
 */
int epx_cavity_batch(epx_ctx *ctx, int which, int k0, int count, uint8_t *posdef);
/* cavity of ONE site from caller-supplied arrays (the find_damp.py:160 pattern) */
int epx_cavity_site(epx_ctx *ctx, int k, const double *Q, const double *r, const double *Qi,
                    const double *ri, uint8_t *posdef);
/* Worker.Mat / Worker.vec of site k after cavity (phase 1) */
int epx_get_cavity(epx_ctx *ctx, int k, double *Mat, double *vec);

/*
 * Worker.tilted for sites k0..k0+count (method.py:305-475), batched: on-GPU
 * NUTS draws from each site's tilted distribution (replaces _sample_stan,
 * method.py:43-118), then mean / scatter / precision estimate / site delta
 * (:410-458) into the device dQi, dri.  seeds[count] are the per-site Stan
 * seeds of method.py:346.  The global (Q, r) subtracted at :457-458 is the one
 * held by the context.  posdef[count] out; stats[count*EPX_ST_COUNT] out (may
 * be NULL); elapsed_ms out (may be NULL): device time of the sampling kernel.
 */
int epx_tilted_batch(epx_ctx *ctx, int k0, int count, const int64_t *seeds,
                     const epx_sampler_opts *opts, int prec_estim, uint8_t *posdef,
                     double *stats, double *elapsed_ms);
/*
 * TEST HOOK: the moment stage alone on injected draws.  samples is
 * (S, d, count) F-order, i.e. per site an (S,d) column-major block like the
 * `samp` array of method.py:362.
 */
int epx_moments_batch(epx_ctx *ctx, int k0, int count, const double *samples, int S,
                      int prec_estim, uint8_t *posdef);
/* Worker.vec (tilted mean) and Worker.Mat (unnormalised scatter C'C) of site k
 * after tilted (phase 2); nsamp out (method.py:408) */
int epx_get_tilted(epx_ctx *ctx, int k, double *Mat, double *vec, int *nsamp);
/* phi draws of site k in the reference's layout: (S, dphi) F-order, chains
 * concatenated chain-major (util.py:475-484); all sampled coordinates with
 * npar_out = 1: (S, P) F-order. */
int epx_get_draws(epx_ctx *ctx, int k, int all_params, double *out);
int epx_num_draws(epx_ctx *ctx, int *S);

/*
 * Local part of the reduction of method.py:1073-1074:
 * packed = [sum_k Qi (d*d), sum_k ri (d), sum_k dQi (d*d), sum_k dri (d)] over
 * this rank's sites.  Because Q(df) = Q0 + sum Qi + df * sum dQi is affine in
 * df, ONE all-reduce of this buffer per EP iteration serves every damping
 * trial.  Exactly one of packed_host / packed_dev may be non-NULL.
 */
int epx_site_sums(epx_ctx *ctx, double *packed_host, double *packed_dev);
int epx_packed_len(epx_ctx *ctx, int *len);

/*
 * One trial of the damping loop (method.py:1067-1143) with damping factor df:
 * forms Q, r from the (all-reduced) packed sums, Cholesky-checks Q
 * (global_pd), and if positive definite runs the cavities of all local sites
 * against Qi + df*dQi (cav_pd = 1 iff all local cavities are pos.def.,
 * first_bad = first failing local site or -1).
 */
int epx_damped_trial(epx_ctx *ctx, double df, const double *packed_host,
                     const double *packed_dev, int *global_pd, int *cav_pd, int *first_bad);
/* accept: Qi <- Qi + df*dQi, ri <- ri + df*dri (the swap of method.py:1145-1158) */
int epx_accept(epx_ctx *ctx, double df);
/* moments of the accepted global approximation (method.py:1211-1219):
 * S = Q^-1 (d,d), m = S r */
int epx_global_moments(epx_ctx *ctx, double *S, double *m);
/* force-pd fallback, method.py:1119-1129: min eigenvalue of Qi + df*dQi per
 * local site; where it is < thresh, adds (min_eig_target - min_eig) to the
 * diagonal of Qi.  forced[K_local] out. */
int epx_force_pd(epx_ctx *ctx, double df, double thresh, double min_eig_target, uint8_t *forced);

/* Adaptation history of site k kept for `adapt = carry` (written by every sampling call): final step size of
 * each chain (chains entries, -1: none) and the site's diagonal metric (P entries).  Either may be NULL. */
int epx_get_adapt(epx_ctx *ctx, int k, double *eps, double *metric);

/* TEST HOOK: log density and gradient of site k at theta (P) against the
 * cavity currently held for that site (Appendix A of SURVEY.md). */
int epx_logdensity_grad(epx_ctx *ctx, int k, const double *theta, double *lp, double *grad);
/* The same through a chosen sampler layout (epx_sampler_opts.layout; 0 = what a one-site launch would pick):
 * every layout evaluates the density with its own gradient code. */
int epx_logdensity_grad_layout(epx_ctx *ctx, int k, const double *theta, int layout, double *lp, double *grad);
/* TEST HOOK: sampler only (no moment stage) for sites k0..k0+count */
int epx_sample_batch(epx_ctx *ctx, int k0, int count, const int64_t *seeds,
                     const epx_sampler_opts *opts, double *stats, double *elapsed_ms);
/* TEST HOOK: `nt` plain NUTS transitions (no adaptation) per (site, chain) from
 * given positions q0 (count, chains, P) with given step sizes eps (count, chains)
 * and diagonal inverse metrics inv_e (count, chains, P); the random stream is the
 * one a full run uses at transitions t_offset, t_offset+1, ...  q_out is
 * (count, chains, nt, P), chain_stats (count, chains, EPX_ST_COUNT). */
int epx_nuts_transitions(epx_ctx *ctx, int k0, int count, const int64_t *seeds, int chains, int nt,
                         int t_offset, int layout, const double *q0, const double *eps,
                         const double *inv_e, double *q_out, double *chain_stats);
/* chain stats of the last sampling call: (count, chains, EPX_ST_COUNT) */
/* Sums for Master.mix_phi (method.py:1250-1296, the posterior approximation from the pooled tilted
 * samples): out = [sum_k scatter_k (d*d, column-major), sum_k mean_k (d), sum_k mean_k mean_k' (d*d)] over this
 * context's sites, from the tilted moments of the last epx_tilted_batch / epx_moments_batch. */
int epx_mix_sums(epx_ctx *ctx, double *out);

/* ---------------------------------------------------------------------------------------------
 * Several GPUs: sites are sharded over the ranks (one context each); the only exchange of an EP
 * iteration is the reduction of method.py:1073-1074 (Q = sum_k Qi2 + Q0 over ALL sites) and the logical
 * AND of the cavity flags (:1145).  The library runs both as RCCL all-reduces on the context's stream.
 *   rank 0:      epx_comm_unique_id(id)           -> hand the EPX_COMM_ID_BYTES bytes to every rank
 *   every rank:  epx_comm_init(ctx, id, rank, nranks)      (collective: returns when all ranks called)
 * The reference runs in one process and has no counterpart. */
#define EPX_COMM_ID_BYTES 128
enum epx_op { EPX_OP_SUM = 0, EPX_OP_MIN = 1, EPX_OP_MAX = 2 };
int epx_comm_unique_id(void *id_out);
int epx_comm_init(epx_ctx *ctx, const void *id, int rank, int nranks);
/* The same binding over a transport of the caller's (an MPI / gloo / socket all-reduce on HOST memory) for nodes
 * without RCCL peer access and for tests: `fn(buf, n, op, user)` reduces buf[n] over the ranks in place (op: enum
 * epx_op) and returns 0.  Every library collective then stages through the host: same results, one extra
 * synchronisation per collective. */
typedef int (*epx_host_allreduce_fn)(double *buf, long long n, int op, void *user);
int epx_comm_init_host(epx_ctx *ctx, int rank, int nranks, epx_host_allreduce_fn fn, void *user);
int epx_comm_destroy(epx_ctx *ctx);
/* rank and number of ranks of the context's communicator as RCCL reports them (0 of 1 without one) */
int epx_comm_size(epx_ctx *ctx, int *rank, int *nranks);
/* small host-side collectives over the communicator (staged through device memory, synchronous):
 * buf[n] reduced in place with op; in[n] of every rank gathered into out[n * nranks] in rank order.
 * Without a communicator both are the identity. */
int epx_comm_allreduce(epx_ctx *ctx, double *buf, int n, int op);
int epx_comm_allgather(epx_ctx *ctx, const double *in, int n, double *out);

/*
 * One damping trial of the update phase (method.py:1067-1143) as ONE stream-ordered batch with ONE host
 * synchronisation:  [reduce_sums: packed site sums of this rank -> all-reduce(sum) over the communicator,
 * the caller's statistics riding on the same buffer] -> Q, r for df -> Cholesky check (:1077-1080)
 * [-> S = Q^-1, m = S r with want_moments (:1211-1216)] -> cavities of all local sites against Qi + df*dQi
 * (:1138-1143) -> flags -> all-reduce(min) of the flags.
 *   reduce_sums  1 on the first trial of an iteration (or after epx_force_pd), 0 on the following ones: the
 *                proposal is affine in df, the reduced sums are kept;
 *   stat_sum[n_sum], stat_max[n_max]  in/out, only with reduce_sums, at most 8 each: summed / maximised over
 *                the ranks (the any-site-ok / all-sites-ok counts and the analytics of method.py:1043-1045);
 *   site_base    global index of the context's first site;
 *   global_pd, cav_pd  the two checks over ALL ranks; first_bad: first failing site (global index) or -1;
 *   S (d,d), m (d)  moments of the proposal (valid when both checks hold), may be NULL.
 */
int epx_update_trial(epx_ctx *ctx, double df, int reduce_sums, int site_base, double *stat_sum, int n_sum,
                     double *stat_max, int n_max, int want_moments, int *global_pd, int *cav_pd,
                     int64_t *first_bad, double *S, double *m);

/* Damping sweep (experiment/find_damp.py:146-173, the loop `for di, df in enumerate(damps)`): for every
 * dfs[i] form the proposal Q = Q0 + sum(Qi + df dQi), r likewise, factorise, S = Q^-1, m = S r, all cavities,
 * and score the proposal against a target posterior: out[i*5 + {0..4}] =
 *   global_pd (0/1), all local cavities pd (0/1), mse = mean((m - m_target)^2)      (:157),
 *   KL(N(m_target,S_target) || N(m,S))   (kl_mvn, :38-56; half_logdet_S_target = sum(log(diag(cho(S_target))))),
 *   ll = sum_s log N(x_s | m, S) over the target samples (:159-160), given by their sufficient statistics
 *        samp_mean (d), samp_scatter = sum (x_s - mean)(x_s - mean)' (d x d), n_samp  (NULL / 0: ll = NaN).
 * The criteria are NaN unless both flags are 1 (the reference leaves its pre-filled NaN there).  All trials
 * run back to back on the context's stream with one synchronisation at the end.  packed_*: as in
 * epx_damped_trial.  With several ranks the caller min-reduces column 1 (and voids rows accordingly). */
int epx_damp_sweep(epx_ctx *ctx, int ndf, const double *dfs, const double *packed_host, const double *packed_dev,
                   const double *m_target, const double *S_target, double half_logdet_S_target,
                   const double *samp_mean, const double *samp_scatter, int n_samp, double *out);

/* Scheduling hint for the next sampling calls that cover sites 0..count-1: workgroup i takes site
 * order[i] (a permutation of 0..count-1).  Workgroups are dispatched in index order, so listing the
 * sites by decreasing expected work (e.g. the leapfrogs of the previous EP iteration) shortens the
 * tail of the launch.  Results do not depend on it.  NULL / count 0 clears the hint.
 * No reference counterpart (the reference runs its sites one after the other, method.py:1005-1023). */
int epx_set_site_order(epx_ctx *ctx, const int32_t *order, int count);

/* With a site order set and the layout left to the library (epx_sampler_opts.layout 0) on a batch
 * large enough for layout 1: the first n_lead sites of the order run one workgroup per chain
 * (layout 2) on a second queue, concurrently with layout 1 for the others.  A sampling launch ends
 * with its slowest chain (max_treedepth leapfrogs in every transition); listing the sites expected
 * to hold such chains first lets them run at layout 2's shorter leapfrog.  n_lead is clamped so
 * that the lead sites occupy at most half of the CUs; 0 (default) disables.  Draws of a site are
 * those of the layout it ran in.  epx_last_split: lead sites of the last sampling call.
 * No reference counterpart. */
int epx_set_site_split(epx_ctx *ctx, int n_lead);
int epx_last_split(epx_ctx *ctx);
/* Pieced launch of the samplers that keep one workgroup per site (layouts 5, 7 and 3): with a piece queue set, a
 * sampling call over ALL sites cuts every site's run into PIECES (piece_len transitions) and runs as many workgroups as
 * the device holds at a time; a workgroup claims the site with the largest predicted remaining work (transitions left x
 * rate[site], rate = predicted leapfrogs per transition, NULL = all equal) that nobody holds, runs its next piece,
 * leaves a checkpoint at the transition boundary, puts the site back and looks for the next one until no site has
 * anything left -- longest remaining processing time first, which ends all sites at about the same time whatever they
 * really cost (a launch of one workgroup per site ends with the CU that drew two heavy sites).  Exactly the draws of
 * the plain launch.  (Environment, diagnostics only: EPX_PIECE_GRID=1 launches one workgroup per piece instead of
 * looping ones -- same draws, 1-6 % slower: the device deals a grid's workgroups to its XCDs in order.)
 * piece_len <= 0 clears.  No counterpart in the reference (scheduling only). */
int epx_set_piece_queue(epx_ctx *ctx, int piece_len, const double *rate);
/* TEST HOOK: a per-transition trace of the sampler.  With sites > 0 every sampling call (epx_tilted_batch /
 * epx_sample_batch) also records, for its first `sites` sites, ONE RECORD PER TRANSITION of every chain -- the warm-up
 * included, which the draws returned to the caller (method.py:88-104: post-warm-up only) never show:
 *   [0] step size used  [1] leapfrogs  [2] accept statistic  [3] tree depth  [4] divergent
 *   [5] step size after stepsize_adaptation::learn_stepsize / complete_adaptation (in front of the step-size search that
 *       follows a new metric)  [6] sum of the diagonal metric after var_adaptation::learn_variance  [7] log density
 *   [8 .. 8 + P) the new sample.
 * epx_get_trace copies sites x chains x iter x (8 + P) doubles.  What stands behind it in the reference is PyStan's own
 * sampler state (get_sampler_params(inc_warmup=True), /root/reference/epstan/method.py:99-102 reads stepsize__ from it);
 * tests/test_gpu_round5.py and bench.py's parity record compare it with the same trace of oracle/nuts_oracle.c. */
int epx_set_trace(epx_ctx *ctx, int sites);
/* TEST HOOK: teacher-forced transition.  Every chain of every site takes ONE transition, number t0 (0 < t0 < iter) of a
 * run with the given options, from the state in `records_in` -- per (site, chain) a checkpoint record of the pieced launch
 * (csrc/epx_pieces.h: (4 NV + 1) x 64 doubles, NV = ceil(P / 64): sample, Welford mean, Welford sum of squares, metric in
 * element order with stride NV x 64, then the 20 scalars of EPX_CK_LIST: log density, step size, dual-averaging state,
 * window counters, statistics) -- and leaves the record of boundary t0 + 1 in `records_out`: the adaptation a warm-up
 * transition performs (stepsize_adaptation::learn_stepsize, var_adaptation::learn_variance and the step-size search behind a
 * new metric) becomes comparable with oracle/nuts_oracle.c from ANY state of the oracle's run, warm-up included, without
 * the two runs having to stay together up to there.  Runs the kernels of the piece queue (layouts 5, 7, 3), one workgroup
 * per site.  With epx_set_trace the transition's trace record is available as well. */
int epx_sample_piece(epx_ctx *ctx, const int64_t *seeds, const epx_sampler_opts *opts, int t0,
                     const double *records_in, double *records_out);
int epx_get_trace(epx_ctx *ctx, double *out, long long n_out);
/* Passes over the site rows that the last sampling call's ROW TEAM made for sites k0 .. k0 + count - 1 (layout 7: the four
 * chains of a site share a pass; a pass that one chain sat out while its bookkeeping ran -- a yield -- counts, which the
 * chains' own gradient counts cannot show).  Zero for sites the other layouts sampled.  A measurement aid (bench.py's
 * pass_cycles); nothing in the reference corresponds to it. */
int epx_get_team_passes(epx_ctx *ctx, int k0, int count, double *out);
/* minus the pieces per site of the last sampling call if it ran from the piece queue, 0: one workgroup per site */
int epx_last_segments(epx_ctx *ctx);
/* Compute units of the context's device (the host-side scheduling heuristics size themselves by it). */
int epx_cu_count(epx_ctx *ctx);

/* Thread layout the last sampling call ran with (1 ... 7, see epx_sampler_opts.layout; 0 before
 * the first call).  Measurement aid: layout 3 streams the rows from HBM once per leapfrog, so its
 * roofline is the HBM one (bench.py).  No reference counterpart. */
int epx_last_layout(epx_ctx *ctx);
int epx_get_chain_stats(epx_ctx *ctx, int k0, int count, double *out);
/* TEST HOOK: uniforms/normals of the device random stream */
int epx_rng_probe(int device, uint64_t seed, int chain, uint32_t t, uint32_t kind, uint32_t a,
                  uint32_t b, double *out4);

/*
 * Stand-alone batched forms of epstan/util.py on device (no context needed):
 * util.invert_normal_params (util.py:51-125): (A,b) -> (A^-1, A^-1 b), A given
 * as SPD matrices or, with cho_form, as UPPER Cholesky factors; nb matrices of
 * order d, F-order, in place; b may be NULL; info[nb]: 0 ok, 1 not pos.def.
 */
int epx_invert_normal_params(int device, int d, int nb, double *A, double *b, int cho_form,
                             int32_t *info);
/* util.olse (util.py:128-194): shrinkage precision estimate of sample
 * covariances S (nb of order d, in place) from n draws with prior matrix P
 * (nb matrices, or NULL for the naive I/d prior). */
int epx_olse(int device, int d, int nb, double *S, int n, const double *P, int32_t *info);

#ifdef __cplusplus
}
#endif
#endif /* EPX_H */
