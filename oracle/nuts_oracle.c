/*
 * nuts_oracle.c -- CPU restatement of the per-site sampler -- TEST INFRASTRUCTURE.
 *
 * PARITY UNPINNED (see oracle/ep_oracle.py header and DESIGN.md): the
 * reference samples each site's tilted distribution with PyStan 2.17.0.0 /
 * Stan 2.17 C++ NUTS (/root/reference/README.md:10, epstan/util.py:34,716,
 * epstan/method.py:43-118), a third-party dependency that is not in
 * /root/reference and cannot be installed here.  This file restates
 *   (1) the site log-densities of the reference's own Stan programs
 *       /root/reference/experiment/models/m{1,2,3,4,5}b_sg.stan (model block,
 *       transformed parameters) and their analytic gradients,
 *       and of the Gaussian-likelihood family m{1..5}a_sg.stan (model ids 5..9:
 *       phi = [log sigma | the b-model's phi], y ~ normal(alpha + X beta, sigma);
 *       for these ids the `y` argument of every entry point points to DOUBLES),
 *   (2) Stan 2.17's published sampler: multinomial NUTS with the generalised
 *       U-turn criterion (stan/mcmc/hmc/nuts/base_nuts.hpp @ v2.17), diagonal
 *       Euclidean metric, step-size heuristic (base_hmc::init_stepsize),
 *       dual averaging (stepsize_adaptation.hpp: delta .8, gamma .05, t0 10,
 *       kappa .75), windowed variance adaptation (windowed_adaptation.hpp,
 *       var_adaptation.hpp: 75/25/50 buffers, 15%/75%/10% when warm-up < 150).
 * It is anchored on the reference's call sites (chains/iter/warmup/thin/init,
 * method.py:154-160; draws concatenated chain-major, util.py:475-484) and
 * checked by finite differences and Gaussian known answers in tests/.
 *
 * The random stream is counter based (Philox4x32-10) and the tree is built
 * iteratively so that the HIP kernel (ep-stan_amd/csrc/nuts.hip) can follow the
 * same sequence of decisions; the two are compared draw by draw in tests.
 *
 * One deliberate, distribution-preserving restatement: inside a new subtree
 * base_nuts.hpp picks the proposal by progressive multinomial sampling at every
 * merge (exp/log per merge).  Here every leaf i draws a Gumbel key
 * dH_i - log(-log u_i) and a subtree proposes its arg-max leaf, which IS a
 * multinomial draw with weights exp(dH_i) (Gumbel-max), so merges only compare
 * keys; the subtree's total weight log sum_i exp(dH_i) is accumulated on the
 * side for the top-level (biased progressive) acceptance, which is unchanged.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPO_MAX_DEPTH_CAP 16

enum { M1B = 0, M2B = 1, M3B = 2, M4B = 3, M5B = 4, M1A = 5, M2A = 6, M3A = 7, M4A = 8, M5A = 9 };
static inline int is_gauss(int model) { return model >= M1A && model <= M5A; }
static inline int base_model(int model) { return is_gauss(model) ? model - M1A : model; }
enum { K_INIT = 0, K_MOM = 1, K_DIR = 2, K_TOP = 3, K_MERGE = 4, K_SSMOM = 5 };

/* per-chain statistics written to `stats` (doubles) */
enum {
    ST_STEPSIZE_MEAN = 0, /* mean of stepsize__ over ALL iterations (method.py:99-102) */
    ST_STEPSIZE_FINAL = 1,
    ST_NLEAP = 2,         /* total leapfrogs == gradient evaluations in trees */
    ST_NGRAD = 3,         /* all gradient evaluations incl. step-size searches */
    ST_NDIV = 4,          /* post-warm-up divergent transitions */
    ST_ACCEPT_MEAN = 5,   /* post-warm-up mean accept_stat */
    ST_DEPTH_MEAN = 6,    /* post-warm-up mean tree depth */
    ST_FAIL = 7,          /* 1 if the initial point had a non-finite density */
    ST_COUNT = 8
};

/* ------------------------------------------------------------------ dims */
int epo_dphi(int model, int D) {
    const int o = is_gauss(model);          /* m*a_sg.stan: one more shared parameter, log sigma, in front */
    switch (base_model(model)) {
    case M1B: return D + 1 + o;
    case M2B: return 2 + o;
    case M3B: return D + 1 + o;
    case M4B: case M5B: return 2 * D + 2 + o;
    }
    return -1;
}
int epo_npar(int model, int D) {
    const int o = is_gauss(model);
    switch (base_model(model)) {
    case M1B: return D + 2 + o;
    case M2B: return D + 3 + o;
    case M3B: return 2 * D + 2 + o;
    case M4B: case M5B: return 3 * D + 3 + o;
    }
    return -1;
}

/* --------------------------------------------------------------- Philox */
static inline void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                              uint32_t c2, uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline double u01(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}
typedef struct { uint32_t k0, k1; } rng_key;
static inline rng_key make_key(uint64_t seed, int chain) {
    rng_key k;
    k.k0 = (uint32_t)seed;
    k.k1 = (uint32_t)(seed >> 32) ^ (0x85EBCA6Bu * (uint32_t)(chain + 1));
    return k;
}
static inline void rng_u2(rng_key k, uint32_t t, uint32_t kind, uint32_t a, uint32_t b,
                          double *u1, double *u2) {
    uint32_t o[4];
    philox4x32(k.k0, k.k1, t, kind, a, b, o);
    *u1 = u01(o[0], o[1]);
    *u2 = u01(o[2], o[3]);
}
static inline double rng_uniform(rng_key k, uint32_t t, uint32_t kind, uint32_t a, uint32_t b) {
    double u1, u2;
    rng_u2(k, t, kind, a, b, &u1, &u2);
    return u1;
}
/* standard normal for vector element e (Box-Muller pair shared by e, e^1) */
static inline double rng_normal(rng_key k, uint32_t t, uint32_t kind, int e, uint32_t b) {
    double u1, u2;
    rng_u2(k, t, kind, (uint32_t)(e >> 1), b, &u1, &u2);
    double r = sqrt(-2.0 * log(u1));
    double a = 6.283185307179586476925286766559 * u2;
    return (e & 1) ? r * sin(a) : r * cos(a);
}

/* ----------------------------------------------------------- log density */
typedef struct {
    int model, n, D, d, P;
    const double *X;   /* n x D row-major (C-order view, method.py:829) */
    const int32_t *y;  /* n, 0/1 */
    const double *yd;  /* n, real responses of the Gaussian family (m*a_sg.stan `real y[N]`), else NULL */
    const double *mu;  /* d cavity mean   (Worker.vec, method.py:221) */
    const double *Om;  /* d x d cavity precision, symmetric (Worker.Mat, :222) */
    /* groups of the site (K < J, experiment/models/m*b.stan `j_ind`): ng contiguous row blocks,
     * gl[j]..gl[j+1] relative to the site's first row; NULL = one group (the `_sg` programs) */
    int ng;
    const int64_t *gl;
    double *beta;      /* ng x D scratch */
    double *db;        /* ng x D scratch */
    double *da;        /* ng scratch */
    double *Ov;        /* d scratch */
} site_t;

/* sampled coordinates of a site with ng groups: phi (d), eta (ng), etb (ng x D, group-major) */
int epo_npar_groups(int model, int D, int ng) {
    const int d = epo_dphi(model, D);
    if (d < 0 || ng < 1) return -1;
    return d + ng * (base_model(model) == M1B ? 1 : 1 + D);
}
static size_t site_scratch(int D, int d, int ng) { return (size_t)2 * ng * D + ng + d; }
static void site_bind_scratch(site_t *s, double *w) {
    s->beta = w; s->db = w + (size_t)s->ng * s->D; s->da = s->db + (size_t)s->ng * s->D; s->Ov = s->da + s->ng;
}

#ifdef EPO_FAST
/* TIMING BUILD ONLY (oracle/Makefile, FAST_LIB; bench.py's cpu_baseline): the logistic terms of a block of rows with
 * branch-free polynomial exp / log1p that the compiler vectorises (libm's exp and log1p are scalar calls: 60 % of a
 * gradient of the strict build).  Same mathematics, ~1e-16 relative; never used as the checker -- the strict build
 * below keeps libm.  tests/test_nuts_oracle.py compares the two builds. */
static inline double epo_exp_nonpos(double x) {          /* exp(x), x <= 0 */
    const double xc = x < -700.0 ? -700.0 : x;
    const double kf = rint(xc * 1.4426950408889634074);
    double r = fma(kf, -6.93147180369123816490e-01, xc);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09); p = fma(p, r, 2.505210838544172e-08); p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06); p = fma(p, r, 2.48015873015873e-05); p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03); p = fma(p, r, 8.333333333333333e-03); p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01); p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    /* 2^k through the exponent field (k >= -1010): kf + 1.5 * 2^52 has k in its low mantissa bits */
    union { double d; uint64_t u; } t, sc;
    t.d = kf + 6755399441055744.0;
    sc.u = (t.u - 0x4338000000000000ull + 1023ull) << 52;
    return p * sc.d;
}
static inline double epo_log1p_unit(double e) {         /* log(1 + e), 0 <= e <= 1: 2 atanh((m-1)/(m+1)), m folded into [1/sqrt2, sqrt2] */
    const double m = 1.0 + e;
    const int big = m > 1.4142135623730951;
    const double a = big ? fma(0.5, m, -1.0) : e, b = big ? fma(0.5, m, 1.0) : 2.0 + e;
    const double sq = a / b, z = sq * sq;
    double p = 4.7619047619047616e-02;
    p = fma(p, z, 5.2631578947368418e-02); p = fma(p, z, 5.8823529411764705e-02); p = fma(p, z, 6.6666666666666666e-02);
    p = fma(p, z, 7.6923076923076927e-02); p = fma(p, z, 9.0909090909090912e-02); p = fma(p, z, 1.1111111111111110e-01);
    p = fma(p, z, 1.4285714285714285e-01); p = fma(p, z, 0.2); p = fma(p, z, 3.3333333333333331e-01); p = fma(p, z, 1.0);
    const double rr = 2.0 * sq * p;
    return big ? rr + 6.931471805599453094e-01 : rr;
}
#define EPO_BLK 16
#endif
/* log(1+exp(-|f|)) and sigmoid(f) sharing one exp */
static inline void logistic_terms(double f, double y, double *ll, double *g) {
    double e = exp(-fabs(f));
    double l1p = log1p(e);
    double s = (f >= 0) ? 1.0 / (1.0 + e) : e / (1.0 + e);   /* sigmoid(f) */
    *ll = y * f - (fmax(f, 0.0) + l1p);   /* y f - log(1+e^f), bernoulli_logit */
    *g = y - s;
}

/* m*b_sg.stan / m*b.stan model blocks; SURVEY.md Appendix A.  With ng groups every group j has
 * its own eta_j (and etb_j), alpha_j, beta_j (m4b.stan:33-41); the hyper-parameters phi are shared. */
static double site_lp_grad(const site_t *s, const double *th, double *grad) {
    const int D = s->D, d = s->d, model = base_model(s->model), ng = s->ng;
    const int gauss = is_gauss(s->model);
    const double *phi = th + gauss;                     /* the b-model's phi; th[0] = log sigma for the a-models */
    const double *eta = th + d;
    const double *etb = th + d + ng;                    /* [group][D] */
    const int laplace = (model == M5B);
    const double inv_s2 = gauss ? exp(-2.0 * th[0]) : 0.0;
    double rss = 0.0;                                   /* sum of squared residuals / sigma^2 */
    int64_t nrow = 0;
    const double sa = exp(model >= M4B ? phi[1] : phi[0]);
    const double a0 = model >= M4B ? phi[0] : 0.0;
    double ll = 0.0;
    for (int j = 0; j < ng; ++j) {
        double *beta = s->beta + (size_t)j * D, *db = s->db + (size_t)j * D;
        const double *eb = etb + (size_t)j * D;
        const double alpha = a0 + eta[j] * sa;
        switch (model) {
        case M1B: for (int c = 0; c < D; ++c) beta[c] = phi[1 + c]; break;
        case M2B: { const double sb = exp(phi[1]); for (int c = 0; c < D; ++c) beta[c] = eb[c] * sb; break; }
        case M3B: for (int c = 0; c < D; ++c) beta[c] = eb[c] * exp(phi[1 + c]); break;
        default:  for (int c = 0; c < D; ++c) beta[c] = phi[2 + c] + eb[c] * exp(phi[2 + D + c]); break;
        }
        /* y ~ bernoulli_logit(alpha_j + X*beta_j): one fused pass over the rows of the group */
        double da = 0.0;
        for (int c = 0; c < D; ++c) db[c] = 0.0;
        const int64_t lo = s->gl ? s->gl[j] : 0, hi = s->gl ? s->gl[j + 1] : s->n;
#ifdef EPO_FAST
        int64_t i0 = lo;
        if (!gauss) {
            /* blocks of EPO_BLK rows: dot products, the logistic terms of the block in one vectorised loop, rank-1 updates */
            for (; i0 + EPO_BLK <= hi; i0 += EPO_BLK) {
                double fb[EPO_BLK], gb[EPO_BLK], lb[EPO_BLK];
                for (int b = 0; b < EPO_BLK; ++b) {
                    const double *x = s->X + (size_t)(i0 + b) * D;
                    double f = alpha;
#pragma omp simd reduction(+ : f)
                    for (int c = 0; c < D; ++c) f += x[c] * beta[c];
                    fb[b] = f;
                }
#pragma omp simd
                for (int b = 0; b < EPO_BLK; ++b) {
                    const double f = fb[b], yv = (double)s->y[i0 + b];
                    const double e = epo_exp_nonpos(-fabs(f));
                    const double inv = 1.0 / (1.0 + e);
                    const double sg = (f >= 0) ? inv : e * inv;
                    lb[b] = yv * f - (fmax(f, 0.0) + epo_log1p_unit(e));
                    gb[b] = yv - sg;
                }
                for (int b = 0; b < EPO_BLK; ++b) {
                    const double *x = s->X + (size_t)(i0 + b) * D;
                    const double g = gb[b];
                    ll += lb[b]; da += g;
#pragma omp simd
                    for (int c = 0; c < D; ++c) db[c] += g * x[c];
                }
            }
        }
        for (int64_t i = i0; i < hi; ++i) {
#else
        for (int64_t i = lo; i < hi; ++i) {
#endif
            const double *x = s->X + (size_t)i * D;
            double f = alpha;
            for (int c = 0; c < D; ++c) f += x[c] * beta[c];
            double l, g;
            if (gauss) {            /* y ~ normal(f, sigma): -log sigma - (y-f)^2 / (2 sigma^2) */
                const double r = s->yd[i] - f;
                g = r * inv_s2; l = -0.5 * r * g; rss += r * g; ++nrow;
            } else logistic_terms(f, (double)s->y[i], &l, &g);
            ll += l; da += g;
            for (int c = 0; c < D; ++c) db[c] += g * x[c];
        }
        s->da[j] = da;
    }
    /* phi ~ multi_normal_prec(mu_phi, Omega_phi) */
    double quad = 0.0;
    for (int i = 0; i < d; ++i) {
        double acc = 0.0;
        const double *row = s->Om + (size_t)i * d;
#ifdef EPO_FAST
#pragma omp simd reduction(+ : acc)
#endif
        for (int j = 0; j < d; ++j) acc += row[j] * (th[j] - s->mu[j]);
        s->Ov[i] = acc;
        quad += (th[i] - s->mu[i]) * acc;
    }
    double lp = -0.5 * quad + ll;
    for (int i = 0; i < d; ++i) grad[i] = -s->Ov[i];
    if (gauss) { lp -= (double)nrow * th[0]; grad[0] += rss - (double)nrow; }
    grad += gauss;                                      /* the b-model's indices below; d, eta, etb shift too */
    const int db_ = d - gauss;
#define SGN(v) (((v) > 0) - ((v) < 0))
    /* eta, etb ~ normal(0,1) (double_exponential(0,1) for m5b*.stan:40-41), and the chain rule */
    for (int j = 0; j < ng; ++j) {
        const double *db = s->db + (size_t)j * D, *eb = etb + (size_t)j * D;
        const double da = s->da[j], et = eta[j];
        double *geb = grad + db_ + ng + (size_t)j * D;
        lp -= laplace ? fabs(et) : 0.5 * et * et;
        grad[db_ + j] = da * sa - (laplace ? (double)SGN(et) : et);
        if (model != M1B)
            for (int c = 0; c < D; ++c) lp -= laplace ? fabs(eb[c]) : 0.5 * eb[c] * eb[c];
        switch (model) {
        case M1B:
            grad[0] += da * et * sa;
            for (int c = 0; c < D; ++c) grad[1 + c] += db[c];
            break;
        case M2B: {
            const double sb = exp(phi[1]);
            double dot = 0.0;
            for (int c = 0; c < D; ++c) dot += db[c] * eb[c];
            grad[0] += da * et * sa;
            grad[1] += dot * sb;
            for (int c = 0; c < D; ++c) geb[c] = db[c] * sb - eb[c];
            break; }
        case M3B:
            grad[0] += da * et * sa;
            for (int c = 0; c < D; ++c) {
                const double sb = exp(phi[1 + c]);
                grad[1 + c] += db[c] * eb[c] * sb;
                geb[c] = db[c] * sb - eb[c];
            }
            break;
        default:
            grad[0] += da;
            grad[1] += da * et * sa;
            for (int c = 0; c < D; ++c) {
                const double sb = exp(phi[2 + D + c]);
                grad[2 + c] += db[c];
                grad[2 + D + c] += db[c] * eb[c] * sb;
                geb[c] = db[c] * sb - (laplace ? (double)SGN(eb[c]) : eb[c]);
            }
            break;
        }
    }
    return lp;
}

/* gl: ng+1 row limits relative to the site's first row, or NULL with ng = 1 */
int epo_logdensity_grad_groups(int model, int n, int D, int ng, const int64_t *gl, const double *X,
                               const int32_t *y, const double *mu, const double *Omega,
                               const double *theta, double *lp, double *grad) {
    site_t s;
    s.model = model; s.n = n; s.D = D; s.d = epo_dphi(model, D); s.ng = ng; s.gl = gl;
    s.P = epo_npar_groups(model, D, ng);
    if (s.d < 0 || s.P < 0) return -1;
    s.X = X; s.y = y; s.mu = mu; s.Om = Omega;
    s.yd = NULL;
    if (is_gauss(model)) { s.yd = (const double *)(const void *)y; s.y = NULL; }
    double *w = (double *)malloc(sizeof(double) * site_scratch(D, s.d, ng));
    site_bind_scratch(&s, w);
    *lp = site_lp_grad(&s, theta, grad);
    free(w);
    return 0;
}
int epo_logdensity_grad(int model, int n, int D, const double *X, const int32_t *y,
                        const double *mu, const double *Omega, const double *theta,
                        double *lp, double *grad) {
    return epo_logdensity_grad_groups(model, n, D, 1, NULL, X, y, mu, Omega, theta, lp, grad);
}

/* ------------------------------------------------------------ NUTS chain */
static inline double log_sum_exp2(double a, double b) {
    if (a == -INFINITY) return b;
    if (a == INFINITY && b == INFINITY) return INFINITY;
    if (a > b) return a + log1p(exp(b - a));
    return b + log1p(exp(a - b));
}

typedef struct {
    int P;
    double *q, *p, *g;   /* position, momentum, gradient of lp */
    double lp;
} zstate;

typedef struct {
    const site_t *site;
    rng_key key;
    int P, max_depth;
    double *inv_e;        /* diagonal inverse metric */
    double eps;
    /* current sample */
    double *qs, *gs; double lps;
    /* work vectors */
    double *zq, *zp, *zg; double zlp;           /* integrator state */
    double *pq, *pp, *pg; double plp;           /* tree +end */
    double *mq, *mp, *mg; double mlp;           /* tree -end */
    double *rho, *psp, *psm;                    /* whole tree */
    double *n_rho, *n_psl, *n_pq, *n_pg; double n_key, n_plp;   /* current node */
    double *psr;                                /* p_sharp of the newest leaf */
    double *st_rho, *st_psl, *st_pq, *st_pg;    /* stack: [level][P] */
    double st_key[EPO_MAX_DEPTH_CAP], st_plp[EPO_MAX_DEPTH_CAP];
    double *tq, *tg; double tlp;                /* proposal of the new subtree */
    long ngrad;
} chain_t;

static inline double kinetic(const chain_t *c, const double *p) {
    double t = 0.0;
#ifdef EPO_FAST
#pragma omp simd reduction(+ : t)
#endif
    for (int i = 0; i < c->P; ++i) t += c->inv_e[i] * p[i] * p[i];
    return 0.5 * t;
}

/* expl_leapfrog: half momentum, full position, gradient, half momentum */
static void leapfrog(chain_t *c, double eps) {
    const int P = c->P;
    for (int i = 0; i < P; ++i) c->zp[i] += 0.5 * eps * c->zg[i];
    for (int i = 0; i < P; ++i) c->zq[i] += eps * c->inv_e[i] * c->zp[i];
    c->zlp = site_lp_grad(c->site, c->zq, c->zg);
    c->ngrad++;
    for (int i = 0; i < P; ++i) c->zp[i] += 0.5 * eps * c->zg[i];
}

static inline int criterion(const chain_t *c, const double *psm, const double *psp,
                            const double *rho) {
    double a = 0.0, b = 0.0;
#ifdef EPO_FAST
#pragma omp simd reduction(+ : a, b)
#endif
    for (int i = 0; i < c->P; ++i) { a += psp[i] * rho[i]; b += psm[i] * rho[i]; }
    return a > 0 && b > 0;
}

typedef struct { double accept; int nleap; int depth; int divergent; } trans_info;

/* base_nuts::transition @ Stan 2.17, with build_tree unrolled into a loop over
 * the 2^depth leaves of the new subtree and a per-level stack of pending left
 * siblings. */
static trans_info transition(chain_t *c, uint32_t t) {
    const int P = c->P;
    const size_t vb = sizeof(double) * P;
    trans_info ti = {0.0, 0, 0, 0};
    /* sample momentum p ~ N(0, M), M = diag(1/inv_e) */
    for (int i = 0; i < P; ++i)
        c->pp[i] = rng_normal(c->key, t, K_MOM, i, 0) / sqrt(c->inv_e[i]);
    memcpy(c->pq, c->qs, vb); memcpy(c->pg, c->gs, vb); c->plp = c->lps;
    memcpy(c->mq, c->qs, vb); memcpy(c->mg, c->gs, vb); c->mlp = c->lps;
    memcpy(c->mp, c->pp, vb);
    for (int i = 0; i < P; ++i) { c->psp[i] = c->inv_e[i] * c->pp[i]; c->psm[i] = c->psp[i]; }
    memcpy(c->rho, c->pp, vb);
    double H0 = -c->lps + kinetic(c, c->pp);
    double lsw = 0.0, sum_metro = 0.0;
    int depth = 0, nleap = 0, divergent = 0;

    while (depth < c->max_depth) {
        int fwd = rng_uniform(c->key, t, K_DIR, (uint32_t)depth, 0) > 0.5;
        double eps = fwd ? c->eps : -c->eps;
        if (fwd) { memcpy(c->zq, c->pq, vb); memcpy(c->zp, c->pp, vb); memcpy(c->zg, c->pg, vb); c->zlp = c->plp; }
        else     { memcpy(c->zq, c->mq, vb); memcpy(c->zp, c->mp, vb); memcpy(c->zg, c->mg, vb); c->zlp = c->mlp; }
        int valid = 1;
        const int nleaf = 1 << depth;
        double lw_sub = -INFINITY;            /* log sum of the leaf weights of the new subtree */
        for (int i = 0; i < nleaf; ++i) {
            leapfrog(c, eps);
            ++nleap;
            double h = -c->zlp + kinetic(c, c->zp);
            if (isnan(h)) h = INFINITY;
            if (h - H0 > 1000.0) divergent = 1;
            double dH = H0 - h;
            sum_metro += (dH > 0) ? 1.0 : exp(dH);
            if (divergent) { valid = 0; break; }
            lw_sub = log_sum_exp2(lw_sub, dH);
            /* Gumbel key of this leaf: arg-max over a subtree == multinomial draw */
            double u1, u2;
            rng_u2(c->key, t, K_MERGE, (uint32_t)depth, (uint32_t)(i >> 1), &u1, &u2);
            double gum = -log(-log((i & 1) ? u2 : u1));
            /* depth-0 node */
            memcpy(c->n_rho, c->zp, vb);
            for (int j = 0; j < P; ++j) c->psr[j] = c->inv_e[j] * c->zp[j];
            memcpy(c->n_psl, c->psr, vb);
            memcpy(c->n_pq, c->zq, vb); memcpy(c->n_pg, c->zg, vb); c->n_plp = c->zlp;
            c->n_key = dH + gum;
            /* merge with pending left siblings while this leaf closes them */
            int l = 0, ii = i;
            while (ii & 1) {
                const double *L_rho = c->st_rho + (size_t)l * P;
                const double *L_psl = c->st_psl + (size_t)l * P;
                int take_right = c->n_key > c->st_key[l];
                if (!take_right) {
                    memcpy(c->n_pq, c->st_pq + (size_t)l * P, vb);
                    memcpy(c->n_pg, c->st_pg + (size_t)l * P, vb);
                    c->n_plp = c->st_plp[l];
                    c->n_key = c->st_key[l];
                }
                for (int j = 0; j < P; ++j) c->n_rho[j] += L_rho[j];
                memcpy(c->n_psl, L_psl, vb);
                if (!criterion(c, c->n_psl, c->psr, c->n_rho)) { valid = 0; break; }
                ii >>= 1; ++l;
            }
            if (!valid) break;
            if (i != nleaf - 1) {
                memcpy(c->st_rho + (size_t)l * P, c->n_rho, vb);
                memcpy(c->st_psl + (size_t)l * P, c->n_psl, vb);
                memcpy(c->st_pq + (size_t)l * P, c->n_pq, vb);
                memcpy(c->st_pg + (size_t)l * P, c->n_pg, vb);
                c->st_key[l] = c->n_key; c->st_plp[l] = c->n_plp;
            }
        }
        if (fwd) { memcpy(c->pq, c->zq, vb); memcpy(c->pp, c->zp, vb); memcpy(c->pg, c->zg, vb); c->plp = c->zlp; }
        else     { memcpy(c->mq, c->zq, vb); memcpy(c->mp, c->zp, vb); memcpy(c->mg, c->zg, vb); c->mlp = c->zlp; }
        if (!valid) break;
        ++depth;
        /* biased progressive sampling of the new subtree's proposal */
        int take;
        if (lw_sub > lsw) take = 1;
        else take = rng_uniform(c->key, t, K_TOP, (uint32_t)(depth - 1), 0) < exp(lw_sub - lsw);
        if (take) { memcpy(c->qs, c->n_pq, vb); memcpy(c->gs, c->n_pg, vb); c->lps = c->n_plp; }
        lsw = log_sum_exp2(lsw, lw_sub);
        for (int j = 0; j < P; ++j) c->rho[j] += c->n_rho[j];
        if (fwd) memcpy(c->psp, c->psr, vb); else memcpy(c->psm, c->psr, vb);
        if (!criterion(c, c->psm, c->psp, c->rho)) break;
    }
    ti.accept = sum_metro / (double)nleap;
    ti.nleap = nleap; ti.depth = depth; ti.divergent = divergent;
    return ti;
}

/* base_hmc::init_stepsize @ Stan 2.17 */
static void init_stepsize(chain_t *c, uint32_t t) {
    const int P = c->P;
    const size_t vb = sizeof(double) * P;
    if (c->eps == 0 || c->eps > 1e7 || isnan(c->eps)) return;
    const double log08 = log(0.8);
    int direction = 0;
    for (uint32_t trial = 0;; ++trial) {
        for (int i = 0; i < P; ++i)
            c->zp[i] = rng_normal(c->key, t, K_SSMOM, i, trial) / sqrt(c->inv_e[i]);
        memcpy(c->zq, c->qs, vb); memcpy(c->zg, c->gs, vb); c->zlp = c->lps;
        double H0 = -c->lps + kinetic(c, c->zp);
        leapfrog(c, c->eps);
        double h = -c->zlp + kinetic(c, c->zp);
        if (isnan(h)) h = INFINITY;
        double dH = H0 - h;
        if (trial == 0) { direction = dH > log08 ? 1 : -1; continue; }
        if (direction == 1 && !(dH > log08)) break;
        else if (direction == -1 && !(dH < log08)) break;
        else c->eps = direction == 1 ? 2 * c->eps : 0.5 * c->eps;
        if (c->eps > 1e7 || c->eps == 0) break;
        if (trial > 200) break;
    }
}

typedef struct {
    int num_warmup, init_buffer, term_buffer, base_window;
    int counter, window_size, next_window;
    double nw; double *mean, *m2;
} var_adapt;

static void va_restart(var_adapt *v) {
    v->counter = 0; v->window_size = v->base_window;
    v->next_window = v->init_buffer + v->window_size - 1;
}
static void va_init(var_adapt *v, int num_warmup, int P, double *buf) {
    v->num_warmup = num_warmup; v->init_buffer = 75; v->term_buffer = 50; v->base_window = 25;
    if (v->init_buffer + v->base_window + v->term_buffer > num_warmup && num_warmup >= 20) {
        v->init_buffer = (int)(0.15 * num_warmup);
        v->term_buffer = (int)(0.1 * num_warmup);
        v->base_window = num_warmup - (v->init_buffer + v->term_buffer);
    }
    v->mean = buf; v->m2 = buf + P; v->nw = 0;
    for (int i = 0; i < 2 * P; ++i) buf[i] = 0.0;
    va_restart(v);
}
/* var_adaptation::learn_variance; returns 1 when the metric was updated */
static int va_learn(var_adapt *v, double *var, const double *q, int P) {
    int in_win = (v->counter >= v->init_buffer) &&
                 (v->counter < v->num_warmup - v->term_buffer) && (v->counter != v->num_warmup);
    if (in_win) {
        v->nw += 1.0;
        for (int i = 0; i < P; ++i) {
            double delta = q[i] - v->mean[i];
            v->mean[i] += delta / v->nw;
            v->m2[i] += (q[i] - v->mean[i]) * delta;
        }
    }
    int end_win = (v->counter == v->next_window) && (v->counter != v->num_warmup);
    if (end_win) {
        /* compute_next_window */
        if (v->next_window != v->num_warmup - v->term_buffer - 1) {
            v->window_size *= 2;
            v->next_window = v->counter + v->window_size;
            if (v->next_window != v->num_warmup - v->term_buffer - 1) {
                int boundary = v->next_window + 2 * v->window_size;
                if (boundary >= v->num_warmup - v->term_buffer)
                    v->next_window = v->num_warmup - v->term_buffer - 1;
            }
        }
        double n = v->nw;
        for (int i = 0; i < P; ++i) {
            double s2 = (n > 1.0) ? v->m2[i] / (n - 1.0) : 0.0;
            var[i] = (n / (n + 5.0)) * s2 + 1e-3 * (5.0 / (n + 5.0));
        }
        v->nw = 0;
        for (int i = 0; i < P; ++i) { v->mean[i] = 0.0; v->m2[i] = 0.0; }
        ++v->counter;
        return 1;
    }
    ++v->counter;
    return 0;
}

/* Per-transition trace (test hook, mirrors epx_set_trace of include/epx.h): when set, the sampling entry points record,
 * for the first g_trace_sites sites, one record of 8 + Pm doubles per (site, chain, transition) -- warm-up included:
 * [eps used, leapfrogs, accept, depth, divergent, eps after learn_stepsize / complete_adaptation, sum of the metric after
 * learn_variance, log density, sample].  Not thread-safe against concurrent callers (the chains of ONE call write
 * disjoint records). */
static double *g_trace = NULL;
static int g_trace_sites = 0;
void epo_set_trace(double *buf, int sites) { g_trace = buf; g_trace_sites = sites; }
/* State dump (test hook, the counterpart of the device library's checkpoint records -- csrc/epx_pieces.h EPX_CK_LIST): at the
 * transition boundaries ts[0..n) (BEFORE transition ts[i]) every chain of the first g_dump_sites sites writes
 *   [20 scalars in EPX_CK_LIST order: lps, eps, da_mu, s_bar, x_bar, da_count, va_n, eps_sum, acc_sum, depth_sum, nleap_tot,
 *    ngrad, t, va_counter, va_wsize, va_next, ndiv, npost, kept, failed] [qs (Pm)] [Welford mean (Pm)] [Welford m2 (Pm)] [metric (Pm)]
 * to buf[((site * chains + chain) * n + i) * (20 + 4 Pm)]: what a device piece needs to continue the chain from there. */
static const int *g_dump_ts = NULL;
static int g_dump_n = 0, g_dump_sites = 0;
static double *g_dump = NULL;
void epo_set_dump(const int *ts, int n, double *buf, int sites) { g_dump_ts = ts; g_dump_n = n; g_dump = buf; g_dump_sites = sites; }

/* One chain of one site update: warm-up + sampling. draws: nkeep x P row-major. */
static void run_chain(const site_t *site_in, uint64_t seed, int chain, int iter, int warmup,
                      int thin, int max_depth, const double *init, double *draws,
                      double *last, double *stats, double eps_in, const double *inv_e_in,
                      int t_offset, double carry_eps, const double *carry_inv_e, double *trace, int trace_stride,
                      double *dump, int dump_stride) {
    const int P = site_in->P, D = site_in->D, d = site_in->d;
    site_t site = *site_in;
    const size_t nvec = 24 + 4 * (size_t)EPO_MAX_DEPTH_CAP;
    double *buf = (double *)calloc(nvec * P + site_scratch(D, d, site.ng), sizeof(double));
    double *w = buf;
#define TAKE(ptr) ptr = w; w += P
    chain_t c;
    memset(&c, 0, sizeof(c));
    c.site = &site; c.P = P; c.max_depth = max_depth; c.key = make_key(seed, chain);
    TAKE(c.inv_e); TAKE(c.qs); TAKE(c.gs); TAKE(c.zq); TAKE(c.zp); TAKE(c.zg);
    TAKE(c.pq); TAKE(c.pp); TAKE(c.pg); TAKE(c.mq); TAKE(c.mp); TAKE(c.mg);
    TAKE(c.rho); TAKE(c.psp); TAKE(c.psm); TAKE(c.n_rho); TAKE(c.n_psl); TAKE(c.n_pq);
    TAKE(c.n_pg); TAKE(c.psr); TAKE(c.tq); TAKE(c.tg);
    double *vabuf = w; w += 2 * P;
    c.st_rho = w; w += (size_t)EPO_MAX_DEPTH_CAP * P;
    c.st_psl = w; w += (size_t)EPO_MAX_DEPTH_CAP * P;
    c.st_pq = w; w += (size_t)EPO_MAX_DEPTH_CAP * P;
    c.st_pg = w; w += (size_t)EPO_MAX_DEPTH_CAP * P;
    site_bind_scratch(&site, w);
    for (int i = 0; i < ST_COUNT; ++i) stats[i] = 0.0;

    for (int i = 0; i < P; ++i) c.inv_e[i] = 1.0;
    /* init='random': U(-2,2) on the unconstrained scale, drawn again -- up to 100 times -- until log density and
     * gradient are finite (stan::services::util::initialize, Stan 2.17, behind PyStan's sampling():
     * /root/reference/epstan/util.py:716); a given start (zeros, the previous draws) gets one attempt */
    int finite = 0;
    for (int attempt = 0; attempt < (init ? 1 : 100) && !finite; ++attempt) {
        if (init) memcpy(c.qs, init, sizeof(double) * P);
        else {
            for (int i = 0; i < P; ++i) {
                double u1, u2;
                rng_u2(c.key, 0, K_INIT, (uint32_t)(i >> 1), (uint32_t)attempt, &u1, &u2);
                c.qs[i] = -2.0 + 4.0 * ((i & 1) ? u2 : u1);
            }
        }
        c.lps = site_lp_grad(&site, c.qs, c.gs);
        c.ngrad++;
        finite = isfinite(c.lps);
        for (int i = 0; i < P; ++i) finite = finite && isfinite(c.gs[i]);
    }
    if (!finite) {
        stats[ST_FAIL] = 1.0;
        int nkeep = (iter - warmup + thin - 1) / thin;
        for (int k = 0; k < nkeep; ++k) memcpy(draws + (size_t)k * P, c.qs, sizeof(double) * P);
        memcpy(last, c.qs, sizeof(double) * P);
        free(buf);
        return;
    }
    /* stepsize_adaptation */
    const double delta = 0.8, gamma = 0.05, t0 = 10.0, kappa = 0.75;
    c.eps = 1.0;
    double da_mu = log(10.0 * c.eps), s_bar = 0.0, x_bar = 0.0, da_count = 0.0;
    /* opt-in `adapt = carry` of the device library (include/epx.h, epx_sampler_opts.reserved bit 1; not the
     * reference's behaviour): start from the step size the chain ended its previous call with and from
     * the site's pooled sample variances as metric; warm-up adapts the step size only */
    const int carry = eps_in <= 0 && carry_eps > 0 && carry_inv_e;
    if (eps_in > 0) {                      /* test hook: fixed step size / metric */
        c.eps = eps_in;
        if (inv_e_in) memcpy(c.inv_e, inv_e_in, sizeof(double) * P);
    } else {
        if (carry) {
            c.eps = carry_eps;
            da_mu = log(10.0 * c.eps);
            memcpy(c.inv_e, carry_inv_e, sizeof(double) * P);
        }
        init_stepsize(&c, 0);
    }
    var_adapt va;
    va_init(&va, warmup, P, vabuf);
    double eps_sum = 0.0, acc_sum = 0.0, depth_sum = 0.0;
    long nleap = 0; int kept = 0, ndiv = 0, npost = 0;
    for (int t = 0; t < iter; ++t) {
        if (dump) {
            for (int i = 0; i < g_dump_n; ++i) {
                if (g_dump_ts[i] != t) continue;
                double *r = dump + (size_t)i * dump_stride;
                const int Pm = (dump_stride - 20) / 4;
                r[0] = c.lps; r[1] = c.eps; r[2] = da_mu; r[3] = s_bar; r[4] = x_bar; r[5] = da_count; r[6] = va.nw;
                r[7] = eps_sum; r[8] = acc_sum; r[9] = depth_sum; r[10] = (double)nleap; r[11] = (double)c.ngrad; r[12] = t;
                r[13] = va.counter; r[14] = va.window_size; r[15] = va.next_window; r[16] = ndiv; r[17] = npost; r[18] = kept;
                r[19] = 0.0;
                memcpy(r + 20, c.qs, sizeof(double) * P);
                memcpy(r + 20 + Pm, va.mean, sizeof(double) * P);
                memcpy(r + 20 + 2 * Pm, va.m2, sizeof(double) * P);
                memcpy(r + 20 + 3 * Pm, c.inv_e, sizeof(double) * P);
            }
        }
        const double eps_used = c.eps;
        trans_info ti = transition(&c, (uint32_t)(t + t_offset + 1));
        eps_sum += c.eps;
        double eps_learned = c.eps;
        nleap += ti.nleap;
        if (t < warmup) {
            /* learn_stepsize */
            da_count += 1.0;
            double a = ti.accept > 1 ? 1 : ti.accept;
            double eta = 1.0 / (da_count + t0);
            s_bar = (1.0 - eta) * s_bar + eta * (delta - a);
            double x = da_mu - s_bar * sqrt(da_count) / gamma;
            double x_eta = pow(da_count, -kappa);
            x_bar = (1.0 - x_eta) * x_bar + x_eta * x;
            c.eps = exp(x);
            eps_learned = c.eps;
            if (!carry && va_learn(&va, c.inv_e, c.qs, P)) {
                init_stepsize(&c, (uint32_t)(t + 1));
                da_mu = log(10.0 * c.eps);
                da_count = 0; s_bar = 0; x_bar = 0;
            }
            if (t == warmup - 1) { c.eps = exp(x_bar); eps_learned = c.eps; }    /* complete_adaptation */
        } else {
            acc_sum += ti.accept; depth_sum += ti.depth; ndiv += ti.divergent; ++npost;
            if ((t - warmup) % thin == 0) {
                memcpy(draws + (size_t)kept * P, c.qs, sizeof(double) * P);
                ++kept;
            }
        }
        if (trace) {
            double *tr = trace + (size_t)t * trace_stride;
            double ms = 0.0;
            for (int i = 0; i < P; ++i) ms += c.inv_e[i];
            tr[0] = eps_used; tr[1] = ti.nleap; tr[2] = ti.accept; tr[3] = ti.depth; tr[4] = ti.divergent;
            tr[5] = eps_learned; tr[6] = ms; tr[7] = c.lps;
            memcpy(tr + 8, c.qs, sizeof(double) * P);
        }
    }
    memcpy(last, c.qs, sizeof(double) * P);
    stats[ST_STEPSIZE_MEAN] = eps_sum / (iter > 0 ? iter : 1);
    stats[ST_STEPSIZE_FINAL] = c.eps;
    stats[ST_NLEAP] = (double)nleap;
    stats[ST_NGRAD] = (double)c.ngrad;
    stats[ST_NDIV] = ndiv;
    stats[ST_ACCEPT_MEAN] = npost ? acc_sum / npost : 0.0;
    stats[ST_DEPTH_MEAN] = npost ? depth_sum / npost : 0.0;
    free(buf);
}

/*
 * Sample `nsites` independent sites (contiguous row blocks k_lim[k]..k_lim[k+1]
 * of X/y, like method.py:829-830).
 *   mu:    nsites x d        cavity means
 *   Omega: nsites x d x d    cavity precisions (symmetric)
 *   seeds: nsites            the per-site "stan seed" of method.py:346
 *   init:  nsites x chains x P or NULL (init='random')
 *   draws: nsites x chains x nkeep x P
 *   last:  nsites x chains x P
 *   stats: nsites x chains x ST_COUNT
 * (site, chain) pairs are spread over `nthreads` OpenMP threads.
 */
/* Group structure for the *_groups entry points: g_cnt[k] groups in site k (NULL: one each),
 * g_lim = row limits of ALL groups in site order (sum(g_cnt)+1 entries, absolute rows).  The
 * per-(site, chain) records of init / draws / last use the stride Pmax = max_k P_k. */
static int sites_pmax(int model, int D, int nsites, const int32_t *g_cnt) {
    int pm = -1;
    for (int k = 0; k < nsites; ++k) {
        const int p = epo_npar_groups(model, D, g_cnt ? g_cnt[k] : 1);
        if (p < 0) return -1;
        if (p > pm) pm = p;
    }
    return pm;
}
int epo_sites_pmax(int model, int D, int nsites, const int32_t *g_cnt) { return sites_pmax(model, D, nsites, g_cnt); }

static void bind_site(site_t *s, int model, int D, int d, int k, const int64_t *k_lim, const double *X,
                      const int32_t *y, const double *mu, const double *Omega, const int32_t *g_cnt,
                      const int64_t *g_off, const int64_t *g_lim, int64_t *gl_rel) {
    s->model = model; s->D = D; s->d = d;
    s->n = (int)(k_lim[k + 1] - k_lim[k]);
    s->X = X + (size_t)k_lim[k] * D; s->y = y + k_lim[k];
    s->yd = NULL;
    if (is_gauss(model)) { s->yd = (const double *)(const void *)y + k_lim[k]; s->y = NULL; }
    s->mu = mu + (size_t)k * d; s->Om = Omega + (size_t)k * d * d;
    s->beta = s->db = s->da = s->Ov = NULL;
    s->ng = g_cnt ? g_cnt[k] : 1;
    s->gl = NULL;
    if (g_cnt) {
        for (int j = 0; j <= s->ng; ++j) gl_rel[j] = g_lim[g_off[k] + j] - k_lim[k];
        s->gl = gl_rel;
    }
    s->P = epo_npar_groups(model, D, s->ng);
}

/* carry_eps (nsites x chains, <= 0: none) / carry_metric (nsites x Pm): the `adapt = carry` history, or NULL */
int epo_nuts_sites_carry(int model, int nsites, int D, const int64_t *k_lim, const int32_t *g_cnt,
                         const int64_t *g_lim, const double *X,
                         const int32_t *y, const double *mu, const double *Omega,
                         const int64_t *seeds, int chains, int iter, int warmup, int thin,
                         int max_depth, const double *init, double *draws, double *last,
                         double *stats, int nthreads, const double *carry_eps, const double *carry_metric) {
    const int d = epo_dphi(model, D), Pm = sites_pmax(model, D, nsites, g_cnt);
    if (d < 0 || Pm < 0 || max_depth > EPO_MAX_DEPTH_CAP || thin < 1 || warmup > iter) return -1;
    const int nkeep = (iter - warmup + thin - 1) / thin;
    const long njobs = (long)nsites * chains;
    int64_t *g_off = (int64_t *)malloc(sizeof(int64_t) * (nsites + 1));
    g_off[0] = 0;
    int ngmax = 1;
    for (int k = 0; k < nsites; ++k) {
        const int g = g_cnt ? g_cnt[k] : 1;
        g_off[k + 1] = g_off[k] + g;
        if (g > ngmax) ngmax = g;
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (long job = 0; job < njobs; ++job) {
        int k = (int)(job / chains), c = (int)(job % chains);
        site_t s;
        int64_t *gl_rel = (int64_t *)malloc(sizeof(int64_t) * (ngmax + 1));
        bind_site(&s, model, D, d, k, k_lim, X, y, mu, Omega, g_cnt, g_off, g_lim, gl_rel);
        size_t jc = (size_t)k * chains + c;
        /* records are Pm wide; a site with fewer coordinates uses the leading P_k of each */
        double *dr = (double *)calloc((size_t)nkeep * s.P + 2 * (size_t)s.P, sizeof(double));
        double *la = dr + (size_t)nkeep * s.P, *in0 = la + s.P;
        if (init) memcpy(in0, init + jc * Pm, sizeof(double) * s.P);
        run_chain(&s, (uint64_t)seeds[k], c, iter, warmup, thin, max_depth,
                  init ? in0 : NULL, dr, la, stats + jc * ST_COUNT, -1.0, NULL, 0,
                  carry_eps ? carry_eps[jc] : -1.0, carry_metric ? carry_metric + (size_t)k * Pm : NULL,
                  (g_trace && k < g_trace_sites) ? g_trace + jc * (size_t)iter * (8 + Pm) : NULL, 8 + Pm,
                  (g_dump && k < g_dump_sites) ? g_dump + jc * (size_t)g_dump_n * (20 + 4 * Pm) : NULL, 20 + 4 * Pm);
        for (int t = 0; t < nkeep; ++t) {
            double *dst = draws + (jc * nkeep + t) * Pm;
            memcpy(dst, dr + (size_t)t * s.P, sizeof(double) * s.P);
            for (int e = s.P; e < Pm; ++e) dst[e] = 0.0;
        }
        memcpy(last + jc * Pm, la, sizeof(double) * s.P);
        for (int e = s.P; e < Pm; ++e) last[jc * Pm + e] = 0.0;
        free(dr); free(gl_rel);
    }
    free(g_off);
    return 0;
}
int epo_nuts_sites_groups(int model, int nsites, int D, const int64_t *k_lim, const int32_t *g_cnt,
                          const int64_t *g_lim, const double *X,
                          const int32_t *y, const double *mu, const double *Omega,
                          const int64_t *seeds, int chains, int iter, int warmup, int thin,
                          int max_depth, const double *init, double *draws, double *last,
                          double *stats, int nthreads) {
    return epo_nuts_sites_carry(model, nsites, D, k_lim, g_cnt, g_lim, X, y, mu, Omega, seeds, chains, iter, warmup,
                                thin, max_depth, init, draws, last, stats, nthreads, NULL, NULL);
}
int epo_nuts_sites(int model, int nsites, int D, const int64_t *k_lim, const double *X,
                   const int32_t *y, const double *mu, const double *Omega,
                   const int64_t *seeds, int chains, int iter, int warmup, int thin,
                   int max_depth, const double *init, double *draws, double *last,
                   double *stats, int nthreads) {
    return epo_nuts_sites_groups(model, nsites, D, k_lim, NULL, NULL, X, y, mu, Omega, seeds, chains, iter,
                                 warmup, thin, max_depth, init, draws, last, stats, nthreads);
}

/* TEST HOOK: `nt` plain transitions (no adaptation) per (site, chain) from given
 * positions q0 with given step sizes eps (nsites x chains) and diagonal inverse
 * metrics inv_e (nsites x chains x P); the random stream is the one a full run
 * uses at transition t_offset, t_offset+1, ...  draws: nsites x chains x nt x P. */
int epo_nuts_transitions_groups(int model, int nsites, int D, const int64_t *k_lim, const int32_t *g_cnt,
                                const int64_t *g_lim, const double *X,
                                const int32_t *y, const double *mu, const double *Omega,
                                const int64_t *seeds, int chains, int nt, int t_offset, int max_depth,
                                const double *q0, const double *eps, const double *inv_e,
                                double *draws, double *last, double *stats) {
    const int d = epo_dphi(model, D), Pm = sites_pmax(model, D, nsites, g_cnt);
    if (d < 0 || Pm < 0 || max_depth > EPO_MAX_DEPTH_CAP) return -1;
    const long njobs = (long)nsites * chains;
    int64_t *g_off = (int64_t *)malloc(sizeof(int64_t) * (nsites + 1));
    g_off[0] = 0;
    int ngmax = 1;
    for (int k = 0; k < nsites; ++k) {
        const int g = g_cnt ? g_cnt[k] : 1;
        g_off[k + 1] = g_off[k] + g;
        if (g > ngmax) ngmax = g;
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (long job = 0; job < njobs; ++job) {
        int k = (int)(job / chains), c = (int)(job % chains);
        site_t s;
        int64_t *gl_rel = (int64_t *)malloc(sizeof(int64_t) * (ngmax + 1));
        bind_site(&s, model, D, d, k, k_lim, X, y, mu, Omega, g_cnt, g_off, g_lim, gl_rel);
        size_t jc = (size_t)k * chains + c;
        double *dr = (double *)calloc((size_t)nt * s.P + 3 * (size_t)s.P, sizeof(double));
        double *la = dr + (size_t)nt * s.P, *in0 = la + s.P, *ie = in0 + s.P;
        memcpy(in0, q0 + jc * Pm, sizeof(double) * s.P);
        memcpy(ie, inv_e + jc * Pm, sizeof(double) * s.P);
        run_chain(&s, (uint64_t)seeds[k], c, nt, 0, 1, max_depth, in0, dr, la, stats + jc * ST_COUNT, eps[jc],
                  ie, t_offset, -1.0, NULL, NULL, 0, NULL, 0);
        for (int t = 0; t < nt; ++t) {
            double *dst = draws + (jc * nt + t) * Pm;
            memcpy(dst, dr + (size_t)t * s.P, sizeof(double) * s.P);
            for (int e = s.P; e < Pm; ++e) dst[e] = 0.0;
        }
        memcpy(last + jc * Pm, la, sizeof(double) * s.P);
        for (int e = s.P; e < Pm; ++e) last[jc * Pm + e] = 0.0;
        free(dr); free(gl_rel);
    }
    free(g_off);
    return 0;
}
int epo_nuts_transitions(int model, int nsites, int D, const int64_t *k_lim, const double *X,
                         const int32_t *y, const double *mu, const double *Omega,
                         const int64_t *seeds, int chains, int nt, int t_offset, int max_depth,
                         const double *q0, const double *eps, const double *inv_e,
                         double *draws, double *last, double *stats) {
    return epo_nuts_transitions_groups(model, nsites, D, k_lim, NULL, NULL, X, y, mu, Omega, seeds, chains, nt,
                                       t_offset, max_depth, q0, eps, inv_e, draws, last, stats);
}

int epo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* raw uniforms / normals of the shared stream, for RNG parity tests */
void epo_rng_probe(uint64_t seed, int chain, uint32_t t, uint32_t kind, uint32_t a, uint32_t b,
                   double *u1, double *u2, double *n0, double *n1) {
    rng_key k = make_key(seed, chain);
    rng_u2(k, t, kind, a, b, u1, u2);
    *n0 = rng_normal(k, t, kind, (int)(2 * a), b);
    *n1 = rng_normal(k, t, kind, (int)(2 * a + 1), b);
}
