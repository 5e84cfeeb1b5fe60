// Batched small dense linear algebra of the EP update on gfx950: one site (one
// d x d problem) per workgroup, matrices staged in LDS when 2*d*(d|1) doubles
// fit, otherwise in an L2-resident global workspace (d = 258 at config C5).
//
// Replaces the SciPy/LAPACK calls of /root/reference/epstan/method.py and
// util.py (potrf / potrs / potri, dgeqrf -> scatter + potrf, gemm C'C) and the
// Cython helpers fro_norm_squared / copy_triu_to_tril (cython_util.pyx:17-40,
// 86-106), which are fused into the kernels below.
#include "epx_device.h"
#include "epx_kernels.h"

namespace epx {

// ------------------------------------------------------------------ helpers
__device__ inline double block_sum(double v, double *red /* >= 16 doubles LDS */) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}

// In-place lower Cholesky of the column-major matrix W (ld >= d), reading and
// writing the LOWER triangle only.  All threads of the block take part.
// Returns false (uniformly) when a pivot is <= 0 or NaN (LAPACK dpotrf rule).
__device__ bool block_potrf(double *W, int d, int ld) {
    const int tid = threadIdx.x, T = blockDim.x;
    for (int j = 0; j < d; ++j) {
        __syncthreads();
        const double ajj = W[j + (size_t)j * ld];
        if (!(ajj > 0.0) || isinf(ajj)) return false;
        const double ljj = sqrt(ajj);
        const double inv = 1.0 / ljj;
        __syncthreads();
        for (int i = j + tid; i < d; i += T)
            W[i + (size_t)j * ld] = (i == j) ? ljj : W[i + (size_t)j * ld] * inv;
        __syncthreads();
        const int m = d - j - 1;
        for (int idx = tid; idx < m * m; idx += T) {
            const int ii = idx % m, kk = idx / m;
            if (kk <= ii) {
                const int i = j + 1 + ii, k = j + 1 + kk;
                W[i + (size_t)k * ld] -= W[i + (size_t)j * ld] * W[k + (size_t)j * ld];
            }
        }
    }
    __syncthreads();
    return true;
}

// x <- (L L')^-1 x, L in the lower triangle of W; x has d entries (LDS or global).
__device__ void block_potrs(const double *W, int d, int ld, double *x) {
    const int tid = threadIdx.x, T = blockDim.x;
    for (int j = 0; j < d; ++j) {            // L y = b
        __syncthreads();
        const double xj = x[j] / W[j + (size_t)j * ld];
        __syncthreads();
        if (tid == 0) x[j] = xj;
        for (int i = j + 1 + tid; i < d; i += T) x[i] -= W[i + (size_t)j * ld] * xj;
    }
    for (int j = d - 1; j >= 0; --j) {       // L' x = y
        __syncthreads();
        const double xj = x[j] / W[j + (size_t)j * ld];
        __syncthreads();
        if (tid == 0) x[j] = xj;
        for (int i = tid; i < j; i += T) x[i] -= W[j + (size_t)i * ld] * xj;
    }
    __syncthreads();
}

// W <- (L L')^-1, full symmetric, from L in the lower triangle of W; V is a
// d x d workspace (same ld).  dpotri + copy_triu_to_tril of util.py:118-124.
__device__ void block_potri(double *W, double *V, int d, int ld) {
    const int tid = threadIdx.x, T = blockDim.x;
    // column t of L^-1 by forward substitution, one thread per column
    for (int t = tid; t < d; t += T) {
        double *z = V + (size_t)t * ld;
        z[t] = 1.0 / W[t + (size_t)t * ld];
        for (int i = t + 1; i < d; ++i) {
            double s = 0.0;
            for (int k = t; k < i; ++k) s += W[i + (size_t)k * ld] * z[k];
            z[i] = -s / W[i + (size_t)i * ld];
        }
    }
    __syncthreads();
    // A^-1 = L^-T L^-1 : out(i,j) = sum_{k>=i} Linv(k,i) Linv(k,j), i >= j
    for (int idx = tid; idx < d * d; idx += T) {
        const int i = idx % d, j = idx / d;
        if (i >= j) {
            double s = 0.0;
            const double *zi = V + (size_t)i * ld, *zj = V + (size_t)j * ld;
            for (int k = i; k < d; ++k) s += zi[k] * zj[k];
            W[i + (size_t)j * ld] = s;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < d * d; idx += T) {
        const int i = idx % d, j = idx / d;
        if (i > j) W[j + (size_t)i * ld] = W[i + (size_t)j * ld];
    }
    __syncthreads();
}

__device__ inline void ws_pointers(const DenseWs &ws, int slot, int d, int ld, double *lds,
                                   double *&W, double *&V, double *&vec) {
    if (ws.use_lds) {
        W = lds;
        V = lds + (size_t)d * ld;
        vec = V + (size_t)d * ld;
    } else {
        W = ws.global + (size_t)slot * (2 * (size_t)d * ld + 4 * (size_t)ld);
        V = W + (size_t)d * ld;
        vec = V + (size_t)d * ld;
    }
}

// ------------------------------------------------------------------ cavity
// Worker.cavity (method.py:267-302) for one site per block.
__global__ void __launch_bounds__(256)
k_cavity(CavityArgs a) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double red[16];
    const int k = a.k0 + blockIdx.x;
    const int d = a.d, ld = a.ld, tid = threadIdx.x, T = blockDim.x;
    double *W, *V, *vec;
    ws_pointers(a.ws, blockIdx.x, d, ld, lds, W, V, vec);
    const double *Qk = a.Qsite + (size_t)k * a.site_stride;
    const double *dQk = a.dQsite ? a.dQsite + (size_t)k * a.site_stride : nullptr;
    const double *rk = a.rsite + (size_t)k * a.rsite_stride;
    const double *drk = a.drsite ? a.drsite + (size_t)k * a.rsite_stride : nullptr;
    double *Om = a.cav_Om + (size_t)k * d * d;
    for (int idx = tid; idx < d * d; idx += T) {
        const int i = idx % d, j = idx / d;
        double q = Qk[idx];
        if (dQk) q += a.df * dQk[idx];              // Qi2 = Qi + df dQi (method.py:1071)
        const double m = a.Q[idx] - q;              // Mat = Q - Qi     (method.py:288)
        W[i + (size_t)j * ld] = m;
        Om[idx] = m;
    }
    for (int i = tid; i < d; i += T) {
        double ri = rk[i];
        if (drk) ri += a.df * drk[i];
        vec[i] = a.r[i] - ri;                       // vec = r - ri     (method.py:289)
    }
    __syncthreads();
    const bool ok = block_potrf(W, d, ld);          // cho_factor       (method.py:294)
    if (ok) block_potrs(W, d, ld, vec);             // cho_solve        (method.py:295)
    __syncthreads();
    for (int i = tid; i < d; i += T) a.cav_mu[(size_t)k * d + i] = vec[i];
    if (tid == 0) a.flags[k] = ok ? 1 : 0;
    (void)red;
}

// ------------------------------------------------------------------ moments
// v_mfma_f64_16x16x4_f64: A is 16x4 (lane l: row l&15, k l>>4), B is 4x16
// (lane l: k l>>4, col l&15); D: col = l&15, row = (l>>4) + 4*reg.
typedef double v4d __attribute__((ext_vector_type(4)));

// Worker.tilted moment stage (method.py:410-475) for one site per block.
__global__ void __launch_bounds__(256)
k_moments(MomentArgs a) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double red[16];
    __shared__ double scal[8];
    const int k = a.k0 + blockIdx.x;
    const int d = a.d, ld = a.ld, tid = threadIdx.x, T = blockDim.x, S = a.S;
    double *W, *V, *vec;
    ws_pointers(a.ws, blockIdx.x, d, ld, lds, W, V, vec);
    double *mean = vec;            // d
    const double *X = a.draws + (size_t)(a.draws_site0 + blockIdx.x) * a.stride_site;
    const long ss = a.stride_s, si = a.stride_i;

    // ---- mean over the S draws (method.py:415) : thread = (coordinate, slice)
    {
        int nsl = T / d;                          // slices of the draw index
        if (nsl > d) nsl = d;
        if (nsl < 1) nsl = 1;
        for (int idx = tid; idx < d * nsl; idx += T) {       // d may exceed the block size
            const int i = idx % d, sl = idx / d;
            double s = 0.0;
            for (int t = sl; t < S; t += nsl) s += X[(long)t * ss + (long)i * si];
            V[i + (size_t)sl * ld] = s;          // slices are combined in a fixed order below
        }
        __syncthreads();
        for (int i = tid; i < d; i += T) {
            double s = 0.0;
            for (int sl = 0; sl < nsl; ++sl) s += V[i + (size_t)sl * ld];
            mean[i] = s / (double)S;
        }
        __syncthreads();
    }
    // ---- scatter C'C of the centred draws (method.py:417-420 / :444-446), MFMA f64
    {
        const int wave = tid >> 6, lane = tid & 63, nw = T >> 6;
        const int nt = (d + 15) / 16;
        const int ntile = nt * (nt + 1) / 2;
        for (int tile = wave; tile < ntile; tile += nw) {
            // unrank (ti <= tj)
            int ti = 0, rem = tile;
            while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
            const int tj = ti + rem;
            const int ia = ti * 16 + (lane & 15), ib = tj * 16 + (lane & 15);
            const double ma = ia < d ? mean[ia] : 0.0, mb = ib < d ? mean[ib] : 0.0;
            const int kq = lane >> 4;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int s0 = 0; s0 < S; s0 += 4) {
                const int s = s0 + kq;
                double av = 0.0, bv = 0.0;
                if (s < S) {
                    if (ia < d) av = X[(long)s * ss + (long)ia * si] - ma;
                    if (ib < d) bv = X[(long)s * ss + (long)ib * si] - mb;
                }
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + (lane >> 4) + 4 * r, col = tj * 16 + (lane & 15);
                if (row < d && col < d) {
                    W[row + (size_t)col * ld] = acc[r];
                    W[col + (size_t)row * ld] = acc[r];
                }
            }
        }
        __syncthreads();
    }
    double *scat = a.tilt_scatter + (size_t)k * d * d;
    for (int idx = tid; idx < d * d; idx += T) scat[idx] = W[(idx % d) + (size_t)(idx / d) * ld];
    for (int i = tid; i < d; i += T) a.tilt_mean[(size_t)k * d + i] = mean[i];
    __syncthreads();

    double *dQ = a.dQi + (size_t)k * d * d, *dr = a.dri + (size_t)k * d;
    bool ok;
    if (a.prec_estim == 0) {
        // 'sample': (R'R)^-1 (S-d-2)  (method.py:420-435)
        ok = block_potrf(W, d, ld);
        if (ok) {
            block_potri(W, V, d, ld);
            const double ub = (double)(S - d - 2);
            for (int i = tid; i < d; i += T) {
                double s = 0.0;
                for (int j = 0; j < d; ++j) s += W[i + (size_t)j * ld] * mean[j];
                dr[i] = s * ub - a.r[i];                       // :435, :458
            }
            for (int idx = tid; idx < d * d; idx += T)
                dQ[idx] = W[(idx % d) + (size_t)(idx / d) * ld] * ub - a.Q[idx];   // :434, :457
        }
    } else {
        // 'olse' (method.py:446-451, util.py:128-194) with prior matrix P = global Q
        const double invS = 1.0 / (double)S;
        for (int idx = tid; idx < d * d; idx += T) {
            const int i = idx % d, j = idx / d;
            if (i >= j) W[i + (size_t)j * ld] *= invS;
        }
        __syncthreads();
        ok = block_potrf(W, d, ld);
        if (ok) {
            block_potri(W, V, d, ld);
            double tr = 0.0, f2 = 0.0, f2p = 0.0, tsp = 0.0;
            for (int idx = tid; idx < d * d; idx += T) {
                const int i = idx % d, j = idx / d;
                const double w = W[i + (size_t)j * ld], p = a.Q[idx];
                if (i == j) tr += w;
                f2 += w * w; f2p += p * p; tsp += w * p;
            }
            tr = block_sum(tr, red); f2 = block_sum(f2, red);
            f2p = block_sum(f2p, red); tsp = block_sum(tsp, red);
            const double n = (double)S, dd = (double)d, tr2 = tr * tr;
            const double alpha = 1.0 - (dd + tr2 * f2p / (f2 * f2p - tsp * tsp)) / n;   // util.py:190
            const double beta = (tsp / f2p) * (1.0 - dd / n - alpha);                   // util.py:191
            __syncthreads();
            for (int idx = tid; idx < d * d; idx += T) {
                const int i = idx % d, j = idx / d;
                W[i + (size_t)j * ld] = alpha * W[i + (size_t)j * ld] + beta * a.Q[idx];
            }
            __syncthreads();
            for (int i = tid; i < d; i += T) {
                double s = 0.0;
                for (int j = 0; j < d; ++j) s += W[i + (size_t)j * ld] * mean[j];
                dr[i] = s - a.r[i];
            }
            for (int idx = tid; idx < d * d; idx += T)
                dQ[idx] = W[(idx % d) + (size_t)(idx / d) * ld] - a.Q[idx];
        }
    }
    if (!ok) {                                                   // method.py:460-465
        for (int idx = tid; idx < d * d; idx += T) dQ[idx] = 0.0;
        for (int i = tid; i < d; i += T) dr[i] = 0.0;
    }
    if (tid == 0) a.flags[k] = ok ? 1 : 0;
    (void)scal;
}

// ------------------------------------------------------- site sums (2 stage)
// partial[b][e] = sum over the sites of slice b of {Qi,ri,dQi,dri}[e]
__global__ void __launch_bounds__(256)
k_site_sums_partial(SumArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;     // element of the packed vector
    const int b = blockIdx.y;
    if (e >= a.len) return;
    const int d2 = a.d * a.d, d = a.d;
    const double *src; size_t stride; int off;
    if (e < d2) { src = a.Qi; stride = d2; off = e; }
    else if (e < d2 + d) { src = a.ri; stride = d; off = e - d2; }
    else if (e < 2 * d2 + d) { src = a.dQi; stride = d2; off = e - d2 - d; }
    else { src = a.dri; stride = d; off = e - 2 * d2 - d; }
    const int per = (a.K + a.nslice - 1) / a.nslice;
    const int kb = b * per, ke = min(a.K, kb + per);
    double s = 0.0;
    for (int k = kb; k < ke; ++k) s += src[(size_t)k * stride + off];
    a.partial[(size_t)b * a.len + e] = s;
}
__global__ void __launch_bounds__(256)
k_site_sums_final(SumArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.len) return;
    double s = 0.0;
    for (int b = 0; b < a.nslice; ++b) s += a.partial[(size_t)b * a.len + e];
    a.out[e] = s;
}

// Pooled tilted moments (Master.mix_phi, method.py:1250-1296): sums over the sites of the
// scatter matrices, of the means and of the outer products of the means; same slicing as above
// (a.Qi = tilted scatter, a.ri = tilted mean; len = 2 d^2 + d).
__global__ void __launch_bounds__(256)
k_mix_partial(SumArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (e >= a.len) return;
    const int d2 = a.d * a.d, d = a.d;
    const int per = (a.K + a.nslice - 1) / a.nslice;
    const int kb = b * per, ke = min(a.K, kb + per);
    double s = 0.0;
    if (e < d2) { for (int k = kb; k < ke; ++k) s += a.Qi[(size_t)k * d2 + e]; }
    else if (e < d2 + d) { for (int k = kb; k < ke; ++k) s += a.ri[(size_t)k * d + (e - d2)]; }
    else {
        const int i = (e - d2 - d) % d, j = (e - d2 - d) / d;
        for (int k = kb; k < ke; ++k) s += a.ri[(size_t)k * d + i] * a.ri[(size_t)k * d + j];
    }
    a.partial[(size_t)b * a.len + e] = s;
}

// ------------------------------------------------------------------ global
// Q = Q0 + sum Qi + df sum dQi, r likewise (method.py:1071-1074), Cholesky
// check (:1077-1080); with want_moments also S = Q^-1, m = S r (:1211-1216).
__global__ void __launch_bounds__(256)
k_global(GlobalArgs a) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double red[16];
    const int d = a.d, ld = a.ld, tid = threadIdx.x, T = blockDim.x;
    double *W, *V, *vec;
    ws_pointers(a.ws, 0, d, ld, lds, W, V, vec);
    const int d2 = d * d;
    if (a.packed) {
        const double *sQ = a.packed, *sr = a.packed + d2, *sdQ = a.packed + d2 + d,
                     *sdr = a.packed + 2 * d2 + d;
        for (int idx = tid; idx < d2; idx += T) a.Q[idx] = a.Q0[idx] + sQ[idx] + a.df * sdQ[idx];
        for (int i = tid; i < d; i += T) a.r[i] = a.r0[i] + sr[i] + a.df * sdr[i];
    }
    __syncthreads();
    for (int idx = tid; idx < d2; idx += T) W[(idx % d) + (size_t)(idx / d) * ld] = a.Q[idx];
    for (int i = tid; i < d; i += T) vec[i] = a.r[i];
    __syncthreads();
    const bool ok = block_potrf(W, d, ld);
    if (tid == 0) *a.flag = ok ? 1 : 0;
    double half_logdet_Q = 0.0;
    if (ok && a.crit) {
        double t = 0.0;
        for (int j = tid; j < d; j += T) t += log(W[j + (size_t)j * ld]);
        half_logdet_Q = block_sum(t, red);
    }
    if (ok && a.want_moments) {
        block_potrs(W, d, ld, vec);
        block_potri(W, V, d, ld);
        for (int idx = tid; idx < d2; idx += T) a.S[idx] = W[(idx % d) + (size_t)(idx / d) * ld];
        for (int i = tid; i < d; i += T) a.m[i] = vec[i];
    }
    if (a.crit) {
        // selection criteria of find_damp.py:155-163 with S^-1 = Q:
        //   mse = mean((m - m_t)^2)
        //   KL(N(m_t,S_t) || N(m,S)) = (tr(Q S_t) + dm'Q dm - d)/2 - logdet(S_t)/2 - logdet(Q)/2      (kl_mvn, :38-56)
        //   ll  = sum_s log N(x_s | m, S) = -n/2 (d log 2pi - logdet Q) - (tr(Q Sc) + n (xbar-m)'Q(xbar-m))/2
        double mse = NAN, kl = NAN, ll = NAN;
        if (ok && a.want_moments) {
            const double *mt = a.tgt, *St = a.tgt + d, *xb = St + d2, *Sc = xb + d;
            const double hl_St = Sc[d2], ns = Sc[d2 + 1];
            double e2 = 0.0, trt = 0.0, trc = 0.0, qt = 0.0, qc = 0.0;
            for (int i = tid; i < d; i += T) { const double e = vec[i] - mt[i]; e2 += e * e; }
            for (int idx = tid; idx < d2; idx += T) {
                const int i = idx % d, j = idx / d;
                const double q = a.Q[idx];
                trt += q * St[idx];
                qt += q * (vec[i] - mt[i]) * (vec[j] - mt[j]);
                if (ns > 0.0) { trc += q * Sc[idx]; qc += q * (xb[i] - vec[i]) * (xb[j] - vec[j]); }
            }
            e2 = block_sum(e2, red); trt = block_sum(trt, red); qt = block_sum(qt, red);
            trc = block_sum(trc, red); qc = block_sum(qc, red);
            mse = e2 / d;
            kl = 0.5 * (trt + qt - d) - hl_St - half_logdet_Q;
            if (ns > 0.0) ll = -0.5 * ns * (d * 1.8378770664093454836 - 2.0 * half_logdet_Q) - 0.5 * (trc + ns * qc);
        }
        if (tid == 0) { a.crit[0] = ok ? 1.0 : 0.0; a.crit[1] = 0.0; a.crit[2] = mse; a.crit[3] = kl; a.crit[4] = ll; }
    }
}

// second half of a sweep entry: all-cavities flag of the trial (find_damp.py:150-153)
__global__ void k_sweep_flag(const int *all_flag, double *crit) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        crit[1] = (crit[0] != 0.0 && *all_flag) ? 1.0 : 0.0;
        if (crit[1] == 0.0) { crit[2] = NAN; crit[3] = NAN; crit[4] = NAN; }
    }
}

// Qi <- Qi + df dQi ; ri <- ri + df dri   (accept, method.py:1145-1158)
// or Qi2 <- Qi + df dQi when out != Qi
__global__ void k_axpy(double *out, const double *x, const double *dx, double df, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = x[i] + df * dx[i];
}

// flags of one damping trial as doubles, so that ONE all-reduce(min) makes them global:
// out = [global_pd, all local cavities pd (0 when the global check failed), first failing site
// (global index) or 1e18]
__global__ void k_trial_flags(const uint8_t *flags, int count, int site_base, const int *global_pd, double *out) {
    __shared__ int first;
    if (threadIdx.x == 0) first = 0x7fffffff;
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += blockDim.x)
        if (!flags[i]) atomicMin(&first, i);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = *global_pd;
        out[0] = g ? 1.0 : 0.0;
        out[1] = (g && first == 0x7fffffff) ? 1.0 : 0.0;
        out[2] = (g && first != 0x7fffffff) ? (double)(site_base + first) : 1e18;
    }
}
__global__ void k_all_flags(const uint8_t *flags, int k0, int count, int *out /* [2]: all, first_bad */) {
    __shared__ int first;
    if (threadIdx.x == 0) first = 0x7fffffff;
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += blockDim.x)
        if (!flags[k0 + i]) atomicMin(&first, i);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = first == 0x7fffffff ? 1 : 0; out[1] = first == 0x7fffffff ? -1 : first; }
}

// ------------------------------------------------ util.invert_normal_params
// (util.py:51-125) batched: one matrix per block, in place, F-order.
__global__ void __launch_bounds__(256)
k_invert(InvertArgs a) {
    extern __shared__ __align__(16) double lds[];
    const int b = blockIdx.x, d = a.d, ld = a.ld, tid = threadIdx.x, T = blockDim.x;
    double *W, *V, *vec;
    ws_pointers(a.ws, b, d, ld, lds, W, V, vec);
    double *A = a.A + (size_t)b * d * d;
    double *x = a.b ? a.b + (size_t)b * d : nullptr;
    for (int idx = tid; idx < d * d; idx += T) {
        const int i = idx % d, j = idx / d;
        // cho_form: caller gives the UPPER factor U (A = U'U); L = U'
        if (i >= j) W[i + (size_t)j * ld] = a.cho_form ? A[j + (size_t)i * d] : A[idx];
    }
    if (x) for (int i = tid; i < d; i += T) vec[i] = x[i];
    __syncthreads();
    bool ok = true;
    if (!a.cho_form) ok = block_potrf(W, d, ld);
    else {
        // dpotri reports an exactly singular factor (util.py:118-122)
        int bad = 0;
        for (int i = tid; i < d; i += T) {
            const double v = W[i + (size_t)i * ld];
            if (v == 0.0 || !isfinite(v)) bad = 1;
        }
        ok = !__syncthreads_or(bad);
    }
    if (ok) {
        if (x) block_potrs(W, d, ld, vec);
        block_potri(W, V, d, ld);
        for (int idx = tid; idx < d * d; idx += T) A[idx] = W[(idx % d) + (size_t)(idx / d) * ld];
        if (x) for (int i = tid; i < d; i += T) x[i] = vec[i];
    }
    if (tid == 0) a.info[b] = ok ? 0 : 1;
}

// ------------------------------------------------------------- util.olse
// (util.py:128-194) batched, in place on the sample covariances.
__global__ void __launch_bounds__(256)
k_olse(OlseArgs a) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double red[16];
    const int b = blockIdx.x, d = a.d, ld = a.ld, tid = threadIdx.x, T = blockDim.x;
    double *W, *V, *vec;
    ws_pointers(a.ws, b, d, ld, lds, W, V, vec);
    double *A = a.S + (size_t)b * d * d;
    const double *P = a.P ? a.P + (size_t)b * d * d : nullptr;
    for (int idx = tid; idx < d * d; idx += T) W[(idx % d) + (size_t)(idx / d) * ld] = A[idx];
    __syncthreads();
    const bool ok = block_potrf(W, d, ld);
    if (ok) {
        block_potri(W, V, d, ld);
        double tr = 0.0, f2 = 0.0, f2p = 0.0, tsp = 0.0;
        for (int idx = tid; idx < d * d; idx += T) {
            const int i = idx % d, j = idx / d;
            const double w = W[i + (size_t)j * ld];
            if (i == j) tr += w;
            f2 += w * w;
            if (P) { f2p += P[idx] * P[idx]; tsp += w * P[idx]; }
        }
        tr = block_sum(tr, red); f2 = block_sum(f2, red);
        const double n = (double)a.n, dd = (double)d, tr2 = tr * tr;
        if (!P) {
            const double alpha = 1.0 - (dd + tr2 / (f2 - tr2 / dd)) / n;     // util.py:181
            const double beta = tr * (1.0 - dd / n - alpha);                 // util.py:182
            for (int idx = tid; idx < d * d; idx += T) {
                const int i = idx % d, j = idx / d;
                A[idx] = alpha * W[i + (size_t)j * ld] + (i == j ? beta / dd : 0.0);
            }
        } else {
            f2p = block_sum(f2p, red); tsp = block_sum(tsp, red);
            const double alpha = 1.0 - (dd + tr2 * f2p / (f2 * f2p - tsp * tsp)) / n;
            const double beta = (tsp / f2p) * (1.0 - dd / n - alpha);
            for (int idx = tid; idx < d * d; idx += T)
                A[idx] = alpha * W[(idx % d) + (size_t)(idx / d) * ld] + beta * P[idx];
        }
    }
    if (tid == 0) a.info[b] = ok ? 0 : 1;
}

// ---------------------------------------------------------------- force pd
// Smallest eigenvalue of Qi + df dQi per site by bisection on "A - s I is
// pos.def." (method.py:1124-1129 uses eigvalsh(..., eigvals=(0,0))); where it
// is below thresh the diagonal of Qi gets (target - min_eig).
__global__ void __launch_bounds__(256)
k_force_pd(ForceArgs a) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double red[16];
    const int k = blockIdx.x, d = a.d, ld = a.ld, tid = threadIdx.x, T = blockDim.x;
    double *W, *V, *vec;
    ws_pointers(a.ws, k, d, ld, lds, W, V, vec);
    double *Qk = a.Qi + (size_t)k * d * d;
    const double *dQk = a.dQi + (size_t)k * d * d;
    // Gershgorin bounds
    double lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < d; i += T) {
        double c = 0.0, rad = 0.0;
        for (int j = 0; j < d; ++j) {
            const double v = Qk[i + (size_t)j * d] + a.df * dQk[i + (size_t)j * d];
            if (j == i) c = v; else rad += fabs(v);
        }
        lo = fmin(lo, c - rad); hi = fmax(hi, c + rad);
    }
    for (int m = 32; m >= 1; m >>= 1) { lo = fmin(lo, __shfl_xor(lo, m, 64)); hi = fmax(hi, __shfl_xor(hi, m, 64)); }
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = lo; red[8 + (tid >> 6)] = hi; }
    __syncthreads();
    for (int w = 0; w < (T >> 6); ++w) { lo = fmin(lo, red[w]); hi = fmax(hi, red[8 + w]); }
    // invariant: A - lo I pos.def. (lo <= lambda_min), A - hi I is not
    const double span = fmax(hi - lo, 1e-300);
    lo -= 1e-12 * span + 1e-300;
    for (int it = 0; it < 64; ++it) {
        const double mid = 0.5 * (lo + hi);
        __syncthreads();
        for (int idx = tid; idx < d * d; idx += T) {
            const int i = idx % d, j = idx / d;
            if (i >= j) W[i + (size_t)j * ld] = Qk[idx] + a.df * dQk[idx] - (i == j ? mid : 0.0);
        }
        __syncthreads();
        if (block_potrf(W, d, ld)) lo = mid; else hi = mid;
        if (hi - lo <= 4e-16 * fmax(fabs(lo), fabs(hi))) break;
    }
    const double min_eig = 0.5 * (lo + hi);
    const bool force = min_eig < a.thresh;
    __syncthreads();
    if (force) for (int i = tid; i < d; i += T) Qk[i + (size_t)i * d] += a.target - min_eig;
    if (tid == 0) { a.forced[k] = force ? 1 : 0; a.min_eig[k] = min_eig; }
}

}  // namespace epx
