// em_f64.hip -- EM / MAP iterations in float64 on the device for SHORT data and models of any size: what a speaker's MAP enrolment
// from a large UBM is (gmmubm.cc:29-81: 512 - 2048 mixtures adapted on one utterance of ~3000 frames; interface.py:55-109).
//
// Why: iteration at a time through the scoring engines (em.hip) such an iteration is ~0.25 ms of kernels inside ~3 ms of host work
// (K = 2048, 3000 frames): the model packed into the engines' layouts again (1.1 ms), and -- most mixtures of a large UBM see next to
// nothing of a short utterance -- the sums of every mixture whose responsibilities are 0 in fp32 and tiny in the reference's float64
// formed again on host threads (0.7 ms).  In float64 there is nothing to pack and nothing to redo: the model stays on the device as
// plain arrays, an iteration is five or six launches, and the host waits -- for 16 bytes -- only where the stop rule wants the total.
//
//   e64_density  (128 frames, block of 64 mixtures; a thread = 2 frames x 16 mixtures): log densities -> L[k][frame]; the block's
//                maximum and sum of exponentials per frame
//   e64_lse      (64 frames): the frame's total from its blocks' pairs (a term below DBL_MIN is 0, a frame without a surviving term
//                carries no responsibility and counts ln 1e-15: gmm.cc:482-498, :34-38, lse.hpp); the chunk's sum and flag
//   e64_stats    (64 frames, block of 64 mixtures): responsibilities exp(lp - ll), the three sums of the block's mixtures over the
//                chunk's frames -> partial[chunk][k][2 D + 1]
//   e64_head     total log-likelihood and the (accumulating) flag
//   e64_mstep    sums over the chunks in order + the M-step of gmm.cc:388-437 / gmmubm.cc:53-74 (em.hip's host M-step restated) in place
//   e64_weights  (EM only) weights N_k / n normalised by their sum in mixture order, the mixtures' constants
// The stop rule (gmm.cc:622-650) reads the total under the updated model off the NEXT iteration's e64_head, as em_small.hip does; a
// live frame within 110 nats of the underflow boundary hands the fit to the iteration-at-a-time path (partial-product flushes).
#include "score.hpp"
#include "wave_ops.hpp"

#include "../../include/pygmm_hip.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <limits>
#include <vector>

namespace sr {

namespace {

constexpr int E64_FR = 64, E64_KB = 64, E64_THREADS = 256, E64_PER = E64_KB / 4;     // density: a thread = two frames x 16 mixtures
constexpr int E64_DFR = 128;                                                          // frames of a density workgroup
constexpr int E64_STHREADS = 512, E64_SPER = E64_KB / 8;                              // statistics: a thread = one frame x 8 mixtures
constexpr int E64_MAX_D = 64;
constexpr long E64_MAX_FRAMES = 8192, E64_MAX_CELLS = 32L << 20;                     // L: <= 256 MB
constexpr double E64_MINLOG = -708.396418532264, E64_BAND = -598.0, E64_LN_1E_15 = -34.538776394910684, E64_SQRT_2_PI = 2.5066282746310002;

struct E64Args {
    const float *X;
    int n, n_pad, dim, K, n_chunks, n_kb, map;
    double *w, *mu, *sg, *h, *c;       // the model on the device: [K], [K][D], [K][D], 1 / (2 sigma^2), ln w - sum ln(sqrt(2 pi) sigma)
    const double *ubm_mu;
    double *L;                         // [n_kb * 64][n_pad]
    double *mb, *sb;                   // [n_kb][n_pad]
    double *llf;                       // [n_pad]  a frame's total; +inf where it carries no responsibility
    double *partial;                   // [n_chunks][K][2 D + 1]
    double *llpart;                    // [n_chunks][2]
    double *head;                      // [2]
    double min_sigma, relevance;
};

__global__ __launch_bounds__(256)
void e64_derive_kernel(const E64Args a, int what /* 1: h, 2: c, 3: both */) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int D = a.dim;
    if ((what & 1) && i < a.K * D) a.h[i] = 0.5 / (a.sg[i] * a.sg[i]);
    if ((what & 2) && i < a.K) {
        double c = a.w[i] > 0.0 ? log(a.w[i]) : -__builtin_inf();
        for (int d = 0; d < D; d++) c -= log(E64_SQRT_2_PI * a.sg[i * D + d]);
        a.c[i] = c;
    }
}

__global__ __launch_bounds__(E64_THREADS)
void e64_density_kernel(const E64Args a) {
    extern __shared__ __attribute__((aligned(16))) double e64_lds[];
    const int D = a.dim, XS = D + 1;
    double2 *s_p = reinterpret_cast<double2 *>(e64_lds);          // [64][D] {mean, 1 / (2 sigma^2)}: one 16-byte read per pair
    double *s_c = e64_lds + 2 * E64_KB * D;          // [64]
    double *s_pm = s_c + E64_KB;                     // [4][128]
    float *s_x = reinterpret_cast<float *>(s_pm + 4 * E64_DFR);    // [128][D + 1]
    const int tid = threadIdx.x, f = tid & 63, g = tid >> 6;
    const int f0 = blockIdx.x * E64_DFR, kb = blockIdx.y, k0 = kb * E64_KB;
    for (int i = tid; i < E64_KB * D; i += E64_THREADS) {
        const int k = k0 + i / D;
        s_p[i] = k < a.K ? make_double2(a.mu[(size_t)k0 * D + i], a.h[(size_t)k0 * D + i]) : make_double2(0.0, 0.0);
    }
    if (tid < E64_KB) s_c[tid] = k0 + tid < a.K ? a.c[k0 + tid] : -__builtin_inf();
    for (int i = tid; i < E64_DFR * D; i += E64_THREADS) {
        const int fr = i / D, d = i - fr * D;
        s_x[fr * XS + d] = f0 + fr < a.n ? a.X[(size_t)(f0 + fr) * D + d] : 0.f;
    }
    __syncthreads();
    // frames f and f + 64 against this wave's 16 mixtures: a parameter pair read from LDS serves two frames (the reads, not the
    // arithmetic, bound this kernel: 2048 x 3000 x 39 takes 100 us with a read per value, 68-75 with a mixture's {mean, h} as ONE
    // 16-byte read; the loop 60 of them, the exponentials 7, the stores of L 6 -- parts switched off, profiles/r06_em_f64.txt)
    double lp0[E64_PER], lp1[E64_PER];
#pragma unroll
    for (int j = 0; j < E64_PER; j++) lp0[j] = lp1[j] = s_c[g * E64_PER + j];
    for (int d = 0; d < D; d++) {
        const double x0 = (double)s_x[f * XS + d], x1 = (double)s_x[(f + 64) * XS + d];
#pragma unroll
        for (int j = 0; j < E64_PER; j++) {
            const double2 ph = s_p[(g * E64_PER + j) * D + d];
            const double mu = ph.x, h = ph.y;
            const double t0 = x0 - mu, t1 = x1 - mu;
            lp0[j] = fma(-(t0 * t0), h, lp0[j]);
            lp1[j] = fma(-(t1 * t1), h, lp1[j]);
        }
    }
    double pm0 = -__builtin_inf(), pm1 = -__builtin_inf();
#pragma unroll
    for (int j = 0; j < E64_PER; j++) {
        double *row = a.L + (size_t)(k0 + g * E64_PER + j) * a.n_pad + f0 + f;
        row[0] = lp0[j];
        row[64] = lp1[j];
        if (lp0[j] >= E64_MINLOG) pm0 = fmax(pm0, lp0[j]);
        if (lp1[j] >= E64_MINLOG) pm1 = fmax(pm1, lp1[j]);
    }
    s_pm[g * E64_DFR + f] = pm0;
    s_pm[g * E64_DFR + f + 64] = pm1;
    __syncthreads();
    double m[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int ff = f + 64 * q;
        m[q] = fmax(fmax(s_pm[ff], s_pm[E64_DFR + ff]), fmax(s_pm[2 * E64_DFR + ff], s_pm[3 * E64_DFR + ff]));
    }
    __syncthreads();
    double ps0 = 0.0, ps1 = 0.0;
    if (m[0] >= E64_MINLOG) {
#pragma unroll
        for (int j = 0; j < E64_PER; j++) ps0 += lp0[j] >= E64_MINLOG ? exp(lp0[j] - m[0]) : 0.0;
    }
    if (m[1] >= E64_MINLOG) {
#pragma unroll
        for (int j = 0; j < E64_PER; j++) ps1 += lp1[j] >= E64_MINLOG ? exp(lp1[j] - m[1]) : 0.0;
    }
    s_pm[g * E64_DFR + f] = ps0;
    s_pm[g * E64_DFR + f + 64] = ps1;
    __syncthreads();
    if (g < 2) {                                     // (wave 0: frames f, wave 1: frames f + 64)
        const int ff = f + 64 * g;
        a.mb[(size_t)kb * a.n_pad + f0 + ff] = m[g];
        a.sb[(size_t)kb * a.n_pad + f0 + ff] = ((s_pm[ff] + s_pm[E64_DFR + ff]) + s_pm[2 * E64_DFR + ff]) + s_pm[3 * E64_DFR + ff];
    }
}

// A frame's total from its blocks' (maximum, sum) pairs -- four groups of threads take every fourth block each, their maxima and then
// their sums meet in LDS in group order (one thread per frame walking up to 32 blocks' exponentials in a row: 20 us per pass at 2048
// mixtures); the chunk's sum of totals (safe_log: ln 1e-15 for a frame without a surviving term) and its flag.  64 frames per workgroup.
__global__ __launch_bounds__(256)
void e64_lse_kernel(const E64Args a) {
    __shared__ double s_part[4][E64_FR];
    const int f = threadIdx.x & 63, g = threadIdx.x >> 6, chunk = blockIdx.x, F = chunk * E64_FR + f;
    const bool valid = F < a.n;
    double pm = -__builtin_inf();
    for (int b = g; b < a.n_kb; b += 4) pm = fmax(pm, a.mb[(size_t)b * a.n_pad + F]);
    s_part[g][f] = pm;
    __syncthreads();
    const double m = fmax(fmax(s_part[0][f], s_part[1][f]), fmax(s_part[2][f], s_part[3][f]));
    const bool live = m >= E64_MINLOG;
    __syncthreads();
    double ps = 0.0;
    if (live)
        for (int b = g; b < a.n_kb; b += 4) {
            const double bm = a.mb[(size_t)b * a.n_pad + F];
            if (bm >= E64_MINLOG) ps += a.sb[(size_t)b * a.n_pad + F] * exp(bm - m);
        }
    s_part[g][f] = ps;
    __syncthreads();
    if (g != 0) return;
    const double s = ((s_part[0][f] + s_part[1][f]) + s_part[2][f]) + s_part[3][f];
    const double ll = live ? m + log(s) : 0.0;
    a.llf[F] = valid && live ? ll : __builtin_inf();
    const int bad = valid && ((live && m < E64_BAND) || !(s == s));
    const double t = wave_sum_f64(valid ? (live ? ll : E64_LN_1E_15) : 0.0);
    const unsigned long long any = __builtin_amdgcn_ballot_w64(bad);
    if (f == 0) {
        a.llpart[2 * chunk] = t;
        a.llpart[2 * chunk + 1] = any ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(E64_STHREADS)
void e64_stats_kernel(const E64Args a) {
    extern __shared__ __attribute__((aligned(16))) double e64_lds[];
    const int D = a.dim, XS = D + 1, REC = 2 * D + 1;
    double *s_g = e64_lds;                           // [64][64]   responsibilities of the block's mixtures
    double *s_mu = s_g + E64_KB * E64_FR;            // [64][D]
    float *s_x = reinterpret_cast<float *>(s_mu + E64_KB * D);     // [64][D + 1]
    __shared__ int s_bad;
    const int tid = threadIdx.x, f = tid & 63, g = tid >> 6;
    const int chunk = blockIdx.x, kb = blockIdx.y, f0 = chunk * E64_FR, k0 = kb * E64_KB;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < E64_KB * D; i += E64_STHREADS) s_mu[i] = k0 + i / D < a.K ? a.mu[(size_t)k0 * D + i] : 0.0;
    for (int i = tid; i < E64_FR * D; i += E64_STHREADS) {
        const int fr = i / D, d = i - fr * D;
        s_x[fr * XS + d] = f0 + fr < a.n ? a.X[(size_t)(f0 + fr) * D + d] : 0.f;
    }
    const double ll = a.llf[f0 + f];                 // (+inf: no responsibility -- exp(lp - inf) = 0)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < E64_SPER; j++) {
        const int kl = g * E64_SPER + j;
        const double lp = a.L[(size_t)(k0 + kl) * a.n_pad + f0 + f];
        if (f0 + f < a.n && !(lp == lp) && k0 + kl < a.K) atomicOr(&s_bad, 1);
        s_g[kl * E64_FR + f] = lp >= E64_MINLOG ? exp(lp - ll) : 0.0;
    }
    __syncthreads();
    // sums of the block's mixtures over the chunk's frames: a (mixture, dimension) pair per thread and step, four running sums per
    // moment (frames i = q mod 4) added up in a fixed order
    const int R = E64_KB * (D + 1);
    for (int role = tid; role < R; role += E64_STHREADS) {
        const int kl = role / (D + 1), d = role - kl * (D + 1);
        if (k0 + kl >= a.K) continue;
        const double *gam = s_g + kl * E64_FR;
        double p1[4] = {0.0, 0.0, 0.0, 0.0}, p2[4] = {0.0, 0.0, 0.0, 0.0};
        double *dst = a.partial + ((size_t)chunk * a.K + k0 + kl) * REC;
        if (d < D) {
            const double mu = s_mu[kl * D + d];
            for (int i = 0; i < E64_FR; i += 4) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const double dv = (double)s_x[(i + q) * XS + d] - mu;
                    const double gd = gam[i + q] * dv;
                    p1[q] += gd;
                    p2[q] = fma(gd, dv, p2[q]);
                }
            }
            dst[d] = (p1[0] + p1[1]) + (p1[2] + p1[3]);
            dst[D + d] = (p2[0] + p2[1]) + (p2[2] + p2[3]);
        } else {
            for (int i = 0; i < E64_FR; i += 4) {
#pragma unroll
                for (int q = 0; q < 4; q++) p1[q] += gam[i + q];
            }
            dst[2 * D] = (p1[0] + p1[1]) + (p1[2] + p1[3]);
        }
    }
    if (tid == 0 && s_bad) a.llpart[2 * chunk + 1] = 1.0;       // (a NaN density: any block of the chunk raises the chunk's flag)
}

// total log-likelihood of the pass (the chunks' sums: two per lane, then the wave's fixed-order sum) and the flag, which ACCUMULATES
// over the passes of a fit (the host does not wait for every pass)
__global__ __launch_bounds__(64)
void e64_head_kernel(const E64Args a) {
    const int lane = threadIdx.x;
    double v = 0.0, bad = 0.0;
    for (int c = lane; c < a.n_chunks; c += 64) {
        v += a.llpart[2 * c];
        bad += a.llpart[2 * c + 1];
    }
    const double ll = wave_sum_f64(v), b = wave_sum_f64(bad);
    if (lane == 0) {
        a.head[0] = ll;
        a.head[1] += b;
    }
}

__global__ __launch_bounds__(256)
void e64_mstep_kernel(const E64Args a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int D = a.dim, REC = 2 * D + 1;
    if (i >= a.K * D) return;
    const int k = i / D, d = i - k * D;
    double nk = 0.0, sd = 0.0, sdd = 0.0;
    for (int c = 0; c < a.n_chunks; c++) {                       // (the chunks in order)
        const double *p = a.partial + ((size_t)c * a.K + k) * REC;
        nk += p[2 * D];
        sd += p[d];
        sdd += p[D + d];
    }
    if (nk == 0.0) nk = 1e-6;                                    // min_n_k, gmm.cc:502-509
    const double mu_old = a.mu[i];
    const double shift = sd / nk;                                // E_k[x] - mu_old
    if (a.map) {                                                 // update_means, gmmubm.cc:53-74
        const double alpha = nk / (nk + a.relevance);
        a.mu[i] = alpha * (mu_old + shift) + (1 - alpha) * a.ubm_mu[i];
    } else {                                                     // gmm.cc:396-437
        a.mu[i] = mu_old + shift;
        double var = sdd / nk - shift * shift;                   // sum g (x - mu_new)^2 = sdd - N shift^2
        if (var < 0) var = 0;
        const double sg = fmax(a.min_sigma, sqrt(var));
        a.sg[i] = sg;
        a.h[i] = 0.5 / (sg * sg);
    }
}

// EM only: update_weights (gmm.cc:388-394) -- the quotients side by side, their sum in mixture order -- and the constants.  One
// workgroup (after e64_mstep_kernel: the constants take the new sigmas).
__global__ __launch_bounds__(1024)
void e64_weights_kernel(const E64Args a) {
    __shared__ double s_sum;
    const int D = a.dim, REC = 2 * D + 1;
    for (int k = threadIdx.x; k < a.K; k += 1024) {
        double nk = 0.0;
        for (int c = 0; c < a.n_chunks; c++) nk += a.partial[((size_t)c * a.K + k) * REC + 2 * D];
        if (nk == 0.0) nk = 1e-6;
        a.w[k] = nk / (double)a.n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double wsum = 0.0;
        for (int k = 0; k < a.K; k++) wsum += a.w[k];
        s_sum = wsum;
    }
    __syncthreads();
    const double wsum = s_sum;
    for (int k = threadIdx.x; k < a.K; k += 1024) {
        const double w = a.w[k] / wsum;
        a.w[k] = w;
        double c = w > 0.0 ? log(w) : -__builtin_inf();
        for (int d = 0; d < D; d++) c -= log(E64_SQRT_2_PI * a.sg[k * D + d]);
        a.c[k] = c;
    }
}

struct E64Workspace {
    DevBuf<double> model, L, ms, partial, small;
    PinnedBuf<double> h_head, h_model;
};

}  // namespace

bool em_f64_eligible(int K, int dim, long n, const Parameter &param) {
    return K >= 1 && dim >= 1 && dim <= E64_MAX_D && n >= 1 && n <= E64_MAX_FRAMES && param.nr_iteration >= 1 && param.verbosity < 2 &&
           (long)((K + E64_KB - 1) / E64_KB * E64_KB) * ((n + E64_DFR - 1) / E64_DFR * E64_DFR) <= E64_MAX_CELLS;
}

// The fit of `gmm` (its parameters are the start) on the n resident frames dX, an iteration = four launches.  true: done -- gmm holds
// the result, *iterations the count train_em returns; false: a frame this path leaves to the other one (gmm untouched).
bool train_em_f64(GMM &gmm, const GMM *ubm, const float *dX, long n, int dim, const Parameter &param, double relevance, int *iterations) {
    const int K = gmm.nr_mixtures, KD = K * dim, REC = 2 * dim + 1;
    auto &w = per_device<E64Workspace>();
    E64Args a;
    a.X = dX;
    a.n = (int)n;
    a.dim = dim;
    a.K = K;
    a.n_pad = (int)((n + E64_DFR - 1) / E64_DFR) * E64_DFR;
    a.n_chunks = a.n_pad / E64_FR;                 // (the last 64-frame chunk may be all padding: zeros in every sum)
    a.n_kb = (K + E64_KB - 1) / E64_KB;
    a.map = ubm ? 1 : 0;
    a.min_sigma = std::sqrt(param.min_covar);
    a.relevance = relevance;
    // model block: w [K], mu [KD], sg [KD], h [KD], c [K], ubm mu [KD]
    std::vector<double> init((size_t)K + 2 * (size_t)KD);
    for (int k = 0; k < K; k++) init[k] = gmm.weights[k];
    for (int i = 0; i < KD; i++) {
        init[K + i] = gmm.mean[i];
        init[K + KD + i] = gmm.sigma[i];
    }
    w.model.ensure((size_t)2 * K + 4 * (size_t)KD);
    a.w = w.model.p;
    a.mu = a.w + K;
    a.sg = a.mu + KD;
    a.h = a.sg + KD;
    a.c = a.h + KD;
    double *ubm_mu = a.c + K;
    a.ubm_mu = ubm_mu;
    SR_HIP(hipMemcpyAsync(a.w, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice, ctx().stream));
    if (ubm) SR_HIP(hipMemcpyAsync(ubm_mu, ubm->mean.data(), (size_t)KD * sizeof(double), hipMemcpyHostToDevice, ctx().stream));
    w.L.ensure((size_t)a.n_kb * E64_KB * a.n_pad);
    w.ms.ensure((size_t)2 * a.n_kb * a.n_pad);
    w.partial.ensure((size_t)a.n_chunks * K * REC);
    w.small.ensure((size_t)2 * a.n_chunks + 2 + a.n_pad);
    w.h_head.ensure(2);
    a.L = w.L.p;
    a.mb = w.ms.p;
    a.sb = w.ms.p + (size_t)a.n_kb * a.n_pad;
    a.partial = w.partial.p;
    a.llpart = w.small.p;
    a.head = w.small.p + 2 * a.n_chunks;
    a.llf = a.head + 2;
    SR_HIP(hipMemsetAsync(a.head, 0, 2 * sizeof(double), ctx().stream));
    hipStream_t st = ctx().stream;
    const unsigned g_kd = (unsigned)((KD + 255) / 256);
    hipLaunchKernelGGL(e64_derive_kernel, dim3(g_kd), dim3(256), 0, st, a, 3);
    const size_t lds_a = (size_t)(2 * E64_KB * dim + E64_KB + 4 * E64_DFR) * sizeof(double) + (size_t)E64_DFR * (dim + 1) * sizeof(float);
    const size_t lds_b = (size_t)(E64_KB * E64_FR + E64_KB * dim) * sizeof(double) + (size_t)E64_FR * (dim + 1) * sizeof(float);
    if (lds_a > 64 * 1024)
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&e64_density_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    if (lds_b > 64 * 1024)
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&e64_stats_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    const dim3 grid_a((unsigned)(a.n_pad / E64_DFR), (unsigned)a.n_kb), grid_b((unsigned)a.n_chunks, (unsigned)a.n_kb);
    const int nit = param.nr_iteration;
    double last_ll = -std::numeric_limits<double>::max();
    int done = nit;
    for (int it = 0;; it++) {
        const bool ll_only = it == nit;                // the total after the LAST iteration, when that one is an odd one (gmm.cc:622)
        if (ll_only && ((nit - 1) & 1) == 0) break;
        hipLaunchKernelGGL(e64_density_kernel, grid_a, dim3(E64_THREADS), lds_a, st, a);
        hipLaunchKernelGGL(e64_lse_kernel, dim3((unsigned)a.n_chunks), dim3(256), 0, st, a);
        hipLaunchKernelGGL(e64_stats_kernel, grid_b, dim3(E64_STHREADS), lds_b, st, a);
        hipLaunchKernelGGL(e64_head_kernel, dim3(1), dim3(64), 0, st, a);
        SR_HIP(hipGetLastError());
        // the host waits only where the stop rule wants the total (every second pass) and at the end; the flag accumulates
        const bool wants_ll = ll_only || (it >= 1 && ((it - 1) & 1));
        if (wants_ll) {
            SR_HIP(hipMemcpyAsync(w.h_head.p, a.head, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            sync_stream();
            if (w.h_head.p[1] > 0.0) return false;
        }
        // the total under the model as iteration it - 1 left it: the reference takes it after odd iterations (gmm.cc:622-650)
        if (it >= 1 && ((it - 1) & 1)) {
            const double ll = w.h_head.p[0];
            if (param.verbosity >= 1) printf("iter %d: ll %lf\n", it - 1, ll);
            const double ll_diff = ll - last_ll;
            if (std::fabs(ll_diff) / std::fabs(ll) < param.threshold && ll_diff < param.threshold) {
                done = it;
                break;
            }
            last_ll = ll;
        }
        if (ll_only) break;
        hipLaunchKernelGGL(e64_mstep_kernel, dim3(g_kd), dim3(256), 0, st, a);
        if (!ubm) hipLaunchKernelGGL(e64_weights_kernel, dim3(1), dim3(1024), 0, st, a);
    }
    w.h_model.ensure((size_t)K + 2 * (size_t)KD);
    SR_HIP(hipMemcpyAsync(w.h_model.p, a.w, ((size_t)K + 2 * (size_t)KD) * sizeof(double), hipMemcpyDeviceToHost, st));
    SR_HIP(hipMemcpyAsync(w.h_head.p, a.head, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    sync_stream();
    if (w.h_head.p[1] > 0.0) return false;
    for (int k = 0; k < K; k++) gmm.weights[k] = w.h_model.p[k];
    for (int i = 0; i < KD; i++) {
        gmm.mean[i] = w.h_model.p[K + i];
        gmm.sigma[i] = w.h_model.p[K + KD + i];
    }
    gmm.drop_single();
    *iterations = done;
    return true;
}

}  // namespace sr
