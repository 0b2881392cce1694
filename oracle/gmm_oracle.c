/*
 * oracle/gmm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, float64) of the reference's diagonal-GMM scoring
 * arithmetic.  It exists so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can CHECK the HIP path; nothing in the product
 * (speaker-recognition_amd/) may include, link or call it.
 *
 * Parity pinning: this restatement is validated against the reference's own
 * compiled C++ (oracle/_ref/pygmm_ref.so, built from /root/reference/src/gmm
 * by oracle/Makefile) on the reference's shipped model fixtures; the resulting
 * known-answer vectors are committed under tests/golden/ (see
 * tests/golden/make_golden.py) and re-checked by tests/test_oracle_golden.py.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* src/gmm/src/gmm.cc:22 */
static const double SQRT_2_PI = 2.5066282746310002;

/* src/gmm/src/gmm.cc:34-38 -- log with a floor of 1e-15 for non-positive input */
static double safe_log(double x)
{
    if (x <= 0)
        x = 1e-15;
    return log(x);
}

/* The reference DSO is built with -ffast-math, whose start-up object switches
 * the SSE unit to flush-to-zero / denormals-are-zero (SURVEY.md section 7,
 * hard part 3).  `ftz` reproduces that: results below DBL_MIN become 0. */
static double flush(double v, int ftz)
{
    if (ftz && fabs(v) < DBL_MIN)
        return 0.0;
    return v;
}

/*
 * Scalar restatement of remez5_0_log2_sse, src/gmm/src/fastexp.cc:99-212:
 * clamp to [minlog, maxlog] (:104-105,128-131), a = x*log2e (:134-137),
 * k = trunc(a - (a<0)) (:140-149), x -= k*C1, x -= k*C2 (:152-161),
 * degree-5 polynomial in Horner form (:164-192), scale by 2^k built from the
 * exponent bits (:195-206).  Under FTZ the last product flushes: at the lower clamp
 * (x = minlog, k = -1022, polynomial = w0 < 1) the result is below DBL_MIN, i.e. exactly 0 --
 * a dimension whose exponent reaches the floor zeroes its whole mixture.
 */
static double remez5_exp(double x, int ftz)
{
    const double maxlog = 7.09782712893383996843e2;
    const double minlog = -7.08396418532264106224e2;
    const double log2e = 1.4426950408889634073599;
    const double c1 = 6.93145751953125E-1;
    const double c2 = 1.42860682030941723212E-6;
    const double w5 = 1.185268231308989403584147407056378360798378534739e-2;
    const double w4 = 3.87412011356070379615759057344100690905653320886699e-2;
    const double w3 = 0.16775408658617866431779970932853611481292418818223;
    const double w2 = 0.49981934577169208735732248650232562589934399402426;
    const double w1 = 1.00001092396453942157124178508842412412025643386873;
    const double w0 = 0.99999989311082729779536722205742989232069120354073;

    if (x > maxlog) x = maxlog;
    if (x < minlog) x = minlog;
    double a = x * log2e;
    if (a < 0) a -= 1.0;
    int32_t k = (int32_t)a; /* truncation, as _mm_cvttpd_epi32 */
    double p = (double)k;
    x -= p * c1;
    x -= p * c2;
    a = x * w5 + w4;
    a = a * x + w3;
    a = a * x + w2;
    a = a * x + w1;
    a = a * x + w0;
    uint64_t bits = ((uint64_t)(uint32_t)(k + 1023)) << 52;
    double scale;
    memcpy(&scale, &bits, sizeof scale);
    return flush(a * scale, ftz);
}

/* src/gmm/src/gmm.cc:176-202 -- Gaussian::probability_of_fast_exp:
 * buf[i] = -d*d/(2 s s) (:186-190), vector exp (:191), prod buf[i]/(sqrt(2pi) s) (:192-195).
 *
 * WHICH partial products exist -- and therefore which of them can flush to zero under the DSO's
 * FTZ arithmetic -- is decided by the compiler, not by the source: -ffast-math lets it reassociate.
 *   order 1: the source's order, one running product over the dimensions, p_i = e_i / (sqrt(2 pi) s_i).
 *   order 2: what g++ 11 -O3 -ffast-math -msse2 (the reference's flags, oracle/Makefile) emits, read
 *            off the disassembly of oracle/_ref/pygmm_ref.so: buf[i] = (x-m)(m-x) 0.5 / (s s);
 *            p_i = (e_i * 0.3989422804014327) / s_i  (a multiply by the folded 1/sqrt(2 pi), THEN
 *            the division -- the multiply can flush although the quotient would not); two running
 *            products, one over the even and one over the odd dimensions (mulpd on pairs), then
 *            even * odd, then -- dim odd -- the last dimension's factor.
 * Every operation's result flushes when `ftz`.  Pinned against the DSO by
 * tests/golden/make_flush_golden.py. */
static double gaussian_prob_fastexp(const double *x, const double *mean, const double *sigma,
                                    int dim, int ftz, int order)
{
    if (order == 1) {
        double prob = 1.0;
        for (int i = 0; i < dim; i++) {
            double s = sigma[i];
            double d = x[i] - mean[i];
            double e = remez5_exp(flush(-d * d / (2 * s * s), ftz), ftz);
            double p = flush(e / (SQRT_2_PI * s), ftz);
            prob = flush(prob * p, ftz);
        }
        return prob;
    }
    const double inv_sqrt_2pi = 0.3989422804014327;   /* 0x3FD9884533D43651, the DSO's folded constant */
    double lane[2] = {1.0, 1.0};
    double tail = 1.0;
    const int paired = dim & ~1;
    for (int i = 0; i < dim; i++) {
        double s = sigma[i];
        double t = flush((x[i] - mean[i]) * (mean[i] - x[i]), ftz);
        t = flush(t * 0.5, ftz);
        double b = flush(t / flush(s * s, ftz), ftz);
        double e = remez5_exp(b, ftz);
        double p = flush(flush(e * inv_sqrt_2pi, ftz) / s, ftz);
        if (i < paired)
            lane[i & 1] = flush(lane[i & 1] * p, ftz);
        else
            tail = p;
    }
    double prob = flush(lane[0] * lane[1], ftz);
    if (dim & 1)
        prob = flush(prob * tail, ftz);
    return prob;
}

/* src/gmm/src/gmm.cc:153-174 -- Gaussian::probability_of (libm exp, linear domain). */
static double gaussian_prob_libm(const double *x, const double *mean, const double *sigma,
                                 int dim, int ftz)
{
    double prob = 1.0;
    for (int i = 0; i < dim; i++) {
        double s = sigma[i];
        double d = x[i] - mean[i];
        double p = exp(-d * d / (2 * s * s)) / (SQRT_2_PI * s);
        prob = flush(prob * p, ftz);
    }
    return prob;
}

/* src/gmm/src/gmm.cc:78-99 -- Gaussian::log_probability_of (log domain). */
static double gaussian_logprob(const double *x, const double *mean, const double *sigma, int dim)
{
    double prob = 0;
    for (int i = 0; i < dim; i++) {
        double s = sigma[i];
        double s2 = s * s;
        double d = x[i] - mean[i];
        prob += -safe_log(SQRT_2_PI * s) - 1.0 / (2 * s2) * d * d;
    }
    return prob;
}

/*
 * Per-frame log-likelihoods for one model.
 *   mode 0: what the C ABI computes -- GMM::log_probability_of_fast_exp,
 *           src/gmm/src/gmm.cc:237-244 (sum_k w_k * p_k, then safe_log), which is what
 *           score_batch/score_all reach through threaded_log_probability_of (:533-569).
 *   mode 1: GMM::log_probability_of, src/gmm/src/gmm.cc:229-235 (libm exp, linear domain).
 *   mode 2: float64 log-sum-exp over per-mixture log densities (gmm.cc:78-99 for the
 *           density) -- the formulation the HIP kernel uses; `clamp_compat` applies the
 *           reference's underflow behaviour (terms < DBL_MIN are 0; none left -> ln(1e-15)).
 *   mode 3: mode 0's result at a fraction of its cost (see score_batch_fast).
 * Layout: weights[K], mean[K*D], sigma[K*D] (sigma = standard deviations, gmm.hh:24-46),
 * X[n*D] row-major, out[n].
 */
/*
 * mode 3 -- the reference's result (mode 0) at a fraction of its cost, for the bulk parity checks of bench.py
 * (oracle/parity_check.py: hundreds of utterances x every model on the host cores).  Per-mixture constants
 * c_k = ln w_k - sum ln(sqrt(2 pi) s) and 1/(2 s^2) are formed once; a frame's value is the float64 log-sum-exp
 * of c_k - sum_d (x - mu)^2 / (2 s^2) under the full-product underflow rule of mode 2 -- and every frame whose
 * value falls in the band in which the reference's flushes of PARTIAL products can decide (gmm.cc:192-195;
 * [ln DBL_MIN, ln DBL_MIN + max_k sum_d max(0, -ln s_kd) + ln K + 17.5), the same band the HIP engines use,
 * csrc/lse.hpp) is evaluated again by mode 0 itself.  Differs from mode 0 by remez5's polynomial error only
 * (<= 1.2e-6 absolute); tests/test_oracle_golden.py holds it to mode 0 on every golden.
 */
#define FAST_F 4            /* frames per pass over the mixture table */
#define FAST_KB 16          /* mixtures per block: two 8-wide vectors of accumulators per frame */
typedef double v8d __attribute__((vector_size(64), aligned(8), may_alias));
__attribute__((target_clones("avx512f", "avx2", "default")))
static void fast_terms(const double *xp, const double *muT, const double *hT, const double *c, int Kp, int D, double *v)
{
    /* v[f][k] = c[k] - sum_d (x_fd - mu_kd)^2 h_kd for FAST_F frames at a time, vectorised ACROSS mixtures: the tables
     * are transposed ([d][k], k padded to a multiple of FAST_KB with h = 0), a frame's coordinate is broadcast, the
     * accumulators of a block of FAST_KB mixtures stay in registers over the whole d loop */
    for (int k0 = 0; k0 < Kp; k0 += FAST_KB) {
        v8d a[FAST_F][2];
        for (int f = 0; f < FAST_F; f++) a[f][0] = a[f][1] = (v8d){0, 0, 0, 0, 0, 0, 0, 0};
        for (int d = 0; d < D; d++) {
            const v8d m0 = *(const v8d *)(muT + (long)d * Kp + k0), m1 = *(const v8d *)(muT + (long)d * Kp + k0 + 8);
            const v8d h0 = *(const v8d *)(hT + (long)d * Kp + k0), h1 = *(const v8d *)(hT + (long)d * Kp + k0 + 8);
            for (int f = 0; f < FAST_F; f++) {
                const double x = xp[f * D + d];
                const v8d xb = {x, x, x, x, x, x, x, x};
                const v8d e0 = xb - m0, e1 = xb - m1;
                a[f][0] += e0 * e0 * h0;
                a[f][1] += e1 * e1 * h1;
            }
        }
        for (int f = 0; f < FAST_F; f++) {
            *(v8d *)(v + (long)f * Kp + k0) = *(const v8d *)(c + k0) - a[f][0];
            *(v8d *)(v + (long)f * Kp + k0 + 8) = *(const v8d *)(c + k0 + 8) - a[f][1];
        }
    }
}

static void score_batch_fast(const double *weights, const double *mean, const double *sigma, int K, int D,
                             const double *X, long n, double *out, int ftz, int clamp_compat)
{
    const double minlog = -7.08396418532264106224e2;
    const int Kp = (K + FAST_KB - 1) / FAST_KB * FAST_KB;
    double *c = (double *)malloc(sizeof(double) * (size_t)Kp);
    double *hT = (double *)calloc((size_t)Kp * (size_t)D, sizeof(double));
    double *muT = (double *)calloc((size_t)Kp * (size_t)D, sizeof(double));
    double *xp = (double *)calloc((size_t)D * FAST_F, sizeof(double));
    double *vv = (double *)malloc(sizeof(double) * (size_t)Kp * FAST_F);
    double lift = 0;
    for (int k = 0; k < Kp; k++) c[k] = -INFINITY;          /* padding mixtures: never the maximum, never summed */
    for (int k = 0; k < K; k++) {
        double ck = weights[k] > 0 ? log(weights[k]) : -INFINITY, up = 0;
        for (int d = 0; d < D; d++) {
            double s = sigma[(long)k * D + d];
            ck -= log(SQRT_2_PI * s);
            hT[(long)d * Kp + k] = 1.0 / (2 * s * s);
            muT[(long)d * Kp + k] = mean[(long)k * D + d];
            if (-log(s) > 0) up += -log(s);
        }
        c[k] = ck;
        if (up > lift) lift = up;
    }
    const double band_hi = minlog + lift + log((double)K) + 17.5;
    for (long t0 = 0; t0 < n; t0 += FAST_F) {
        const int nf = (int)((n - t0) < FAST_F ? (n - t0) : FAST_F);
        for (int f = 0; f < FAST_F; f++)          /* (a short last block repeats its last frame) */
            memcpy(xp + (long)f * D, X + (t0 + (f < nf ? f : nf - 1)) * (long)D, sizeof(double) * (size_t)D);
        fast_terms(xp, muT, hT, c, Kp, D, vv);
        for (int f = 0; f < nf; f++) {
            const double *v = vv + (long)f * Kp;
            const double *x = X + (t0 + f) * (long)D;
            double m = -INFINITY;
            for (int k = 0; k < K; k++)
                if (v[k] > m) m = v[k];
            double s = 0;
            for (int k = 0; k < K; k++)      /* (a term 45 nats below the largest adds < 3e-20 of the sum: no exp for it) */
                if ((!clamp_compat || v[k] >= minlog) && v[k] > m - 45.0)
                    s += exp(v[k] - m);
            double ll = (m > -INFINITY) ? m + log(s) : -INFINITY;
            if (clamp_compat && m < minlog)
                ll = log(1e-15);
            else if (clamp_compat && ll < band_hi) {        /* the partial products decide: the reference's own arithmetic */
                double prob = 0;
                for (int k = 0; k < K; k++)
                    prob += flush(weights[k] * gaussian_prob_fastexp(x, mean + (long)k * D, sigma + (long)k * D, D, ftz, 2), ftz);
                ll = safe_log(prob);
            }
            out[t0 + f] = ll;
        }
    }
    free(c);
    free(hT);
    free(muT);
    free(xp);
    free(vv);
}

static int g_flush_order = 2;
/* 1 = source order, 2 = as compiled (see gaussian_prob_fastexp); the default is what the DSO does */
void oracle_set_flush_order(int order) { g_flush_order = order == 1 ? 1 : 2; }

void oracle_gmm_score_batch(const double *weights, const double *mean, const double *sigma,
                            int K, int D, const double *X, long n, double *out,
                            int mode, int ftz, int clamp_compat)
{
    if (mode == 3) {
        score_batch_fast(weights, mean, sigma, K, D, X, n, out, ftz, clamp_compat);
        return;
    }
    for (long t = 0; t < n; t++) {
        const double *x = X + t * (long)D;
        if (mode == 0 || mode == 1) {
            double prob = 0;
            for (int k = 0; k < K; k++) {
                double p = (mode == 0)
                    ? gaussian_prob_fastexp(x, mean + (long)k * D, sigma + (long)k * D, D, ftz, g_flush_order)
                    : gaussian_prob_libm(x, mean + (long)k * D, sigma + (long)k * D, D, ftz);
                prob += flush(weights[k] * p, ftz);
            }
            out[t] = safe_log(prob);
        } else {
            double m = -INFINITY;
            double *v = (double *)malloc(sizeof(double) * (size_t)K);
            for (int k = 0; k < K; k++) {
                /* a mixture of weight 0 adds exactly 0 to the reference's linear-domain sum (gmm.cc:237-244): -inf here,
                 * not safe_log's ln(1e-15) -- found by scripts/debug/fuzz_generic.py */
                v[k] = (weights[k] > 0 ? log(weights[k]) : -INFINITY) +
                       gaussian_logprob(x, mean + (long)k * D, sigma + (long)k * D, D);
                if (v[k] > m) m = v[k];
            }
            /* The reference's sum (gmm.cc:237-244) runs in the LINEAR domain under the FTZ
             * arithmetic its -ffast-math DSO switches on: a term w_k p_k below DBL_MIN =
             * exp(-708.396...) is exactly 0 there.  So with `clamp_compat` such terms are left out
             * of the sum, and when none survives -- the largest term below DBL_MIN -- the result is
             * safe_log's ln(1e-15), gmm.cc:34-38.  (Mode 0 additionally reproduces the flush of
             * partial products in dimension order and the per-dimension exponent floor of
             * fastexp.cc:104-131, which no log-domain formulation can.) */
            const double minlog = -7.08396418532264106224e2;
            double s = 0;
            for (int k = 0; k < K; k++)
                if ((!clamp_compat || v[k] >= minlog) && v[k] > -INFINITY)
                    s += exp(v[k] - m);
            double ll = (m > -INFINITY) ? m + log(s) : -INFINITY;
            free(v);
            if (clamp_compat && m < minlog)
                ll = log(1e-15);
            out[t] = ll;
        }
    }
}

/* score_all, src/gmm/src/pygmm.cc:98-102 + gmm.cc:562-569: sequential sum of the
 * per-frame values in frame order. */
double oracle_gmm_score_all(const double *weights, const double *mean, const double *sigma,
                            int K, int D, const double *X, long n, int mode, int ftz,
                            int clamp_compat)
{
    double *buf = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    oracle_gmm_score_batch(weights, mean, sigma, K, D, X, n, buf, mode, ftz, clamp_compat);
    double prob = 0;
    for (long t = 0; t < n; t++)
        prob += buf[t];
    free(buf);
    return prob;
}

/*
 * One EM iteration, restating GMMTrainerBaseline::iteration, src/gmm/src/gmm.cc:439-531:
 * responsibilities w_k p_k(x) with the fast exp (:455-469), per-frame normalisation with
 * MIN_PROB_SUM 1e-15 (:482-498), N_k with min_n_k 1e-6 (:502-513), weights N_k/n then
 * normalised (:388-394), means (:396-412), variances around the NEW means with the
 * sqrt(min_covar) floor (:415-437).  With map_relevance > 0 it restates the MAP variant
 * instead (src/gmm/src/gmmubm.cc:40-81): weights and sigmas untouched, means
 * alpha*E_k[x] + (1-alpha)*ubm_mean with alpha = N_k/(N_k + relevance).
 * weights/mean/sigma are updated in place; ubm_mean may be NULL when map_relevance <= 0.
 * Returns nothing; resp_scratch must hold K*n doubles.
 */
void oracle_gmm_em_iteration(double *weights, double *mean, double *sigma, int K, int D,
                             const double *X, long n, double min_covar, double map_relevance,
                             const double *ubm_mean, double *resp_scratch, int ftz)
{
    double *resp = resp_scratch; /* [k][i] as prob_of_y_given_x, gmm.hh:96 */
    for (int k = 0; k < K; k++)
        for (long i = 0; i < n; i++)
            resp[(long)k * n + i] = weights[k] *
                gaussian_prob_fastexp(X + i * (long)D, mean + (long)k * D, sigma + (long)k * D, D, ftz, g_flush_order);
    for (long i = 0; i < n; i++) {
        double sum = 0;
        for (int k = 0; k < K; k++)
            sum += resp[(long)k * n + i];
        if (!(sum > 0))
            sum = 1e-15;
        for (int k = 0; k < K; k++)
            resp[(long)k * n + i] /= sum;
    }
    double *Nk = (double *)malloc(sizeof(double) * (size_t)K);
    for (int k = 0; k < K; k++) {
        double s = 0;
        for (long i = 0; i < n; i++)
            s += resp[(long)k * n + i];
        if (s == 0)
            s = 1e-6;
        Nk[k] = s;
    }
    if (map_relevance <= 0) {
        double wsum = 0;
        for (int k = 0; k < K; k++) {
            weights[k] = Nk[k] / (double)n;
            wsum += weights[k];
        }
        for (int k = 0; k < K; k++)
            weights[k] /= wsum;
    }
    double min_sigma = sqrt(min_covar);
    for (int k = 0; k < K; k++) {
        double *mu = mean + (long)k * D;
        for (int d = 0; d < D; d++) {
            double acc = 0;
            for (long i = 0; i < n; i++)
                acc += X[i * (long)D + d] * resp[(long)k * n + i];
            if (map_relevance > 0) {
                double alpha = Nk[k] / (Nk[k] + map_relevance);
                mu[d] = acc * (1.0 / Nk[k] * alpha) + ubm_mean[(long)k * D + d] * (1 - alpha);
            } else {
                mu[d] = acc * (1.0 / Nk[k]);
            }
        }
    }
    if (map_relevance <= 0) {
        for (int k = 0; k < K; k++) {
            double *mu = mean + (long)k * D;
            double *sg = sigma + (long)k * D;
            for (int d = 0; d < D; d++) {
                double acc = 0;
                for (long i = 0; i < n; i++) {
                    double t = X[i * (long)D + d] - mu[d];
                    acc += t * t * resp[(long)k * n + i];
                }
                double s = sqrt(acc * (1.0 / Nk[k]));
                sg[d] = s > min_sigma ? s : min_sigma;
            }
        }
    }
    free(Nk);
}
