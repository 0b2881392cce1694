// lse.hpp -- the online log2-sum-exp shared by the scoring kernels, with the reference's underflow
// semantics as an option.
//
// Reference (gmm.cc:237-244 + safe_log :34-38, compiled -ffast-math => FTZ): the mixture sum runs in
// the LINEAR domain, so a term w_k p_k(x) below DBL_MIN = exp(-708.396) flushes to exactly 0; the sum
// of the surviving terms goes through log(), and an all-zero sum returns ln(1e-15).  Pinned by
// tests/golden/make_clamp_golden.py (reference DSO): clamp <=> largest term < DBL_MIN, and just above
// that boundary the sub-DBL_MIN terms are missing from the sum (up to ln K nats).
//
// Log-domain restatement used here: terms below MINLOG2 (= ln DBL_MIN in log2 units) contribute 0;
// no surviving term -> ln(1e-15).  Dropping a term costs a compare + select per element, so the
// kernels do it only while some lane's running maximum is within NEAR log2 units above the
// boundary (a wave-uniform branch never taken on real data: it needs a frame ~37 sigma from every
// mixture); a term more than NEAR below the running maximum is < 2^-40 of the sum either way.
//
// PARTIAL products (round 3).  A mixture's density is a product of D per-dimension factors
// (gmm.cc:192-195) and every intermediate of it flushes too: a partial product below DBL_MIN is 0 for
// good even when later factors > 1 (sigma < 0.399) would have lifted the full product back, and a single
// dimension whose exponent reaches fastexp.cc's floor (-708.396, :104-105,128-131) zeroes its mixture.
// Those decisions depend on the ORDER of the factors; they are taken by a separate, rare path
// (gmm_flush.hip: the reference's linear-domain arithmetic restated in float64, explicit flushes) on
// exactly the frames that can be affected: a killed term has a full product below exp(-708.396 + R),
// R = sum_d max(0, -ln sigma_d), so a (frame, model) pair whose log-likelihood is at least
// `band_hi` = -708.396 + max R + ln K + 17.5 cannot change by more than 3e-8.  At its per-model close
// every engine compares the frame's value with band_hi (one v_cmp per frame and model; the clamped value
// ln 1e-15 = -34.5 is above any band_hi, and below the boundary the full-product rule and the reference
// agree: everything is 0) and, when a frame of its tile is below, writes FLUSH_POISON instead of the
// tile's partial sum for that model: no list, no atomics, no registers in the hot kernels.
// gmm_finalize_kernel, which reads every partial anyway, leaves such a (tile, model) out of the
// utterance's sum and notes it; gmm_flush.hip re-evaluates the tile's frames for that model.
#pragma once

#include <hip/hip_runtime.h>

namespace sr {

constexpr float LSE_LN2 = 0.69314718055994530942f;
constexpr float LSE_MINLOG2 = -708.396418532264f * 1.4426950408889634f;   // log2(DBL_MIN)
constexpr float LSE_LN_1E_15 = -34.538776394910684f;                      // safe_log floor, gmm.cc:34-38
constexpr float LSE_NEAR = 40.0f;
constexpr float LSE_NEG_BIG = -1.0e30f;
// written in place of a (tile, model) partial sum when a frame of the tile sits in the band where the reference's
// flushes of partial products can decide (never a legitimate sum: log-likelihoods are finite or -inf)
#define SR_FLUSH_POISON (__builtin_inf())
__device__ __forceinline__ bool flush_poisoned(double v) { return v == SR_FLUSH_POISON; }

typedef float lse_f32x16 __attribute__((ext_vector_type(16)));

// threshold below which a lane's running maximum sends its wave down the per-term path
__device__ __forceinline__ float lse_near_threshold(int clamp) {
    return clamp ? LSE_MINLOG2 + LSE_NEAR : -3.0e38f;
}

// (m, ssum) <- (m, ssum) (+) the 16 log2-domain terms of `acc`
__device__ __forceinline__ void lse_update16(const lse_f32x16 &acc, float &m, float &ssum, float near_thr) {
    float mx = fmaxf(acc[0], acc[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);   // v_max3_f32
    const float mn = fmaxf(m, mx);
    float e = 0.0f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mn < near_thr) != 0, 0)) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float t = __builtin_amdgcn_exp2f(acc[r] - mn);
            e += acc[r] >= LSE_MINLOG2 ? t : 0.0f;
        }
        const float keep = m >= LSE_MINLOG2 ? ssum : 0.0f;     // everything summed so far was below DBL_MIN
        ssum = fmaf(keep, __builtin_amdgcn_exp2f(m - mn), e);
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) e += __builtin_amdgcn_exp2f(acc[r] - mn);
        ssum = fmaf(ssum, __builtin_amdgcn_exp2f(m - mn), e);
    }
    m = mn;
}

// natural-log likelihood of two merged partial states (the two half-waves' 16 mixture rows each)
__device__ __forceinline__ float lse_close2(float m, float ssum, float om, float os, int clamp) {
    const float mn = fmaxf(m, om);
    float a = ssum * __builtin_amdgcn_exp2f(m - mn);
    float b = os * __builtin_amdgcn_exp2f(om - mn);
    if (clamp) {
        a = m >= LSE_MINLOG2 ? a : 0.0f;
        b = om >= LSE_MINLOG2 ? b : 0.0f;
    }
    float ll = LSE_LN2 * (mn + log2f(a + b));
    // clamp 2 (the two halves of a hybrid set, merged afterwards): "every term underflowed" is reported out of band, as
    // -inf -- a genuine value that happens to round to ln 1e-15 must not be mistaken for it
    if (clamp && mn < LSE_MINLOG2) ll = clamp == 2 ? -__builtin_inff() : LSE_LN_1E_15;
    return ll;
}

__device__ __forceinline__ float lse_close1(float m, float ssum, int clamp) {
    float ll = LSE_LN2 * (m + log2f(ssum));
    if (clamp && m < LSE_MINLOG2) ll = clamp == 2 ? -__builtin_inff() : LSE_LN_1E_15;
    return ll;
}

}  // namespace sr
