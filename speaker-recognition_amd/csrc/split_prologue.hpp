// split_prologue.hpp -- the frame side of the generic split engines (gmm_score_split.hip, gmm_score_splitp.hip): one lane's
// share of its frame's feature row -> the resident B fragments of the wave's 32-frame column tile.
//
// Contraction step ks covers features 8 ks .. 8 ks + 7, two slots each (x'^2 against A2, x' against A1).  Lane (col, hh) of the
// wave holds slots 8 hh .. 8 hh + 7 of every step for frame `col`: BOTH powers of features f0..f3 = 8 ks + 4 hh + 0..3, in the
// order f0^2 f1^2 f0 f1 f2^2 f3^2 f2 f3 -- pairs as the packed instructions produce them (gmm_model.cpp packs the mixture side
// to match).  The very last slot (feature 8 KS - 1 >= dim, first power) carries the constant 1 that picks up C_k.  breg[ks][part] = the 16-bit parts of this lane's 8 slots of step ks.
//
// Round 4.  This prologue is what the 256 x 39 point mostly WAS: 0.17 of its 0.32 ms with nothing else in the kernel, and --
// unlike the matrix phases -- not a matter of the power cap (zero operands do not change it).  Then: every lane fetched the
// whole row (the half-waves kept x'^2 or x' of all 8 KS features), centre and scale came as one scalar load + wait each
// (index clamped to dim - 1 at run time: nothing merged), every feature load had its own 64-bit address computation for the
// same reason, ~12 vector instructions per slot.  Now a lane owns 4 features per step and keeps both powers (half the loads,
// half the arithmetic, no selects), the packer pads `center` / `scale` to 8 KS entries so that rows and tables arrive as 16-byte
// loads, only the last step clamps its indices (dim >= 8 (KS - 1) by the definition of KS), and a PAIR of slots goes through
// v_pk_add_f32 / v_pk_mul_f32 and v_cvt_pk_f16_f32 (gfx950): ~4 vector instructions per slot.
#pragma once

#include "split_schemes.hpp"

namespace sr {

// The arithmetic on values already in registers: xs[ks][i] = feature 8 ks + 4 hh + i of this lane's frame, cs / ss the table
// entries of the same slots (ss: SC::SCALED only).  zmax: running maximum of |x'| (SC::SCALED: what the caller compares with
// the fp16 range).
template <typename SC, int KS>
__device__ __forceinline__ void split_fragments_from_values(const float (&xs)[KS][4], const float (&cs)[KS][4], const float (&ss)[KS][4],
                                                            int dim, int hh, typename SC::frag (&breg)[KS][SC::PARTS], float &zmax) {
#pragma clang fp contract(off)        // (x'^2 - hi(x'^2) must not become an fma of x' with itself: the parts are those of the fp32 square)
    constexpr int P = SC::PARTS;
    typedef typename SC::frag frag;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        uint32_t w[P][4];
#pragma unroll
        for (int ip = 0; ip < 2; ip++) {
            f32x2v xc = f32x2v{xs[ks][2 * ip], xs[ks][2 * ip + 1]} - f32x2v{cs[ks][2 * ip], cs[ks][2 * ip + 1]};
            if constexpr (SC::SCALED) {
                xc *= f32x2v{ss[ks][2 * ip], ss[ks][2 * ip + 1]};
                // (a padded slot has x' = 0 here; NaN features compare false and fall through to the arithmetic, which propagates them)
                zmax = fmaxf(zmax, fmaxf(fabsf(xc.x), fabsf(xc.y)));
                xc.x = fminf(fmaxf(xc.x, -255.0f), 255.0f);        // x'^2 stays below fp16's 65504
                xc.y = fminf(fmaxf(xc.y, -255.0f), 255.0f);
            }
            f32x2v sq = xc * xc;
            if (ks == KS - 1) {                                       // (earlier steps: every feature < 8 (KS - 1) <= dim)
                const int d = 8 * ks + 4 * hh + 2 * ip;
                sq.x = d < dim ? sq.x : 0.0f;
                xc.x = d < dim ? xc.x : 0.0f;
                sq.y = d + 1 < dim ? sq.y : 0.0f;
                xc.y = d + 1 < dim ? xc.y : 0.0f;
                if (ip == 1) xc.y = hh ? 1.0f : xc.y;
            }
            uint32_t p[P];
            SC::split2(sq, p);
#pragma unroll
            for (int pi = 0; pi < P; pi++) w[pi][2 * ip] = p[pi];
            SC::split2(xc, p);
#pragma unroll
            for (int pi = 0; pi < P; pi++) w[pi][2 * ip + 1] = p[pi];
        }
#pragma unroll
        for (int pi = 0; pi < P; pi++) breg[ks][pi] = __builtin_bit_cast(frag, make_uint4(w[pi][0], w[pi][1], w[pi][2], w[pi][3]));
    }
}

// `row` = this lane's frame in global memory; `center`, `scale` = the packer's tables, 8 KS entries (scale: SC::SCALED only).
template <typename SC, int KS>
__device__ __forceinline__ void split_frame_fragments(const float *__restrict__ row, int dim, int hh,
                                                      const float *__restrict__ center, const float *__restrict__ scale,
                                                      typename SC::frag (&breg)[KS][SC::PARTS], float &zmax) {
    float xs[KS][4], cs[KS][4], ss[KS][4];
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int d = 8 * ks + 4 * hh + i;
            xs[ks][i] = row[(ks < KS - 1 || d < dim) ? d : dim - 1];
            cs[ks][i] = center[d];
            ss[ks][i] = 1.0f;
            if constexpr (SC::SCALED) ss[ks][i] = scale[d];
        }
    // keep the loads unconditional and batched: without this the compiler sinks them into the branches of the last step
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("" : "+v"(xs[ks][i]));
    split_fragments_from_values<SC, KS>(xs, cs, ss, dim, hh, breg, zmax);
}

}  // namespace sr
