// split_schemes.hpp -- the 16-bit operand splits of the generic split engines (gmm_score_split.hip, gmm_score_splitp.hip):
// how an fp32 operand becomes 16-bit parts, which part products are formed and in what order.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

struct bf16x3 {
    static constexpr int PARTS = 3, NPROD = 6;
    typedef bf16x8 frag;
    // small products first, and consecutive MFMAs share one operand
    static constexpr int AI[6] = {2, 1, 1, 0, 0, 0};
    static constexpr int BI[6] = {0, 0, 1, 1, 2, 0};
    static constexpr bool SCALED = false;
    __device__ static __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // 16-bit patterns of the parts of v (round-to-nearest-even at each step: the split is exact)
    __device__ static __forceinline__ void split(float v, uint32_t (&p)[3]) {
        auto rne = [](float f) {
            const uint32_t u = __float_as_uint(f);
            return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        };
        const uint32_t p0 = rne(v);
        const float r1 = v - __uint_as_float(p0);
        const uint32_t p1 = rne(r1);
        const float r2 = r1 - __uint_as_float(p1);
        p[0] = p0 >> 16;
        p[1] = p1 >> 16;
        p[2] = rne(r2) >> 16;
    }
    // the parts of a pair of values, packed (v.x in the low half of every word): v_cvt_pk_bf16_f32 (gfx950; RNE, as `split`)
    // and a shift / a mask to widen the parts again -- 9 instructions for 6 parts
    __device__ static __forceinline__ void split2(f32x2v v, uint32_t (&p)[3]) {
#pragma clang fp contract(off)
        typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
        const bf16x2v h = __builtin_convertvector(v, bf16x2v);
        const f32x2v r1 = v - __builtin_convertvector(h, f32x2v);
        const bf16x2v m = __builtin_convertvector(r1, bf16x2v);
        const f32x2v r2 = r1 - __builtin_convertvector(m, f32x2v);
        const bf16x2v l = __builtin_convertvector(r2, bf16x2v);
        p[0] = __builtin_bit_cast(uint32_t, h);
        p[1] = __builtin_bit_cast(uint32_t, m);
        p[2] = __builtin_bit_cast(uint32_t, l);
    }
};

struct f16x2 {
    static constexpr int PARTS = 2, NPROD = 3;
    typedef f16x8 frag;
    static constexpr int AI[3] = {1, 0, 0};
    static constexpr int BI[3] = {0, 1, 0};
    static constexpr bool SCALED = true;
    __device__ static __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void split(float v, uint32_t (&p)[2]) {
        const _Float16 h = (_Float16)v;                    // v_cvt_f16_f32, RNE, gradual underflow
        const _Float16 l = (_Float16)(v - (float)h);
        p[0] = (uint32_t)__builtin_bit_cast(unsigned short, h);
        p[1] = (uint32_t)__builtin_bit_cast(unsigned short, l);
    }
    // a pair at a time: v_cvt_pk_f16_f32 (gfx950; RNE like v_cvt_f16_f32) and one v_pk_add_f32 -- 5 instructions for 4 parts
    __device__ static __forceinline__ void split2(f32x2v v, uint32_t (&p)[2]) {
#pragma clang fp contract(off)
        const f16x2v h = __builtin_convertvector(v, f16x2v);
        const f16x2v l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2v), f16x2v);
        p[0] = __builtin_bit_cast(uint32_t, h);
        p[1] = __builtin_bit_cast(uint32_t, l);
    }
};

// The fp16 image with only the product of the high parts: a third of f16x2's MFMAs for a value good to a few parts in a
// thousand of every term -- what the shared-sigma engine's reference-offset pre-pass needs (gmm_score_h2_shared.hip: any
// offset within ~60 nats of a model's log-likelihood does; the result does not depend on it beyond the rounding of a sum).
struct f16x1 : f16x2 {
    static constexpr int NPROD = 1;
    static constexpr int AI[1] = {0};
    static constexpr int BI[1] = {0};
};

}  // namespace sr
