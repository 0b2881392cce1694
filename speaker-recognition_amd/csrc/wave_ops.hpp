// wave_ops.hpp -- wave64 cross-lane helpers that stay on the vector ALU (DPP / readlane /
// permlane32_swap).  __shfl_xor lowers to ds_bpermute (an LDS round trip, ~100+ cycles of
// latency per dependent step); these do not.
#pragma once

#include <hip/hip_runtime.h>

namespace sr {

// Orders LDS traffic between the lanes of ONE wave (no instruction emitted: a wave's LDS operations
// execute in order; this only stops the compiler from moving them across).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Barrier that publishes LDS-DMA (global_load_lds) data to the other waves of the workgroup: this
// wave's pieces must have LANDED before it arrives.  hipcc tracks LDS-DMA by alias analysis and puts
// its vmcnt wait in front of this wave's own ds_reads -- possibly behind the barrier, and (seen in
// gmm_score_h2_shared.hip) not at all for a DMA issued in the previous iteration of a loop -- so the
// wait is spelled out.
__device__ __forceinline__ void dma_publish_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Sum of `v` over the 64 lanes of the wave; the result is valid in every lane (it comes back
// through SGPRs).  float64, fixed order: row-wise prefix sums by DPP row_shr 1/2/4/8, then the
// four row totals.
__device__ __forceinline__ double wave_sum_f64(double v) {
    union { double d; int i[2]; } a, b;
#define SR_DPP_STEP(CTRL)                                                        \
    a.d = v;                                                                     \
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xf, 0xf, true);       \
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xf, 0xf, true);       \
    v += b.d;
    SR_DPP_STEP(0x111)   // row_shr:1, zeros shifted in
    SR_DPP_STEP(0x112)   // row_shr:2
    SR_DPP_STEP(0x114)   // row_shr:4
    SR_DPP_STEP(0x118)   // row_shr:8  -> lane 15 of each 16-lane row holds the row total
#undef SR_DPP_STEP
    a.d = v;
    double tot = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        union { double d; int i[2]; } t;
        t.i[0] = __builtin_amdgcn_readlane(a.i[0], 16 * r + 15);
        t.i[1] = __builtin_amdgcn_readlane(a.i[1], 16 * r + 15);
        tot += t.d;
    }
    return tot;
}

// Value of `x` held by lane (l ^ 32): one v_permlane32_swap.
__device__ __forceinline__ float other_half(float x) {
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0]: lanes 32-63 now hold the lower half's values; r[1]: lanes 0-31 hold the upper half's
    const unsigned o = (threadIdx.x & 32) ? r[0] : r[1];
    return __uint_as_float(o);
}

}  // namespace sr
