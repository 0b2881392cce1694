// gmm_score_bx3_shared.hip -- split-bf16 scoring of speaker sets that share sigma and weights:
// a UBM and the speakers MAP-adapted from it (train_model_from_ubm moves the means only,
// gmmubm.cc:40-81; BASELINE configs[2] and [3]).  Same math and arithmetic as
// gmm_score_bf16x3.hip (gmm.cc:176-202, :237-244, :533-569; every fp32 operand as three exact bf16
// parts, six part products per fp32 product, fp32 accumulate), but the contraction is cut in two:
//   log2 density_sk(x) = Q_k(x) + L_sk(x),   Q_k = sum_d A2_kd x'_d^2   (the same for every model)
//                                            L_sk = sum_d A1_skd x'_d + C_sk
// Q of a (mixture tile, frame tile) is evaluated once per block of SHARED_SB = 15 models and enters
// each model's chain as the C operand, so a model costs 6*KL MFMAs per tile instead of 6*KS:
// at D = 39, 18 + 15*18 = 288 MFMAs per 15 models against 15*30 = 450 (1.56x fewer).
// The price is the per-model log-sum-exp state of a whole block held in registers (2*15 VGPRs) and
// a frame's parts kept twice (squares, values): 2 waves/SIMD instead of 4.
//
// Stream order (gmm_model.hpp): per block, per mixture tile: [Q][L_0]...[L_14] -- 16 images, so the
// two LDS buffers alternate statically (hipcc keeps an LDS-DMA in flight only across separately
// named arrays); the walk is fully unrolled over the 15 models.
#include "lse.hpp"
#include "score.hpp"
#include "wave_ops.hpp"

#include <algorithm>

namespace sr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ uint32_t sh_rne(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

// three bf16 parts of 8 values -> one fragment per part
__device__ __forceinline__ void sh_split8(const float (&v)[8], bf16x8 (&out)[3]) {
    uint32_t w[3][4];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t p0 = sh_rne(v[j]);
        const float r1 = v[j] - __uint_as_float(p0);
        const uint32_t p1 = sh_rne(r1);
        const float r2 = r1 - __uint_as_float(p1);
        const uint32_t p2 = sh_rne(r2);
        if (j & 1) {
            w[0][j >> 1] |= p0;
            w[1][j >> 1] |= p1;
            w[2][j >> 1] |= p2;
        } else {
            w[0][j >> 1] = p0 >> 16;
            w[1][j >> 1] = p1 >> 16;
            w[2][j >> 1] = p2 >> 16;
        }
    }
#pragma unroll
    for (int p = 0; p < 3; p++) out[p] = __builtin_bit_cast(bf16x8, make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]));
}

// KN steps of six part products on `acc`; `init` is the C operand of the first one
template <int KN>
__device__ __forceinline__ void sh_chain(f32x16 &acc, const f32x16 &init, const uint4 *at, const bf16x8 (&b)[KN][3]) {
    uint4 n0 = at[0], n1 = at[64], n2 = at[128];
#pragma unroll
    for (int ks = 0; ks < KN; ks++) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, n0);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, n1);
        const bf16x8 a2 = __builtin_bit_cast(bf16x8, n2);
        if (ks + 1 < KN) {
            n0 = at[((ks + 1) * 3 + 0) * 64];
            n1 = at[((ks + 1) * 3 + 1) * 64];
            n2 = at[((ks + 1) * 3 + 2) * 64];
        }
        // small products first, consecutive MFMAs share one operand (as gmm_score_bf16x3.hip)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[ks][0], ks == 0 ? init : acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[ks][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[ks][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[ks][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[ks][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[ks][0], acc, 0, 0, 0);
    }
}

template <int KQ, int KL>
__global__ __launch_bounds__(256, 2)
void gmm_score_bx3_shared_kernel(const float *__restrict__ X, const TileDesc *__restrict__ tiles,
                                 const uint4 *__restrict__ params, const SharedBlock *__restrict__ blocks,
                                 const int *__restrict__ group_block_begin, const float *__restrict__ center,
                                 double *__restrict__ partial, float *__restrict__ frame_ll,
                                 int64_t n_frames, int dim, int n_models, int n_mix_tiles, int clamp,
                                 int n_groups, int n_tiles, float band_hi) {
    constexpr int SB = SHARED_SB;
    constexpr int Q_U4 = KQ * 3 * 64, L_U4 = KL * 3 * 64;
    constexpr int BUF_U4 = Q_U4 > L_U4 ? Q_U4 : L_U4;
    constexpr int STRIDE_U4 = Q_U4 + SB * L_U4;        // one mixture tile of one block
    __shared__ uint4 lds_a[BUF_U4];
    __shared__ uint4 lds_b[BUF_U4];
    __shared__ double close_slot[SB][4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;
    const int hh = lane >> 5;
    const int tile_lo = blockIdx.x & 7;        // XCD-aware order, as gmm_score_kernel
    const int q = blockIdx.x >> 3;
    const int g = q % n_groups;
    const int tile_id = (q / n_groups) * 8 + tile_lo;
    if (tile_id >= n_tiles) return;
    const TileDesc tile = tiles[tile_id];
    const int blk_begin = group_block_begin[g], blk_end = group_block_begin[g + 1];

    auto stage = [&](uint4 *dst, const uint4 *src, int n4) {
        for (int base = wave * 64; base < n4; base += 256)
            if (base + lane < n4)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + base + lane),
                    (__attribute__((address_space(3))) void *)(dst + base), 16, 0, 0);
    };

    // ---- resident B fragments of this lane's frame: squares (against A2) and values + the constant 1
    //      (against A1, C); slot c = 16 ks + 8 hh + j ----
    bf16x8 bq[KQ][3], bl[KL][3];
    const int local = wave * 32 + col;
    const bool valid = local < tile.count;
    const int64_t row = tile.start + (valid ? local : 0);
    {
        const float *src = X + row * dim;
        constexpr int KM = KQ > KL ? KQ : KL;
        float xs[KM][8];
#pragma unroll
        for (int ks = 0; ks < KM; ks++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int d = 16 * ks + 8 * hh + j;
                const int dc = d < dim ? d : dim - 1;
                xs[ks][j] = src[dc] - center[dc];
            }
#pragma unroll
        for (int ks = 0; ks < KM; ks++)
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("" : "+v"(xs[ks][j]));     // loads stay unconditional and batched
#pragma unroll
        for (int ks = 0; ks < KQ; ks++) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int d = 16 * ks + 8 * hh + j;
                v[j] = d < dim ? xs[ks][j] * xs[ks][j] : 0.0f;
            }
            sh_split8(v, bq[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < KL; ks++) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int d = 16 * ks + 8 * hh + j;
                v[j] = d < dim ? xs[ks][j] : (d == 16 * KL - 1 ? 1.0f : 0.0f);
            }
            sh_split8(v, bl[ks]);
        }
    }

    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float near_thr = lse_near_threshold(clamp);
    for (int blk = blk_begin; blk < blk_end; blk++) {
        const SharedBlock sb = blocks[blk];
        const uint4 *stream = params + sb.offset_u4;
        float m[SB], ssum[SB];
#pragma unroll
        for (int si = 0; si < SB; si++) {
            m[si] = NEG_BIG;
            ssum[si] = 0.0f;
        }
        __syncthreads();                      // previous block's readers are done with lds_a
        stage(lds_a, stream, Q_U4);
        dma_publish_barrier();
        for (int t = 0; t < n_mix_tiles; t++) {
            const uint4 *tsrc = stream + (size_t)t * STRIDE_U4;
            // image 0 (lds_a): the shared quadratic half of this mixture tile
            stage(lds_b, tsrc + Q_U4, L_U4);
            f32x16 qacc;
            sh_chain<KQ>(qacc, zero16, lds_a + lane, bq);
            __builtin_amdgcn_sched_barrier(0);
            dma_publish_barrier();
            // images 1..15: one model each, alternating lds_b / lds_a
#pragma unroll
            for (int si = 0; si < SB; si++) {
                const uint4 *cur = (si & 1) ? lds_a : lds_b;
                uint4 *other = (si & 1) ? lds_b : lds_a;
                if (si + 1 < SB)
                    stage(other, tsrc + Q_U4 + (size_t)(si + 1) * L_U4, L_U4);
                else if (t + 1 < n_mix_tiles)
                    stage(other, tsrc + STRIDE_U4, Q_U4);           // next tile's quadratic image -> lds_a
                f32x16 acc;
                sh_chain<KL>(acc, qacc, cur + lane, bl);
                lse_update16(acc, m[si], ssum[si], near_thr);
                // the model loop is unrolled and s_barrier orders memory, not ALU work: without pinning
                // the epilogue here the optimiser sinks all 15 of them below the last chain and keeps
                // 15 accumulators live (256 VGPRs + scratch)
                asm volatile("" : "+v"(m[si]), "+v"(ssum[si]));
                __builtin_amdgcn_sched_barrier(0);
                dma_publish_barrier();
            }
        }
        // ---- close the block's models ----
#pragma unroll
        for (int si = 0; si < SB; si++) {
            const float ll = lse_close2(m[si], ssum[si], other_half(m[si]), other_half(ssum[si]), clamp);
            double mine = 0.0;
            if (valid && hh == 0 && si < sb.n_models) {
                mine = (double)ll;
                if (frame_ll) frame_ll[(int64_t)(sb.first_model + si) * n_frames + row] = ll;
            }
            // a frame in the band of the reference's partial-product flushes (lse.hpp) poisons the tile's partial
            const bool hot = valid && hh == 0 && si < sb.n_models && ll < band_hi;
            mine = wave_sum_f64(mine);
            if (__builtin_amdgcn_ballot_w64(hot) != 0) mine = SR_FLUSH_POISON;
            if (lane == 0) close_slot[si][wave] = mine;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (tid < sb.n_models) {
            const double *p = close_slot[tid];
            partial[(int64_t)tile_id * n_models + sb.first_model + tid] = ((p[0] + p[1]) + p[2]) + p[3];
        }
    }
}

template <int KQ, int KL>
static void launch_shared(const SharedLaunch &a) {
    dim3 grid((unsigned)((int64_t)a.n_groups * ((a.n_tiles + 7) / 8) * 8));
    hipLaunchKernelGGL((gmm_score_bx3_shared_kernel<KQ, KL>), grid, dim3(256), 0, ctx().stream, a.X, a.tiles,
                       reinterpret_cast<const uint4 *>(a.params), a.blocks, a.group_block_begin, a.center,
                       a.partial, a.frame_ll, a.n_frames, a.dim, a.n_models, a.n_mix_tiles, a.clamp,
                       a.n_groups, a.n_tiles, a.band_hi);
}

void launch_score_bx3_shared(const SharedLaunch &a, int KQ, int KL) {
#define SR_SH_CASE(Q, L) if (KQ == Q && KL == L) return launch_shared<Q, L>(a);
    SR_SH_CASE(1, 1) SR_SH_CASE(1, 2) SR_SH_CASE(2, 2) SR_SH_CASE(2, 3) SR_SH_CASE(3, 3) SR_SH_CASE(3, 4)
    SR_SH_CASE(4, 4) SR_SH_CASE(4, 5)
#undef SR_SH_CASE
    fail("no shared-sigma scoring kernel for %d + %d contraction steps", KQ, KL);
}

}  // namespace sr
