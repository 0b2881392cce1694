// gmm_score_mfma.hip -- second engine for the same scoring math (gmm.cc:176-202, :237-244,
// :533-569): the per-(frame, mixture) quadratic form written out as a contraction
//   log2 density_k(x) = sum_d ( A2_kd x'_d^2 + A1_kd x'_d ) + C_k ,   x' = x - centre
// and evaluated on the matrix cores with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exactly
// an fp32 FMA chain, so results are deterministic and of fp32-FMA accuracy).
//
// Why it exists: the 2-FMA vector kernel (gmm_score.hip) is FMA-issue bound and the vector ALU
// sustains ~60 % of the 157.3 TFLOP/s fp32 peak on this chip (measured: VALU 83 % busy at
// ~2.0 GHz), while the fp32 MFMA pipe runs at the same nominal rate with one VGPR per operand
// per lane, no broadcast traffic, and leaves the vector ALU free for the log-sum-exp.
// The expanded form cancels (|A2 x^2| + |A1 x| + |C| vs their sum); the dispatcher uses this
// engine only when `amp` = max_k sum_d (mu'_d/sigma_d)^2 is small enough for fp32 (score.hpp).
//
// Mapping: rows of the MFMA = 32 mixtures (A fragments streamed through LDS by LDS-DMA),
// columns = 32 frames (B fragments = the frame's (x'^2, x') components, resident in VGPRs for the
// whole kernel: lanes 0-31 hold the squares and the constant 1, lanes 32-63 the linear terms).
// A wave owns FT column tiles (32*FT frames); a workgroup = 4 waves = 128*FT frames of one
// utterance.  The accumulator layout puts a frame's 16 mixture rows in one lane, so the online
// log-sum-exp is lane-local; the two half-waves (other 16 rows) merge once per model.
#include "lse.hpp"
#include "score.hpp"
#include "wave_ops.hpp"

#include <algorithm>

namespace sr {

typedef float f32x16 __attribute__((ext_vector_type(16)));


__host__ __device__ constexpr int mfma_waves_per_eu(int dp, int ft) {
    const int regs = ft * (((dp + 1 + 3) & ~3) + 16) + 72;
    return regs <= 128 ? 4 : regs <= 140 ? 3 : regs <= 256 ? 2 : 1;
}

template <int DP, int FT>
__global__ __launch_bounds__(256, mfma_waves_per_eu(DP, FT))
void gmm_score_mfma_kernel(const float *__restrict__ X, const TileDesc *__restrict__ tiles,
                           const float4 *__restrict__ params, const ChunkDesc *__restrict__ chunks,
                           const int *__restrict__ group_chunk_begin,
                           const float *__restrict__ center, double *__restrict__ partial,
                           float *__restrict__ frame_ll, int64_t n_frames, int dim, int n_models,
                           int clamp, int n_groups, int n_tiles, float band_hi) {
    constexpr int KKP = (DP + 1 + 3) & ~3;     // contraction steps (2 k-indices each), padded to 4
    constexpr int KQ = KKP / 4;
    constexpr int TILE_F4 = KQ * 64;           // float4 per 32-mixture tile
    constexpr int CHUNK_F4 = MFMA_CT * TILE_F4;
    constexpr int PF = (CHUNK_F4 + 255) / 256;
    __shared__ float4 lds_a[CHUNK_F4];
    __shared__ float4 lds_b[CHUNK_F4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;                 // frame column inside a 32-frame tile
    const int hh = lane >> 5;                  // 0: squares + constant, 1: linear terms
    const int tile_lo = blockIdx.x & 7;        // XCD-aware order, as gmm_score_kernel
    const int q = blockIdx.x >> 3;
    const int g = q % n_groups;
    const int tile_id = (q / n_groups) * 8 + tile_lo;
    if (tile_id >= n_tiles) return;
    const TileDesc tile = tiles[tile_id];
    const int chunk_begin = group_chunk_begin[g];
    const int chunk_end = group_chunk_begin[g + 1];

    auto stage = [&](float4 *dst, const ChunkDesc cd) {
        const float4 *src = params + cd.offset_f4;
        const int n4 = cd.n_records * TILE_F4;
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const int base = (i * 4 + wave) * 64;
            if (base + lane < n4)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + base + lane),
                    (__attribute__((address_space(3))) void *)(dst + base), 16, 0, 0);
        }
    };
    stage(lds_a, chunks[chunk_begin]);

    // ---- resident B fragments: breg[ft][kk] = component 2*kk + hh of frame (wave*FT+ft)*32 + col ----
    float breg[FT][KKP];
    bool valid[FT];
    int64_t row[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ft++) {
        const int local = (wave * FT + ft) * 32 + col;
        valid[ft] = local < tile.count;
        row[ft] = tile.start + (valid[ft] ? local : 0);
        const float *src = X + row[ft] * dim;
#pragma unroll
        for (int kk = 0; kk < KKP; kk++) {
            float v = 0.0f;
            if (kk < DP) {
                if (kk < dim) {
                    const float xc = src[kk] - center[kk];
                    v = hh ? xc : xc * xc;
                }
            } else if (kk == DP) {
                v = hh ? 0.0f : 1.0f;
            }
            breg[ft][kk] = v;
        }
    }

    float m[FT], ssum[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ft++) {
        m[ft] = NEG_BIG;
        ssum[ft] = 0.0f;
    }
    const float near_thr = lse_near_threshold(clamp);
    dma_publish_barrier();

    auto do_chunk = [&](const float4 *cur, float4 *other, int c) {
        const ChunkDesc cd = chunks[c];
        if (c + 1 < chunk_end) stage(other, chunks[c + 1]);

        for (int t = 0; t < cd.n_records; t++) {
            const float4 *at = cur + t * TILE_F4 + lane;
            f32x16 acc[FT];
            // A fragments are fetched one step ahead of the MFMAs that consume them.  The first MFMA
            // of each chain takes a constant-zero C operand (an inline constant in the ISA) instead
            // of 16 zeroed VGPRs: vector-ALU instructions issued next to an MFMA stream delay its
            // issue (measured with scripts/ubench/mfma_f32_lse.hip), so every v_mov saved counts.
            const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            float4 a_nxt = at[0];
#pragma unroll
            for (int kq = 0; kq < KQ; kq++) {
                const float4 a = a_nxt;
                if (kq + 1 < KQ) a_nxt = at[(kq + 1) * 64];
#pragma unroll
                for (int ft = 0; ft < FT; ft++) {
                    acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, breg[ft][4 * kq + 0], kq == 0 ? zero16 : acc[ft], 0, 0, 0);
                    acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, breg[ft][4 * kq + 1], acc[ft], 0, 0, 0);
                    acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, breg[ft][4 * kq + 2], acc[ft], 0, 0, 0);
                    acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, breg[ft][4 * kq + 3], acc[ft], 0, 0, 0);
                }
            }
            // online log2-sum-exp over this lane's 16 mixture rows of each frame column
#pragma unroll
            for (int ft = 0; ft < FT; ft++) {
                float mx = acc[ft][0];
#pragma unroll
                for (int r = 1; r < 16; r++) mx = fmaxf(mx, acc[ft][r]);
                const float mn = fmaxf(m[ft], mx);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(mn < near_thr) != 0, 0)) {
                    lse_update16(acc[ft], m[ft], ssum[ft], near_thr);     // per-term flush next to DBL_MIN (lse.hpp)
                    continue;
                }
                // rows two at a time: the subtraction and the running sum as packed fp32 ops (fewer
                // vector-ALU instructions next to the MFMA stream; v_exp_f32 has no packed form)
                typedef float v2 __attribute__((ext_vector_type(2)));
                const v2 mn2 = {mn, mn};
                v2 e2 = {0.0f, 0.0f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const v2 d = (v2){acc[ft][r], acc[ft][r + 1]} - mn2;
                    e2 += (v2){__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
                }
                ssum[ft] = fmaf(ssum[ft], __builtin_amdgcn_exp2f(m[ft] - mn), e2.x + e2.y);
                m[ft] = mn;
            }
        }

        if (cd.model_done >= 0) {
            const int s = cd.model_done;
            double mine = 0.0;
            bool hot = false;              // a frame in the band of the reference's partial-product flushes (lse.hpp)
#pragma unroll
            for (int ft = 0; ft < FT; ft++) {
                // merge the two half-waves (the other 16 mixture rows of the same frame)
                const float ll = lse_close2(m[ft], ssum[ft], other_half(m[ft]), other_half(ssum[ft]), clamp);
                if (valid[ft] && hh == 0) {
                    mine += (double)ll;
                    if (frame_ll) frame_ll[(int64_t)s * n_frames + row[ft]] = ll;
                    hot |= ll < band_hi;
                }
                m[ft] = NEG_BIG;
                ssum[ft] = 0.0f;
            }
            mine = wave_sum_f64(mine);     // DPP + readlane: no LDS round trips in the per-model close
            if (__builtin_amdgcn_ballot_w64(hot) != 0) mine = SR_FLUSH_POISON;
            if (lane == 0) partial[((int64_t)tile_id * n_models + s) * 4 + wave] = mine;
        }
        dma_publish_barrier();
    };

    for (int c = chunk_begin; c < chunk_end; c += 2) {
        do_chunk(lds_a, lds_b, c);
        if (c + 1 < chunk_end) do_chunk(lds_b, lds_a, c + 1);
    }
}

template <int DP, int FT>
static void launch_mfma(const MfmaLaunch &a) {
    dim3 grid((unsigned)((int64_t)a.n_groups * ((a.n_tiles + 7) / 8) * 8));
    hipLaunchKernelGGL((gmm_score_mfma_kernel<DP, FT>), grid, dim3(256), 0, ctx().stream, a.X, a.tiles,
                       a.params, a.chunks, a.group_chunk_begin, a.center, a.partial, a.frame_ll,
                       a.n_frames, a.dim, a.n_models, a.clamp, a.n_groups, a.n_tiles, a.band_hi);
}

template <int DP>
static void dispatch_ft(const MfmaLaunch &a, int FT) {
    switch (FT) {
        case 1: launch_mfma<DP, 1>(a); break;
        case 2: launch_mfma<DP, 2>(a); break;
        case 3: launch_mfma<DP, 3>(a); break;
        case 4:
            if constexpr (DP <= 40) { launch_mfma<DP, 4>(a); break; }
            [[fallthrough]];
        default: fail("mfma engine: %d column tiles per wave not instantiated for dim %d", FT, DP);
    }
}

void launch_score_mfma(const MfmaLaunch &a, int DP, int FT) {
    switch (DP) {
        case 8: dispatch_ft<8>(a, FT); break;
        case 13: dispatch_ft<13>(a, FT); break;
        case 16: dispatch_ft<16>(a, FT); break;
        case 24: dispatch_ft<24>(a, FT); break;
        case 26: dispatch_ft<26>(a, FT); break;
        case 32: dispatch_ft<32>(a, FT); break;
        case 34: dispatch_ft<34>(a, FT); break;
        case 39: dispatch_ft<39>(a, FT); break;
        case 40: dispatch_ft<40>(a, FT); break;
        case 48: dispatch_ft<48>(a, FT); break;
        case 56: dispatch_ft<56>(a, FT); break;
        case 64: dispatch_ft<64>(a, FT); break;
        default: fail("no mfma scoring kernel for padded dim %d", DP);
    }
}

}  // namespace sr
