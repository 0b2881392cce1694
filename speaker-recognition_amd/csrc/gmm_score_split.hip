// gmm_score_split.hip -- matrix-core engines for the scoring math of gmm.cc:176-202, :237-244, :533-569:
// the expanded quadratic form of gmm_score_mfma.hip
//   log2 density_k(x) = sum_d ( A2_kd x'_d^2 + A1_kd x'_d ) + C_k ,   x' = (x - centre) * scale
// evaluated on the 16-bit matrix cores at (near-)fp32 accuracy by splitting every fp32 operand into
// 16-bit parts whose part products are exact in fp32, accumulated in fp32 by the MFMA:
//
//   scheme bf16x3 (score_engine 3): three bf16 parts (8+8+8 = all 24 significand bits, bf16 has
//       fp32's exponent range), six part products a0b0 + (a0b1 + a1b0) + (a1b1 + a0b2 + a2b0);
//       dropped products < 2^-24 each: fp32-grade, no range restrictions.
//   scheme f16x2  (score_engine 5): two fp16 parts (11+11 = 22 bits), three part products
//       a0b0 + a0b1 + a1b0; dropped a1b1 < 2^-22 -- half the MFMAs.  fp16's narrow exponent range is
//       handled by a per-dimension power-of-two scale folded into both operands (exact), gradual
//       underflow of the low parts (fp16 subnormals are honoured by v_mfma_f32_32x32x16_f16), and a
//       saturation flag: a frame whose scaled |x'| reaches 255 is clamped and reported, and the
//       host re-scores the batch with the bf16x3 engine (score.hpp).
//
// Why: gfx950 has no tf32/xf32 and its fp32 MFMA runs at the vector rate (157 TFLOP/s); the 16-bit
// MFMA is 16x that, so 6 (3) part products cost 6/16 (3/16) of one fp32 MFMA pass over the tile.
//
// Mapping: MFMA rows = 32 mixtures, A fragments streamed through LDS by LDS-DMA (one 32-mixture
// tile = KS*PARTS fragments of 1 KiB per chunk, double-buffered); MFMA columns = 32 frames, B
// fragments (the parts of the frame's (x'^2, x', 1) vector) resident in VGPRs for the whole kernel.
// A wave owns FT column tiles; the accumulator layout keeps a frame's 16 mixture rows in one lane,
// so the online log-sum-exp is lane-local and the two half-waves merge once per model.
//
// Tried in round 3 and measured slower: this engine in the shape the shared-sigma engine's large batches take -- one 12-wave
// workgroup per CU, a 32-frame tile per wave, ONE copy of the stream in LDS (two stages of four chunks), a barrier per four
// chunks instead of per chunk.  Bit-identical sums; 3.40 ms against 2.83 on configs[1] (100 x 64 x 39, 1 M frames), 0.53
// against 0.34 on the 256 x 39 point, 0.138 against 0.123 at 10 x 32 x 13.  Here a chunk is 15 MFMAs against a ~60-instruction
// online log-sum-exp (running maximum, rescaling), a wave runs the two one after the other, and what hides the one behind the
// other is MORE waves per SIMD (five 4-wave workgroups = 5 per SIMD) -- worth more than the third of the L2 -> LDS stream the
// wide shape saves.  (The shared-sigma kernel's epilogue is 32 instructions with no maximum: there the wide shape wins.)
#include "lse.hpp"
#include "score.hpp"
#include "split_prologue.hpp"
#include "split_schemes.hpp"
#include "wave_ops.hpp"

#include <algorithm>

namespace sr {

__host__ __device__ constexpr int split_waves_per_eu(int parts, int ks, int ft) {
    // (the long contractions get headroom: at ks * parts >= 14 the 128-register bucket spilled 48 bytes per lane once every wave
    // had a tile of its own, round 4; the short ones must stay at 5 waves per SIMD -- what hides this kernel's log-sum-exp)
    const int regs = ft * (ks * parts * 4 + 16) + (ks * parts * ft >= 14 ? 32 : 16) + 12 * parts;
    return regs <= 96 ? 5 : regs <= 128 ? 4 : regs <= 168 ? 3 : regs <= 256 ? 2 : 1;
}

template <typename SC, int KS, int FT>
__global__ __launch_bounds__(256, split_waves_per_eu(SC::PARTS, KS, FT))
void gmm_score_split_kernel(const float *__restrict__ X, const TileDesc *__restrict__ tiles,
                            const uint4 *__restrict__ params, const ChunkDesc *__restrict__ chunks,
                            const int *__restrict__ group_chunk_begin,
                            const float *__restrict__ center, const float *__restrict__ scale,
                            double *__restrict__ partial, float *__restrict__ frame_ll,
                            int *__restrict__ oor_flag, int64_t n_frames, int dim, int n_models,
                            int clamp, int n_groups, int n_tiles, float band_hi) {
    constexpr int P = SC::PARTS;
    constexpr int TILE_U4 = KS * P * 64;       // 16-byte fragments-per-lane of one 32-mixture tile
    constexpr int PF = (TILE_U4 + 255) / 256;
    typedef typename SC::frag frag;
    __shared__ uint4 lds_a[TILE_U4];
    __shared__ uint4 lds_b[TILE_U4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;                 // frame column inside a 32-frame tile
    const int hh = lane >> 5;                  // which 8 of the 16 contraction indices of a step
    // Round 4: the unit of work is a 32-frame tile of ONE utterance per (wave, column tile) -- as in the wide form of this engine
    // (gmm_score_splitp.hip) and the shared-sigma engines -- instead of a 128-frame tile per workgroup: a workgroup takes 4 FT
    // consecutive tiles and may straddle utterances, short utterances pad to 32 frames instead of 128, and both forms of the engine
    // leave the SAME partial per (32-frame tile, model): which one a batch's size selects does not show in the results.
    const int tile_lo = blockIdx.x & 7;        // XCD-aware order, as gmm_score_kernel
    const int q = blockIdx.x >> 3;
    const int g = q % n_groups;
    const int tile0 = ((q / n_groups) * 8 + tile_lo) * (4 * FT);
    if (tile0 >= n_tiles) return;
    const int chunk_begin = group_chunk_begin[g];
    const int chunk_end = group_chunk_begin[g + 1];

    // every chunk of this layout is one mixture tile of TILE_U4 fragments: chunk c starts at c * TILE_U4
    auto stage = [&](uint4 *dst, int c) {
        const uint4 *src = params + (size_t)c * TILE_U4;
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const int base = (i * 4 + wave) * 64;
            if (base + lane < TILE_U4)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + base + lane),
                    (__attribute__((address_space(3))) void *)(dst + base), 16, 0, 0);
        }
    };
    stage(lds_a, chunk_begin);
    int done_next = chunks[chunk_begin].model_done;   // fetched one chunk ahead of its use

    // ---- resident B fragments (split_prologue.hpp): breg[ft][ks][part] = the 16-bit parts of this lane's 8 slots of step ks.
    //      (Tried, round 4: the tile's rows as coalesced LDS-DMA into a stage buffer, one row per lane read back -- the row
    //      fetch alone 18 % faster, the kernel 9 % slower: the wait for the rows is also a wait for the first chunk.) ----
    frag breg[FT][KS][P];
    bool valid[FT], has[FT];
    int tile_id[FT];
    int64_t tile_start[FT];                    // (wave-uniform: a lane's row = tile_start + col where it is needed)
    float zmax = 0.0f;
#pragma unroll
    for (int ft = 0; ft < FT; ft++) {
        tile_id[ft] = tile0 + wave * FT + ft;
        has[ft] = tile_id[ft] < n_tiles;
        const TileDesc tile = tiles[has[ft] ? tile_id[ft] : n_tiles - 1];
        valid[ft] = has[ft] && col < tile.count;
        tile_start[ft] = tile.start;
        split_frame_fragments<SC, KS>(X + (tile.start + (valid[ft] ? col : 0)) * dim, dim, hh, center, scale, breg[ft], zmax);
    }
    if constexpr (SC::SCALED) {
        // NaN features compare false and fall through to the arithmetic, which propagates them
        if (zmax >= 255.0f) atomicOr(oor_flag, 1);
    }

    float m[FT], ssum[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ft++) {
        m[ft] = NEG_BIG;
        ssum[ft] = 0.0f;
    }
    const float near_thr = lse_near_threshold(clamp);
    dma_publish_barrier();

    auto do_chunk = [&](const uint4 *cur, uint4 *other, int c) {
        const int model_done = done_next;
        if (c + 1 < chunk_end) {
            stage(other, c + 1);
            done_next = chunks[c + 1].model_done;
        }

        const uint4 *at = cur + lane;
        f32x16 acc[FT];
        const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // A parts are fetched one contraction step ahead of the MFMAs that consume them; the first
        // MFMA of each chain takes the inline-constant zero as C.
        uint4 nx[P];
#pragma unroll
        for (int pi = 0; pi < P; pi++) nx[pi] = at[pi * 64];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            frag a[P];
#pragma unroll
            for (int pi = 0; pi < P; pi++) a[pi] = __builtin_bit_cast(frag, nx[pi]);
            if (ks + 1 < KS) {
#pragma unroll
                for (int pi = 0; pi < P; pi++) nx[pi] = at[((ks + 1) * P + pi) * 64];
            }
#pragma unroll
            for (int pr = 0; pr < SC::NPROD; pr++)
#pragma unroll
                for (int ft = 0; ft < FT; ft++)
                    acc[ft] = SC::mfma(a[SC::AI[pr]], breg[ft][ks][SC::BI[pr]],
                                       (ks == 0 && pr == 0) ? zero16 : acc[ft]);
        }
        // online log2-sum-exp over this lane's 16 mixture rows of each frame column (lse.hpp)
#pragma unroll
        for (int ft = 0; ft < FT; ft++) lse_update16(acc[ft], m[ft], ssum[ft], near_thr);

        if (model_done >= 0) {
            const int s = model_done;
#pragma unroll
            for (int ft = 0; ft < FT; ft++) {
                // merge the two half-waves (the other 16 mixture rows of the same frame); the
                // reference's underflow behaviour (safe_log, gmm.cc:34-38) is in lse.hpp
                const float ll = lse_close2(m[ft], ssum[ft], other_half(m[ft]), other_half(ssum[ft]), clamp);
                double mine = 0.0;
                bool hot = false;          // a frame in the band of the reference's partial-product flushes (lse.hpp)
                if (valid[ft] && hh == 0) {
                    mine = (double)ll;
                    if (frame_ll) frame_ll[(int64_t)s * n_frames + tile_start[ft] + col] = ll;
                    hot = ll < band_hi;
                }
                m[ft] = NEG_BIG;
                ssum[ft] = 0.0f;
                mine = wave_sum_f64(mine);
                if (__builtin_amdgcn_ballot_w64(hot) != 0) mine = SR_FLUSH_POISON;
                if (lane == 0 && has[ft]) partial[(int64_t)tile_id[ft] * n_models + s] = mine;
            }
        }
        dma_publish_barrier();
    };

    for (int c = chunk_begin; c < chunk_end; c += 2) {
        do_chunk(lds_a, lds_b, c);
        if (c + 1 < chunk_end) do_chunk(lds_b, lds_a, c + 1);
    }
}

template <typename SC, int KS, int FT>
static void launch_split(const MfmaLaunch &a) {
    const int n_wg = (a.n_tiles + 4 * FT - 1) / (4 * FT);        // `a.tiles` = 32-frame tiles, 4 FT per workgroup
    dim3 grid((unsigned)((int64_t)a.n_groups * ((n_wg + 7) / 8) * 8));
    hipLaunchKernelGGL((gmm_score_split_kernel<SC, KS, FT>), grid, dim3(256), 0, ctx().stream, a.X, a.tiles,
                       reinterpret_cast<const uint4 *>(a.params), a.chunks, a.group_chunk_begin, a.center,
                       a.scale, a.partial, a.frame_ll, a.oor_flag, a.n_frames, a.dim, a.n_models, a.clamp,
                       a.n_groups, a.n_tiles, a.band_hi);
}

template <typename SC, int KS>
static void dispatch_split_ft(const MfmaLaunch &a, int FT) {
    if (FT == 1) return launch_split<SC, KS, 1>(a);
    if constexpr (KS <= 6) {
        if (FT == 2) return launch_split<SC, KS, 2>(a);
    }
    fail("split engine: %d column tiles per wave not instantiated for %d contraction steps", FT, KS);
}

int split_max_ft(int ks) { return ks <= 6 ? 2 : 1; }

template <typename SC>
static void dispatch_split(const MfmaLaunch &a, int KS, int FT) {
    switch (KS) {
        case 1: dispatch_split_ft<SC, 1>(a, FT); break;
        case 2: dispatch_split_ft<SC, 2>(a, FT); break;
        case 3: dispatch_split_ft<SC, 3>(a, FT); break;
        case 4: dispatch_split_ft<SC, 4>(a, FT); break;
        case 5: dispatch_split_ft<SC, 5>(a, FT); break;
        case 6: dispatch_split_ft<SC, 6>(a, FT); break;
        case 7: dispatch_split_ft<SC, 7>(a, FT); break;
        case 8: dispatch_split_ft<SC, 8>(a, FT); break;
        case 9: dispatch_split_ft<SC, 9>(a, FT); break;
        default: fail("no split scoring kernel for %d contraction steps", KS);
    }
}

void launch_score_split(const MfmaLaunch &a, int scheme, int KS, int FT) {
    if (scheme == SPLIT_F16X2)
        dispatch_split<f16x2>(a, KS, FT);
    else if (scheme == SPLIT_F16X1)
        dispatch_split<f16x1>(a, KS, 1);
    else
        dispatch_split<bf16x3>(a, KS, FT);
}

}  // namespace sr
