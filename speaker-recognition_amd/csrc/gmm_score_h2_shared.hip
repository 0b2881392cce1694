// gmm_score_h2_shared.hip -- split-fp16 scoring of speaker sets that share sigma and weights: a UBM
// and the speakers MAP-adapted from it (train_model_from_ubm moves the means only, gmmubm.cc:40-81;
// BASELINE configs[2] and [3]).  Math of gmm.cc:176-202, :237-244, :533-569 in the form
//   log2 density_sk(x) = Q_k(x) + L_sk(x),   Q_k = sum_d A2_kd x'_d^2   (the same for every model)
//                                            L_sk = sum_d A1_skd x'_d + C_sk
// with Q of a (mixture tile, frame tile) evaluated once per block of SHARED_SB = 15 models and fed to
// each model's chain as the C operand (as gmm_score_bx3_shared.hip), and three further cuts:
//
//  1. Two fp16 parts per operand, three part products (gmm_score_split.hip, scheme f16x2), and the
//     three products laid end to end as ONE contraction: [a_lo b_hi | a_hi b_lo | a_hi b_hi] is
//     3(D+1)-1 slots, padded to a multiple of 16 once instead of three times -- at D = 39 the linear
//     half is 119 slots = 8 MFMAs (3 x ceil(40/16) would be 9), the quadratic half 117 = 8.
//  2. Reference-offset log-sum-exp.  The models of such a set score a frame within a few nats of
//     each other, so the running maximum of the online log-sum-exp is replaced by ONE per-frame
//     offset O = log2 LL of the set's first model (the UBM), computed by a pre-pass of the generic
//     split-fp16 kernel: O is subtracted from Q once per 15 models, and a model-tile's epilogue is
//     16 x (v_exp_f32 + v_add_f32) -- no maxima, no subtractions, no rescaling, one float of state
//     per model.  Nothing is assumed: at the close a lane checks that its sum stayed inside
//     [2^-100, 2^100] and far enough above the reference's underflow boundary (lse.hpp); a
//     workgroup with a frame that fails the check writes (tile, block) to an exception list
//     instead of results, and the same kernel in ONLINE form (classic running maximum, lse.hpp
//     semantics) re-scores exactly those pairs right after the main launch.
//  3. The stream is staged through LDS two images (16 KiB) per barrier.
//
// Stream order per block of 15 models, per mixture tile: [Q][L_0]...[L_14], 16 images of KF KiB
// ([ks][lane][8 x fp16]: one ds_read_b128 per lane and MFMA).
#include "lse.hpp"
#include "score.hpp"
#include "wave_ops.hpp"

#include <algorithm>

namespace sr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H2S_G = 2;                         // images per LDS stage (one barrier per stage)
constexpr int H2S_ROUNDS_PER_LAUNCH = 12;
#ifndef H2S_CHAIN_PRIO
#define H2S_CHAIN_PRIO 0
#endif
constexpr float H2S_SUM_LO = 7.8886090522101181e-31f;    // 2^-100
constexpr float H2S_SUM_HI = 1.2676506002282294e+30f;    // 2^100
constexpr float H2S_LOG2E = 1.4426950408889634f;

// KN steps of one flat chain on `acc`; `init` is the C operand of the first MFMA
template <int KN>
__device__ __forceinline__ void h2s_chain(f32x16 &acc, const f32x16 &init, const uint4 *at, const f16x8 (&b)[KN]) {
    uint4 nx = at[0];
#pragma unroll
    for (int ks = 0; ks < KN; ks++) {
        const f16x8 a = __builtin_bit_cast(f16x8, nx);
        if (ks + 1 < KN) nx = at[(ks + 1) * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[ks], ks == 0 ? init : acc, 0, 0, 0);
    }
}

// The chain on A fragments already in registers: KN MFMAs back to back (a dependent chain on one
// accumulator issues every 32 cycles only when nothing sits between its links: putting the
// previous image's epilogue into the gaps was measured 30 % SLOWER, profiles/r02_h2s_variants.txt).
template <int KN, int KM>
__device__ __forceinline__ void h2s_chain_regs(f32x16 &acc, const f32x16 &init, const uint4 (&fr)[KM], const f16x8 (&b)[KN]) {
#pragma unroll
    for (int ks = 0; ks < KN; ks++)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[ks]), b[ks], ks == 0 ? init : acc, 0, 0, 0);
}

// Resident B fragments of one lane's frame for a flat slot table: desc = d | op << 8, op 0 zero,
// 1 high part, 2 low part, 3 the constant 1; `square` selects x'^2 (quadratic half) over x'.
template <int KN>
__device__ __forceinline__ void h2s_build_b(f16x8 (&out)[KN], const float *__restrict__ src,
                                            const float *__restrict__ center, const float *__restrict__ scale,
                                            const uint16_t *__restrict__ desc, int hh, bool square, float &zmax) {
#pragma unroll
    for (int ks = 0; ks < KN; ks++) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ds = desc[16 * ks + 8 * hh + j];
            const int d = ds & 0xff, op = ds >> 8;
            float xc = (src[d] - center[d]) * scale[d];
            if (op == 1 || op == 2) zmax = fmaxf(zmax, fabsf(xc));
            xc = fminf(fmaxf(xc, -255.0f), 255.0f);            // x'^2 stays below fp16's 65504
            float v = square ? xc * xc : xc;
            if (op == 3) v = 1.0f;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            const _Float16 pick = op == 2 ? l : h;
            const uint32_t bits = op == 0 ? 0u : (uint32_t)__builtin_bit_cast(unsigned short, pick);
            if (j & 1)
                w[j >> 1] |= bits << 16;
            else
                w[j >> 1] = bits;
        }
        out[ks] = __builtin_bit_cast(f16x8, make_uint4(w[0], w[1], w[2], w[3]));
    }
}

struct H2sArgs {
    const float *X;
    const TileDesc *tiles;
    const uint4 *params;
    const SharedBlock *blocks;
    const int *group_block_begin;
    const float *center, *scale;
    const uint16_t *q_desc, *l_desc;
    const float *ref_ll;          // [n_frames] natural-log LL of the reference model, unclamped (main pass)
    double *partial;
    float *frame_ll;
    int *oor_flag;
    int2 *exc_list;               // (tile, block) pairs for the ONLINE pass
    int *exc_count;
    int exc_cap;
    int64_t n_frames;
    int dim, n_models, n_mix_tiles, clamp, n_groups, n_tiles;
    int tile_base;                // first frame tile of this launch (long grids are cut into several launches)
    float log2_k;                 // log2 of the mixture count (bounds largest term >= LL - log2 K)
    int force_exc;                // testing: every workgroup of the main pass defers to the ONLINE pass
};

__host__ __device__ constexpr int h2s_waves_per_eu(int kqf, int klf, bool online) {
    return (online || kqf + klf > 16) ? 2 : 3;
}

template <int KQF, int KLF, bool ONLINE>
__global__ __launch_bounds__(256, h2s_waves_per_eu(KQF, KLF, ONLINE))
void gmm_score_h2s_kernel(const H2sArgs a) {
    constexpr int SB = SHARED_SB;
    constexpr int G = H2S_G;
    constexpr int Q_U4 = KQF * 64, L_U4 = KLF * 64;
    constexpr int IMG_U4 = Q_U4 > L_U4 ? Q_U4 : L_U4;          // every image padded to the larger of the two
    constexpr int STRIDE_U4 = (1 + SB) * IMG_U4;               // one mixture tile of one block
    constexpr int N_STAGES = (1 + SB) / G;
    static_assert((1 + SB) % G == 0 && (N_STAGES % 2) == 0, "stages must tile the 16 images and alternate buffers");
    __shared__ uint4 lds_a[G * IMG_U4];
    __shared__ uint4 lds_b[G * IMG_U4];
    __shared__ double close_slot[SB][4];
    __shared__ int wg_flag;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;
    const int hh = lane >> 5;

    // a stage is N_PIECES wave-instructions of 1 KiB; wave w issues pieces w, w + 4, ... (a scalar test)
    constexpr int N_PIECES = G * IMG_U4 / 64;
    auto stage_load = [&](uint4 *dst, const uint4 *src) {
#pragma unroll
        for (int i = 0; i < (N_PIECES + 3) / 4; i++) {
            const int piece = i * 4 + wave;
            if (piece < N_PIECES)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + piece * 64 + lane),
                    (__attribute__((address_space(3))) void *)(dst + piece * 64), 16, 0, 0);
        }
    };

    // Publishing LDS-DMA data to the other waves needs THIS wave's pieces landed before the barrier.
    // hipcc places its own vmcnt wait by alias analysis in front of this wave's ds_reads, which can
    // sit behind the barrier -- and it lost track of a DMA issued in the previous iteration of the
    // tile loop altogether (the stage-0 barrier had no wait: stale fragments for the second model of
    // a block whenever the parameters came from HBM rather than L2).  So: explicit.
    auto publish_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    const int n_work = ONLINE ? *a.exc_count : (int)gridDim.x;
    for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
        int tile_id, blk_begin, blk_end;
        if constexpr (ONLINE) {
            if (work >= a.exc_cap) break;
            const int2 e = a.exc_list[work];
            tile_id = e.x;
            blk_begin = e.y;
            blk_end = e.y + 1;
            __syncthreads();                       // the previous pair's LDS readers are done
        } else {
            const int tile_lo = work & 7;          // XCD-aware order, as gmm_score_kernel
            const int q = work >> 3;
            const int g = q % a.n_groups;
            tile_id = a.tile_base + (q / a.n_groups) * 8 + tile_lo;
            if (tile_id >= a.n_tiles) return;
            blk_begin = a.group_block_begin[g];
            blk_end = a.group_block_begin[g + 1];
        }
        const TileDesc tile = a.tiles[tile_id];

        // ---- resident B fragments of this lane's frame ----
        f16x8 bq[KQF], bl[KLF];
        const int local = wave * 32 + col;
        const bool valid = local < tile.count;
        const int64_t row = tile.start + (valid ? local : 0);
        float zmax = 0.0f;
        h2s_build_b<KQF>(bq, a.X + row * a.dim, a.center, a.scale, a.q_desc, hh, true, zmax);
        h2s_build_b<KLF>(bl, a.X + row * a.dim, a.center, a.scale, a.l_desc, hh, false, zmax);
        if (zmax >= 255.0f) atomicOr(a.oor_flag, 1);           // saturated: the host re-scores on the fp32-grade engines
        float off = 0.0f;                                      // per-frame offset O (log2 units)
        if constexpr (!ONLINE) off = a.ref_ll[row] * H2S_LOG2E;
        const float near_thr = lse_near_threshold(a.clamp);
        // main pass: the largest term is >= LL - log2 K; below this the ONLINE pass decides
        const float safe_ll2 = a.clamp ? LSE_MINLOG2 + LSE_NEAR + a.log2_k : -3.0e38f;

        const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // Every workgroup streams the same parameter images from L2, and the 4 MiB L2 of an XCD cannot hold
        // the whole stream: the workgroups of an XCD hit in L2 only while they sweep in phase.  They do for
        // the first ~30 rounds of a launch (all start at block 0 together: 3 % L2 misses on 3 M frames), then
        // the start times drift apart (42 % misses = 1.1 TB of fabric reads over 10 M frames).  Steering the
        // starting block by a shared hint or by the clock made it WORSE (44 % at 3 M frames: a workgroup that
        // starts mid-sweep is out of phase with everyone who started at 0), so long grids are simply cut into
        // launches of H2S_ROUNDS_PER_LAUNCH rounds, each of which starts in phase (launch_h2s).
        for (int blk = blk_begin; blk < blk_end; blk++) {
            const SharedBlock sb = a.blocks[blk];
            const uint4 *stream = a.params + sb.offset_u4;
            float m[ONLINE ? SB : 1], ssum[SB];
#pragma unroll
            for (int si = 0; si < SB; si++) {
                if constexpr (ONLINE) m[si] = NEG_BIG;
                ssum[si] = 0.0f;
            }
            if constexpr (ONLINE) {
                __syncthreads();                      // previous block's readers are done with lds_a
                stage_load(lds_a, stream);
                publish_barrier();
                for (int t = 0; t < a.n_mix_tiles; t++) {
                    const uint4 *tsrc = stream + (size_t)t * STRIDE_U4;
                    f32x16 qacc;
#pragma unroll
                    for (int st = 0; st < N_STAGES; st++) {
                        const uint4 *cur = (st & 1) ? lds_b : lds_a;
                        uint4 *other = (st & 1) ? lds_a : lds_b;
                        if (st + 1 < N_STAGES)
                            stage_load(other, tsrc + (size_t)(st + 1) * G * IMG_U4);
                        else if (t + 1 < a.n_mix_tiles)
                            stage_load(other, tsrc + STRIDE_U4);            // next tile's first stage -> lds_a
#pragma unroll
                        for (int gi = 0; gi < G; gi++) {
                            const int img = st * G + gi;
                            const uint4 *at = cur + gi * IMG_U4 + lane;
                            if (img == 0) {
                                h2s_chain<KQF>(qacc, zero16, at, bq);
                            } else {
                                const int si = img - 1;
                                f32x16 acc;
                                h2s_chain<KLF>(acc, qacc, at, bl);
                                lse_update16(acc, m[si], ssum[si], near_thr);
                                // the model loop is unrolled and s_barrier orders memory, not ALU work:
                                // pin each epilogue where it is written (see gmm_score_bx3_shared.hip)
                                asm volatile("" : "+v"(m[si]), "+v"(ssum[si]));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        publish_barrier();
                    }
                }
            } else {
                // Main pass.  Per image: (1) the chain -- its KN A fragments already sit in registers,
                // so the MFMAs issue back to back; (2) the NEXT image's fragments are requested from
                // LDS; (3) this image's epilogue runs while they arrive.  A stage's LDS buffer is free
                // as soon as the fragments of its last image are in registers, so the barrier (and
                // the LDS-DMA of the stage after next into the freed buffer) comes before that image's
                // epilogue, not after it.
                constexpr int KM = KQF > KLF ? KQF : KLF;
                uint4 fr[KM];
                auto load_frags = [&](const uint4 *at, int kn) {
#ifdef H2S_DEBUG_ONE_FRAG         /* energy experiment only: one LDS read per image, wrong results */
                    fr[0] = at[0];
#pragma unroll
                    for (int ks = 1; ks < KM; ks++) fr[ks] = fr[0];
#else
#pragma unroll
                    for (int ks = 0; ks < KM; ks++)
                        if (ks < kn) fr[ks] = at[ks * 64];
#endif
                };
                const int n_stage_total = a.n_mix_tiles * N_STAGES;
                __syncthreads();                      // previous block's readers are done with both buffers
                stage_load(lds_a, stream);
                if (n_stage_total > 1) stage_load(lds_b, stream + (size_t)G * IMG_U4);
                publish_barrier();                    // (drains both; the second is not needed yet, once per block)
                load_frags(lds_a + lane, KQF);
                for (int t = 0; t < a.n_mix_tiles; t++) {
                    const uint4 *tsrc = stream + (size_t)t * STRIDE_U4;
                    const bool more_tiles = t + 1 < a.n_mix_tiles;
                    f32x16 qacc;
#pragma unroll
                    for (int st = 0; st < N_STAGES; st++) {
                        uint4 *cur = (st & 1) ? lds_b : lds_a;
                        const uint4 *nxt = (st & 1) ? lds_a : lds_b;
#pragma unroll
                        for (int gi = 0; gi < G; gi++) {
                            const int img = st * G + gi;
                            f32x16 acc;
                            // the matrix pipe must never wait for an issue slot: a wave in its chain outranks
                            // the waves in their (vector-ALU) epilogues
                            __builtin_amdgcn_s_setprio(H2S_CHAIN_PRIO);
                            if (img == 0)
                                h2s_chain_regs<KQF>(qacc, zero16, fr, bq);
                            else
                                h2s_chain_regs<KLF>(acc, qacc, fr, bl);
                            __builtin_amdgcn_s_setprio(0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (gi == G - 1) {
                                // every wave holds its fragments of this stage: `cur` may be refilled, and
                                // the stage after this one has landed (its DMA was issued a stage ago)
                                publish_barrier();
                                if (st + 2 < N_STAGES)
                                    stage_load(cur, tsrc + (size_t)(st + 2) * G * IMG_U4);
                                else if (more_tiles)
                                    stage_load(cur, tsrc + STRIDE_U4 + (size_t)(st + 2 - N_STAGES) * G * IMG_U4);
                                if (st + 1 < N_STAGES || more_tiles)
                                    load_frags(nxt + lane, (st + 1 < N_STAGES) ? KLF : KQF);
                            } else {
                                load_frags(cur + (gi + 1) * IMG_U4 + lane, KLF);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (img == 0) {
#pragma unroll
                                for (int r = 0; r < 16; r++) qacc[r] -= off;
                            } else {
                                float e0 = 0.0f, e1 = 0.0f;
#ifdef H2S_DEBUG_NO_EPILOGUE      /* energy experiment only: wrong results */
                                e0 = acc[0]; e1 = acc[15];
#else
#pragma unroll
                                for (int r = 0; r < 16; r += 2) {
                                    e0 += __builtin_amdgcn_exp2f(acc[r]);
                                    e1 += __builtin_amdgcn_exp2f(acc[r + 1]);
                                }
#endif
                                ssum[img - 1] += e0 + e1;
                                asm volatile("" : "+v"(ssum[img - 1]));
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            // ---- close the block's models ----
            bool bad = false;
            float ll_keep[SB];
#pragma unroll
            for (int si = 0; si < SB; si++) {
                float ll;
                if constexpr (ONLINE) {
                    ll = lse_close2(m[si], ssum[si], other_half(m[si]), other_half(ssum[si]), a.clamp);
                } else {
                    const float tot = ssum[si] + other_half(ssum[si]);
                    const float ll2 = off + log2f(tot);
                    ll = LSE_LN2 * ll2;
                    // the offset form is only trusted well inside fp32's exponent range and well above
                    // the reference's underflow boundary
                    const bool ok = tot >= H2S_SUM_LO && tot <= H2S_SUM_HI && ll2 >= safe_ll2;
#if defined(H2S_DEBUG_NO_EPILOGUE) || defined(H2S_DEBUG_ONE_FRAG)
                    (void)ok;                                   // energy experiments: results are wrong by construction
#else
                    bad |= (valid && si < sb.n_models && !ok) || a.force_exc;
#endif
                }
                ll_keep[si] = ll;
            }
            bool redo = false;
            if constexpr (!ONLINE) {
                if (tid == 0) wg_flag = 0;
                __syncthreads();
                if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) wg_flag = 1;
                __syncthreads();
                redo = wg_flag != 0;
                if (redo && tid == 0) {
                    const int idx = atomicAdd(a.exc_count, 1);
                    if (idx < a.exc_cap) a.exc_list[idx] = make_int2(tile_id, blk);
                }
            }
            if (!redo) {
#pragma unroll
                for (int si = 0; si < SB; si++) {
                    double mine = 0.0;
                    if (valid && hh == 0 && si < sb.n_models) {
                        mine = (double)ll_keep[si];
                        if (a.frame_ll) a.frame_ll[(int64_t)(sb.first_model + si) * a.n_frames + row] = ll_keep[si];
                    }
                    mine = wave_sum_f64(mine);
                    if (lane == 0) close_slot[si][wave] = mine;
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
                if (tid < sb.n_models) {
                    const double *p = close_slot[tid];
                    a.partial[(int64_t)tile_id * a.n_models + sb.first_model + tid] = ((p[0] + p[1]) + p[2]) + p[3];
                }
            }
        }
        if constexpr (!ONLINE) break;             // the main pass maps one workgroup to one (tile, group)
    }
}

template <int KQF, int KLF>
static void launch_h2s(const H2sLaunch &l) {
    H2sArgs a;
    a.X = l.X;
    a.tiles = l.tiles;
    a.params = reinterpret_cast<const uint4 *>(l.params);
    a.blocks = l.blocks;
    a.group_block_begin = l.group_block_begin;
    a.center = l.center;
    a.scale = l.scale;
    a.q_desc = l.q_desc;
    a.l_desc = l.l_desc;
    a.ref_ll = l.ref_ll;
    a.partial = l.partial;
    a.frame_ll = l.frame_ll;
    a.oor_flag = l.oor_flag;
    a.exc_list = reinterpret_cast<int2 *>(l.exc_list);
    a.exc_count = l.exc_count;
    a.exc_cap = l.exc_cap;
    a.n_frames = l.n_frames;
    a.dim = l.dim;
    a.n_models = l.n_models;
    a.n_mix_tiles = l.n_mix_tiles;
    a.clamp = l.clamp;
    a.n_groups = l.n_groups;
    a.n_tiles = l.n_tiles;
    a.log2_k = l.log2_k;
    a.force_exc = l.force_exc;
    // long grids in launches of ~H2S_ROUNDS_PER_LAUNCH rounds of resident workgroups (see the kernel)
    const int resident = ctx().n_cu * h2s_waves_per_eu(KQF, KLF, false);
    int tiles_per_launch = l.tiles_per_launch > 0 ? l.tiles_per_launch
                           : std::max(8, (H2S_ROUNDS_PER_LAUNCH * resident / std::max(1, l.n_groups)) / 8 * 8);
    for (int base = 0; base < l.n_tiles; base += tiles_per_launch) {
        a.tile_base = base;
        const int n = std::min(tiles_per_launch, l.n_tiles - base);
        dim3 grid((unsigned)((int64_t)l.n_groups * ((n + 7) / 8) * 8));
        hipLaunchKernelGGL((gmm_score_h2s_kernel<KQF, KLF, false>), grid, dim3(256), 0, ctx().stream, a);
    }
    a.tile_base = 0;
    // the exception pass: persistent workgroups over the (tile, block) list the main pass left
    const int fix_grid = std::max(1, std::min(l.exc_cap, ctx().n_cu * 2));
    hipLaunchKernelGGL((gmm_score_h2s_kernel<KQF, KLF, true>), dim3((unsigned)fix_grid), dim3(256), 0, ctx().stream, a);
}

void launch_score_h2_shared(const H2sLaunch &l, int KQF, int KLF) {
#define SR_H2S_CASE(Q, L) if (KQF == Q && KLF == L) return launch_h2s<Q, L>(l);
    // KQF = ceil(3D/16), KLF = ceil((3D+2)/16): equal, or one apart at D = 5, 16, 21, 32, 37, 48
    SR_H2S_CASE(1, 1) SR_H2S_CASE(1, 2) SR_H2S_CASE(2, 2) SR_H2S_CASE(3, 3) SR_H2S_CASE(3, 4) SR_H2S_CASE(4, 4)
    SR_H2S_CASE(4, 5) SR_H2S_CASE(5, 5) SR_H2S_CASE(6, 6) SR_H2S_CASE(6, 7) SR_H2S_CASE(7, 7)
    SR_H2S_CASE(7, 8) SR_H2S_CASE(8, 8) SR_H2S_CASE(9, 9) SR_H2S_CASE(9, 10)
#undef SR_H2S_CASE
    fail("no split-fp16 shared-sigma scoring kernel for %d + %d contraction steps", KQF, KLF);
}

}  // namespace sr
