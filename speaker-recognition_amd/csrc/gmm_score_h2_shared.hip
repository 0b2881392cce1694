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
//     wave with a frame that fails the check writes (32-frame tile, block) to an exception list
//     instead of results, and gmm_score_h2s_online_kernel (classic running maximum, lse.hpp
//     semantics, one wave per pair, fragments straight from L2) re-scores exactly those pairs right
//     after the main launch.  The decision is per 32-frame tile of ONE utterance, so an utterance's
//     results do not depend on the batch around it.
//  3. The stream is staged through LDS G images per barrier; the unit of work is a 32-frame tile per wave
//     (tiles never straddle utterances, workgroups may), so a workgroup can be 4 waves (three of them per
//     CU, each with its own copy of the stream in LDS) or 12 waves -- ONE per CU, one copy of the stream
//     for all of them: a third of the L2 -> LDS traffic, which is what the 4-wave form stalls on
//     (profiles/r02_h2s_stalls.txt).
//
// Stream order per block of 15 models, per mixture tile: [Q][L_0]...[L_14], 16 images of KF KiB
// ([ks][lane][8 x fp16]: one ds_read_b128 per lane and MFMA).
#include "lse.hpp"
#include "score.hpp"
#include "wave_ops.hpp"

#include <algorithm>
#include <cstdint>
#include <type_traits>

namespace sr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H2S_ROUNDS_PER_LAUNCH = 12;
constexpr float H2S_SUM_LO = 7.8886090522101181e-31f;    // 2^-100
constexpr float H2S_SUM_HI = 1.2676506002282294e+30f;    // 2^100
constexpr float H2S_LOG2E = 1.4426950408889634f;

// KN steps of one flat chain per column tile on A fragments already in registers; `init` is the C operand of the
// first MFMA.  The MFMAs go back to back here.  (Round 2 concluded from scripts/ubench/mfma_lse_inwave.hip that a wave
// cannot hide its own vector work behind its own MFMAs; round 3 found that benchmark's schedule was not what it asked
// for -- see gmm_score_h2p_kernel below, which interleaves, and why that changes the time by 1-2 % only.)  COLS = 2 (two column tiles per wave sharing every A fragment) was measured slower and is
// not instantiated.
template <int KN, int KM, int COLS>
__device__ __forceinline__ void h2s_chain_regs(f32x16 (&acc)[COLS], const f32x16 (&init)[COLS], const uint4 (&fr)[KM],
                                               const f16x8 (&b)[COLS][KN]) {
#pragma unroll
    for (int ks = 0; ks < KN; ks++)
#pragma unroll
        for (int c = 0; c < COLS; c++)
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[ks]), b[c][ks],
                                                            ks == 0 ? init[c] : acc[c], 0, 0, 0);
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

// value of `v` in lane (lane + shift) & 63 (ds_bpermute: an LDS-pipe round trip; the packed close of a few percent of the waves only)
__device__ __forceinline__ double lane_shift_down_f64(double v, int shift, int lane) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    const int addr = ((lane + shift) & 63) << 2;
    b.i[0] = __builtin_amdgcn_ds_bpermute(addr, a.i[0]);
    b.i[1] = __builtin_amdgcn_ds_bpermute(addr, a.i[1]);
    return b.d;
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
    int2 *exc_list;               // the ONLINE pass's work: per block, {tile, mask of the COLUMNS (frames) the offset form could not vouch for}
                                  // ([n_blocks][n_tiles]); behind it the plan of the pass: int [n_blocks][n_tiles + 1], entry index at which
                                  // each wave's item starts (h2s_plan_kernel)
    int *exc_count;               // ... how many entries ([n_blocks]), and behind them the number of items per block ([n_blocks])
    int n_blocks;
    int64_t n_frames;
    int dim, n_models, n_mix_tiles, clamp, n_groups, n_tiles;
    int tile_base;                // first frame tile of this launch (long grids are cut into several launches)
    int rows8;                    // rows of 8 workgroups (one per XCD) this launch has per model group
    int group_major;              // launch order of the model groups' workgroups (h2s_wg_assignment): 0 group-fastest, 1 group-major
    int n_wg;                     // workgroups (of TILES_WG tiles) per group in this launch

    float log2_k;                 // log2 of the mixture count (bounds largest term >= LL - log2 K)
    int force_exc;                // testing: every workgroup of the main pass defers to the ONLINE pass
    int plan_inline;              // the exception pass forms its plan itself (no h2s_plan_kernel launch in front of it)
    int n_units;                  // what h2s_wg_assignment hands out: 32-frame tiles, or -- pipelined kernel -- work items (the padded
                                  // count of ensure_work_table: every unit of a launched workgroup is readable)
    float band_hi;                // below it a frame goes to the partial-product path (lse.hpp): by way of the ONLINE pass
};

// Which (model group, frame tiles) a workgroup of the main kernels takes.  Consecutive blockIdx go to consecutive XCDs (8 of
// them, each with its own L2), and workgroups of different groups stream different blocks' images:
//   0  group-fastest (rounds 2-3): one workgroup of every group on each XCD in turn;
//   1  group-major (round 4): a group's workgroups consecutive in launch order, spread over the 8 XCDs;
//   2  (round 6, the model-split kernel of a serving decision only) group g's workgroups all on XCD g % 8.
// (Tried: XCD-major -- every XCD a contiguous run of (group, tile) pairs, whole groups where there are enough tiles, so that a
// block's images come into ONE L2 only.  No better than 0: 64 utterances x 300 frames 0.755 ms against 0.652 for group-major.)
__device__ __forceinline__ bool h2s_wg_assignment(const H2sArgs &a, int tiles_wg, int &g, int &tile0) {
    const int wg_lo = blockIdx.x & 7, q = blockIdx.x >> 3;
    int t;
    if (a.group_major == 2) {           // a group's workgroups on ONE XCD (its blocks' images come into one L2, once)
        g = wg_lo + 8 * (q / a.n_wg);
        t = q - (q / a.n_wg) * a.n_wg;
        tile0 = a.tile_base + t * tiles_wg;
        return g < a.n_groups && tile0 < a.n_units;
    }
    if (a.group_major) {
        g = q / a.rows8;
        t = (q - g * a.rows8) * 8 + wg_lo;
    } else {
        g = q % a.n_groups;
        t = (q / a.n_groups) * 8 + wg_lo;
    }
    tile0 = a.tile_base + t * tiles_wg;
    return tile0 < a.n_units && t < a.n_wg;
}

// Close of one block's models for one 32-frame tile (both main kernels): the offset form is only trusted well inside fp32's
// exponent range and well above the reference's underflow boundary.  A FRAME outside (for any model of the block) is left out
// of the tile's partials and its column is listed -- {tile, column mask} on the block's exception list (round 6; through round 5
// the whole tile went: at configs[3]'s 0.1 % outlier frames 3.1 % of the tiles re-scored in full, 0.37 s of a 6.2 s pass);
// the other frames leave one partial per model: a fixed-order float64 sum over the wave's lanes, to which the exception pass
// adds the listed frames' values.  A frame's fate depends on that frame alone, so an utterance's results do not depend on the
// batch around it.
// MS (round 4, the model-split shape of small batches): the workgroup's four waves hold the SAME tile and every fourth model of
// the block each; the columns' fate is decided by all of them together (through `s_bad` in LDS), wave 0 reports it.
template <bool MS = false>
__device__ __forceinline__ void h2s_close_block(const H2sArgs &a, const SharedBlock &sb, int blk, const float (&ssum)[SHARED_SB], float off,
                                                bool valid, bool has, int tile_id, int64_t row, int lane, int hh, float safe_ll2,
                                                int wave = 0, int *s_bad = nullptr) {
    constexpr int SB = SHARED_SB;
    bool bad = false;
    float ll_keep[SB];
#pragma unroll
    for (int si = 0; si < SB; si++) {
        if (MS && (si & 3) != wave) continue;
        const float tot = ssum[si] + other_half(ssum[si]);
        const float ll2 = off + log2f(tot);
        ll_keep[si] = LSE_LN2 * ll2;
        const bool ok = tot >= H2S_SUM_LO && tot <= H2S_SUM_HI && ll2 >= safe_ll2;
        bad |= si < sb.n_models && !ok;
    }
    bad = valid && (bad || a.force_exc);
    // the frames (columns) of this tile that go to the exception pass: both half-waves hold the same columns and agree on them
    unsigned bad_cols = (unsigned)__builtin_amdgcn_ballot_w64(bad);           // wave-uniform
    if constexpr (MS) {
        if (bad_cols != 0 && lane == 0) atomicOr(s_bad, (int)bad_cols);
        __syncthreads();
        bad_cols = (unsigned)*reinterpret_cast<volatile int *>(s_bad);
        __syncthreads();                               // everybody has read it
        if (threadIdx.x == 0) *s_bad = 0;              // (for the next block: published by the barrier at its top)
        bad = (bad_cols >> (lane & 31)) & 1u;
    }
    if (bad_cols != 0 && lane == 0 && has && (!MS || wave == 0)) {
        const int idx = atomicAdd(a.exc_count + blk, 1);          // (a tile meets a block once: idx < n_tiles)
        a.exc_list[(size_t)blk * a.n_tiles + idx] = make_int2(tile_id, (int)bad_cols);
    }
    // the other frames' values stand: one partial per model, a fixed-order float64 sum over the wave's lanes (the exception pass
    // adds the listed frames' values to it)
#pragma unroll
    for (int si = 0; si < SB; si++) {
        if (MS && (si & 3) != wave) continue;
        double mine = 0.0;
        if (valid && !bad && hh == 0 && si < sb.n_models) {
            mine = (double)ll_keep[si];
            if (a.frame_ll) a.frame_ll[(int64_t)(sb.first_model + si) * a.n_frames + row] = ll_keep[si];
        }
        mine = wave_sum_f64(mine);
        if (lane == 0 && has && si < sb.n_models)
            a.partial[(int64_t)tile_id * a.n_models + sb.first_model + si] = mine;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Work item `unit` of the pipelined kernel -- a full 32-frame tile, or up to four ragged TAIL tiles of different utterances side by
// side in the wave's 32 columns (ensure_work_table; the table sits right behind the tile table in the same buffer, padded with empty
// items to the grid's size: no kernel argument of its own -- the kernel has no scalar register to spare): its canonical tiles
// tids[0 .. n_seg - 1] and the column each starts at (absent ones: 64).  An empty item has tids[0] < 0.
__device__ __forceinline__ void h2p_unit(const H2sArgs &a, int unit, int (&tids)[4], int (&first_col)[4], int &n_seg) {
    const int4 wk = reinterpret_cast<const int4 *>(a.tiles + a.n_tiles)[unit];
    tids[0] = wk.x; tids[1] = wk.y; tids[2] = wk.z; tids[3] = wk.w;
    first_col[0] = 0; first_col[1] = 64; first_col[2] = 64; first_col[3] = 64;
    n_seg = 1;
    int cum = 0;
#pragma unroll
    for (int p = 1; p < 4; p++)
        if (tids[p] >= 0) {                                    // wave-uniform
            cum += a.tiles[tids[p - 1]].count;
            first_col[p] = cum;
            n_seg = p + 1;
        }
}

// The same close for a wave whose 32 columns hold up to four ragged tail tiles of DIFFERENT utterances (the pipelined kernel's packed
// work items): every segment is one canonical tile of the batch's tile table -- its fate (results or exception list) is decided on
// its own frames only, and its float64 partial is formed exactly as if it sat alone in a wave: the segment's values are moved down to
// lane 0 .. count-1 (zeros elsewhere) before the fixed-order wave sum.  An utterance's result does not depend on what it was packed with.
__device__ __forceinline__ void h2s_close_block_packed(const H2sArgs &a, const SharedBlock &sb, int blk, const float (&ssum)[SHARED_SB], float off,
                                                       bool valid, bool has, const int (&tids)[4], const int (&first_col)[4], int n_seg,
                                                       int64_t row, int lane, int hh, float safe_ll2) {
    constexpr int SB = SHARED_SB;
    const int col = lane & 31;
    const int seg = (col >= first_col[1]) + (col >= first_col[2]) + (col >= first_col[3]);      // (absent segments start at column 64)
    bool bad = false;
    float ll_keep[SB];
#pragma unroll
    for (int si = 0; si < SB; si++) {
        const float tot = ssum[si] + other_half(ssum[si]);
        const float ll2 = off + log2f(tot);
        ll_keep[si] = LSE_LN2 * ll2;
        const bool ok = tot >= H2S_SUM_LO && tot <= H2S_SUM_HI && ll2 >= safe_ll2;
        bad |= si < sb.n_models && !ok;
    }
    bad = valid && (bad || a.force_exc);
    const unsigned bad_cols = (unsigned)__builtin_amdgcn_ballot_w64(bad);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        if (p >= n_seg) break;                                          // wave-uniform
        const unsigned seg_cols = (unsigned)__builtin_amdgcn_ballot_w64(seg == p);
        if ((bad_cols & seg_cols) != 0 && lane == 0 && has) {           // the segment's listed frames, as columns of ITS tile
            const int idx = atomicAdd(a.exc_count + blk, 1);
            a.exc_list[(size_t)blk * a.n_tiles + idx] = make_int2(tids[p], (int)((bad_cols & seg_cols) >> first_col[p]));
        }
        const bool mine_seg = valid && !bad && hh == 0 && seg == p;
#pragma unroll
        for (int si = 0; si < SB; si++) {
            double mine = 0.0;
            if (mine_seg && si < sb.n_models) {
                mine = (double)ll_keep[si];
                if (a.frame_ll) a.frame_ll[(int64_t)(sb.first_model + si) * a.n_frames + row] = ll_keep[si];
            }
            mine = lane_shift_down_f64(mine, first_col[p], lane);
            mine = wave_sum_f64(mine);
            if (lane == 0 && has && si < sb.n_models)
                a.partial[(int64_t)tids[p] * a.n_models + sb.first_model + si] = mine;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Workgroup shapes: waves per workgroup x 32-frame column tiles per wave x images per LDS stage.
//   <4,1>   three workgroups per CU, each with its own copy of the stream (small batches)
//   <12,1>  one workgroup per CU, three waves per SIMD sharing one copy
// Tried and measured slower than <4,1> (profiles/r02_h2s_stalls.txt; the code is in commits 401bc2d and f5a34fe):
// two column tiles per wave (COLS = 2: half the LDS fragment reads, 2 waves per SIMD) -4 %; 8 waves x 2 column
// tiles -5 %; 8 waves with the two waves of a SIMD held in anti-phase by a workgroup barrier per phase (one
// chains while the other runs its epilogue) -3 % with one image per phase, -5 % with two.
__host__ __device__ constexpr int h2s_waves_per_eu(int kqf, int klf, int cols, int waves, bool ms = false) {
    // The 4-wave shape at three workgroups per CU (168 registers) kept its quadratic-half frame fragments in scratch from
    // kqf + klf = 10 up (156 .. 364 bytes per lane, one reload inside the image loop = an s_waitcnt vmcnt(0) on the LDS-DMA
    // stream).  Since round 4 it only serves batches below ~2000 frames (score_device: everything larger takes a 12-wave
    // shape) -- a handful of workgroups, latency-bound, where a third workgroup per CU buys nothing: two per CU, 256 registers,
    // nothing in scratch.
    // (the model-split shape carries a few registers more: at three workgroups per CU its 3 + 3 and 4 + 4 forms spilled 24 / 52 bytes)
    // (since the 4-wave shapes stream four images per stage their LDS admits two workgroups per CU at most, and their batches --
    // below ~2000 frames -- never need more: two per CU for every chain length; the third one's 168-register budget was
    // what spilled, last in <4,4> once the exception lists became per block)
    (void)kqf; (void)klf; (void)cols; (void)ms;
    return waves > 4 ? waves / 4 : 2;
}
// the quadratic-half frame fragments in LDS instead of registers: every 12-wave shape (round 3), and the 4-wave shape of the long
// chains (round 4: with them in registers it spilled 76 bytes per lane even at 256 registers)
__host__ __device__ constexpr bool h2s_bq_in_lds(int kqf, int klf, int waves) { return waves > 4 || kqf + klf >= 9; }
__host__ __device__ constexpr int h2s_stage_images(int kqf, int klf, int waves) {
    // 4-wave form: 2 images per stage (three workgroups per CU share the LDS); 12-wave form: 4 (measured on the
    // configs[2]-shaped 5 M-frame pass: 8 images 0.145 s, 4 images 0.137 s, 2 images 0.138 s)
    return waves == 4 ? 2 : 4;
}

// BQ_LDS (round 3): the wave's quadratic-half B fragments -- used once per 15 models, 8 x 16 bytes per lane at D = 39 -- live in
// LDS (dynamic, WAVES x KQF KiB) instead of the registers they never fitted: in round 2 the compiler kept them in scratch
// (344 bytes per lane with the prologue's temporaries: 7 GB of scratch writes per configs[2] pass, profiles/r02b_pmc.txt).
// Taken for the 12-wave shape (one workgroup per CU: the LDS is there); the 4-wave shape shares a CU's LDS three ways.
// Same speed as the scratch form to 0.5 % (profiles/r03_scoring_experiments.txt), bit-identical results, 142 VGPRs, no scratch.
// MS (round 4): the 4-wave shape for the smallest batches -- one serving utterance is ten 32-frame tiles: 42 workgroups of the
// plain shape, each wave alone with a chain of 2048 MFMAs (93 us).  Here a workgroup holds ONE tile and its four waves split the
// block's models (wave w: models w, w + 4, ...; every wave forms the quadratic half itself: 19 images' worth of MFMAs per
// mixture tile instead of 16): four times the workgroups, chains a third as long, same results bit for bit (a (frame, model)
// sum is formed by one lane in the same order either way).
template <int KQF, int KLF, int COLS, int WAVES, bool MS = false>
__global__ __launch_bounds__(WAVES * 64, h2s_waves_per_eu(KQF, KLF, COLS, WAVES, MS))
void gmm_score_h2s_kernel(const H2sArgs a) {
    static_assert(!MS || (WAVES == 4 && COLS == 1), "the model-split shape: four waves, one tile");
    constexpr int SB = SHARED_SB;
    constexpr bool BQ_LDS = h2s_bq_in_lds(KQF, KLF, WAVES);
    constexpr int Q_U4 = KQF * 64, L_U4 = KLF * 64;
    constexpr int IMG_U4 = Q_U4 > L_U4 ? Q_U4 : L_U4;          // every image padded to the larger of the two
    // images per LDS stage.  (2 measured the same as 4 for the 12-wave shape.)  The 4-wave shapes serve small batches only (round 4:
    // below ~2000 frames, at most a workgroup per CU): their few workgroups stream COLD images -- no neighbour on the XCD has
    // fetched them -- and wait out a trip to HBM per stage: four images per stage where two such buffers fit the 64 KiB of
    // static LDS (300 frames, model-split shape: 0.094 -> 0.081 ms).
    constexpr int G = WAVES == 4 ? (2 * 4 * IMG_U4 * (int)sizeof(uint4) <= 65536 ? 4 : 2) : BQ_LDS ? 2 : h2s_stage_images(KQF, KLF, WAVES);
    extern __shared__ uint4 h2s_bq_lds[];                                  // [WAVES][KQF][64] when BQ_LDS; (MS) one more for s_bad
    constexpr int TILES_WG = MS ? 1 : WAVES * COLS;            // 32-frame tiles per workgroup
    constexpr int STRIDE_U4 = (1 + SB) * IMG_U4;               // one mixture tile of one block
    constexpr int N_STAGES = (1 + SB) / G;
    static_assert((1 + SB) % G == 0 && (N_STAGES % 2) == 0, "stages must tile the 16 images and alternate buffers");
    __shared__ uint4 lds_a[G * IMG_U4];
    __shared__ uint4 lds_b[G * IMG_U4];
    // (MS) "a wave's models sent the tile to the exception list": behind the dynamic region (the static one may be full)
    int *const s_bad = reinterpret_cast<int *>(h2s_bq_lds + (BQ_LDS ? WAVES * KQF * 64 : 0));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;
    const int hh = lane >> 5;
    if (MS && tid == 0) *s_bad = 0;                            // (published by the barrier at the top of the first block)

    // a stage is N_PIECES wave-instructions of 1 KiB; wave w issues pieces w, w + WAVES, ... (a scalar test)
    constexpr int N_PIECES = G * IMG_U4 / 64;
    // every wave issues its share of a stage (in a 12-wave workgroup: all twelve 0.136 s on the configs[2]-shaped
    // 5 M-frame pass, only the first four 0.140 s, only the last four 0.152 s -- a piece costs its issuing wave
    // ~110 cycles, and the waves of a SIMD reach a stage barrier far apart; profiles/r02_h2s_stalls.txt)
    auto stage_load = [&](uint4 *dst, const uint4 *src) {
#pragma unroll
        for (int i = 0; i < (N_PIECES + WAVES - 1) / WAVES; i++) {
            const int piece = i * WAVES + wave;
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

    int g, tile0;                                  // this workgroup's model group and first 32-frame tile
    if (!h2s_wg_assignment(a, TILES_WG, g, tile0)) return;
    const int blk_begin = a.group_block_begin[g];
    const int blk_end = a.group_block_begin[g + 1];

    // ---- resident B fragments of this lane's frame(s) ----
    f16x8 bq[COLS][KQF], bl[COLS][KLF];
    bool valid[COLS], has[COLS];
    int tile_id[COLS];
    int64_t row[COLS];
    float off[COLS];                                       // per-frame offset O (log2 units)
    float zmax = 0.0f;
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        tile_id[c] = tile0 + (MS ? 0 : wave * COLS + c);
        has[c] = tile_id[c] < a.n_tiles;
        const TileDesc tile = a.tiles[has[c] ? tile_id[c] : a.n_tiles - 1];
        valid[c] = has[c] && col < tile.count;
        row[c] = tile.start + (valid[c] ? col : 0);
        h2s_build_b<KQF>(bq[c], a.X + row[c] * a.dim, a.center, a.scale, a.q_desc, hh, true, zmax);
        if constexpr (BQ_LDS) {
            static_assert(COLS == 1, "one column tile per wave");
#pragma unroll
            for (int ks = 0; ks < KQF; ks++) h2s_bq_lds[(wave * KQF + ks) * 64 + lane] = __builtin_bit_cast(uint4, bq[c][ks]);
        }
        h2s_build_b<KLF>(bl[c], a.X + row[c] * a.dim, a.center, a.scale, a.l_desc, hh, false, zmax);
        off[c] = a.ref_ll[row[c]] * H2S_LOG2E;
    }
    if (zmax >= 255.0f) atomicOr(a.oor_flag, 1);           // saturated: the host re-scores on the fp32-grade engines
    // the largest term is >= LL - log2 K; below this the online pass decides
    // (and everything below the band in which the reference's partial-product flushes can decide, lse.hpp)
    const float safe_ll2 = a.clamp ? fmaxf(LSE_MINLOG2 + LSE_NEAR + a.log2_k, a.band_hi * H2S_LOG2E + 1.0f) : -3.0e38f;

    const f32x16 zero1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 zero16[COLS];
#pragma unroll
    for (int c = 0; c < COLS; c++) zero16[c] = zero1;
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
        float ssum[COLS][SB];
#pragma unroll
        for (int c = 0; c < COLS; c++)
#pragma unroll
            for (int si = 0; si < SB; si++) ssum[c][si] = 0.0f;
        // Per image: (1) the chain(s) -- the KN A fragments already sit in registers, so the MFMAs
        // issue back to back; (2) the NEXT image's fragments are requested from LDS; (3) this image's
        // epilogue runs while they arrive.  A stage's LDS buffer is free as soon as the fragments of
        // its last image are in registers, so the barrier (and the LDS-DMA of the stage after next
        // into the freed buffer) comes before that image's epilogue, not after it.
        constexpr int KM = KQF > KLF ? KQF : KLF;
        uint4 fr[KM];
        auto load_frags = [&](const uint4 *at, int kn) {
#pragma unroll
            for (int ks = 0; ks < KM; ks++)
                if (ks < kn) fr[ks] = at[ks * 64];
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
            f32x16 qacc[COLS];
#pragma unroll
            for (int st = 0; st < N_STAGES; st++) {
                uint4 *cur = (st & 1) ? lds_b : lds_a;
                const uint4 *nxt = (st & 1) ? lds_a : lds_b;
#pragma unroll
                for (int gi = 0; gi < G; gi++) {
                    const int img = st * G + gi;
                    f32x16 acc[COLS];
                    if (img == 0) {
                        if constexpr (BQ_LDS) {
                            f16x8 bqt[COLS][KQF];                 // this wave's own slab: written once in the prologue (same lanes)
#pragma unroll
                            for (int ks = 0; ks < KQF; ks++)
                                bqt[0][ks] = __builtin_bit_cast(f16x8, h2s_bq_lds[(wave * KQF + ks) * 64 + lane]);
                            h2s_chain_regs<KQF, KM, COLS>(qacc, zero16, fr, bqt);
                        } else {
                            h2s_chain_regs<KQF, KM, COLS>(qacc, zero16, fr, bq);
                        }
                    } else if (!MS || ((img - 1) & 3) == wave)
                        h2s_chain_regs<KLF, KM, COLS>(acc, qacc, fr, bl);
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
                        for (int c = 0; c < COLS; c++)
#pragma unroll
                            for (int r = 0; r < 16; r++) qacc[c][r] -= off[c];
                    } else if (!MS || ((img - 1) & 3) == wave) {
#pragma unroll
                        for (int c = 0; c < COLS; c++) {
                            // (Tried in round 3 and lost, profiles/r03_h2s_experiments.txt: skipping the 16 exponentials of a
                            // model-tile whose 64 x 16 terms are all below 2^-36 of the frame's offset -- 8 v_max3 + a compare +
                            // a wave-uniform branch.  7-9 % SLOWER on every workload: with 32 different frames per wave tile
                            // almost every tile holds a term that matters for some lane; and unsafe as it stood, the offset
                            // being the UBM's value, not the model's own.)
                            float e0 = 0.0f, e1 = 0.0f;
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                e0 += __builtin_amdgcn_exp2f(acc[c][r]);
                                e1 += __builtin_amdgcn_exp2f(acc[c][r + 1]);
                            }
                            ssum[c][img - 1] += e0 + e1;
                            asm volatile("" : "+v"(ssum[c][img - 1]));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- close the block's models: per 32-frame tile, so an utterance's fate does not depend on its neighbours ----
#pragma unroll
        for (int c = 0; c < COLS; c++)
            h2s_close_block<MS>(a, sb, blk, ssum[c], off[c], valid[c], has[c], tile_id[c], row[c], lane, hh, safe_ll2, wave, s_bad);
    }
}

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void h2p_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        h2p_for<I + 1, N>(f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gmm_score_h2m_kernel (round 6): the model-split shape of the smallest batches -- a serving decision is ten 32-frame tiles --
// with every wave fetching ITS images straight into registers, two images ahead, instead of the workgroup staging all sixteen
// through LDS.  What the LDS form (round 4: gmm_score_h2s_kernel<.., MS>) measured: 62 us of a 131 us decision, 140 workgroups
// streaming 2 MB each -- and a two-tile workgroup that streams half as much per frame was SLOWER (0.109 against 0.080 ms): the
// kernel waits out a round trip to the L2 / fabric per stage (64 stages per block, ~1 us each, one 32 KB stage in flight per
// CU), not its bytes.  A wave needs only the quadratic image and its quarter of the fifteen linear ones (4.75 of 16 per
// mixture tile); nothing but the quadratic image is shared, so the LDS bought one image in five and cost every wave the reads
// of all sixteen plus 64 barriers.  Here: no LDS, no barriers in the image loop, two register sets of KM fragments refilled as
// soon as their chain has issued (the compiler places the vmcnt waits: plain loads into registers), 2 x 4 waves x 8 KB in flight
// per CU.  Same arithmetic in the same order as the LDS form (a (frame, model) value is formed by one lane over the mixture
// tiles in order): same bits.
constexpr int H2M_MAX_KLF = 9;        // two register sets of fragments fit up to here (D <= 45); the longest chains keep the LDS form
template <int KQF, int KLF>
__global__ __launch_bounds__(256, 2)
void gmm_score_h2m_kernel(const H2sArgs a) {
    constexpr int SB = SHARED_SB;
    constexpr int KM = KQF > KLF ? KQF : KLF;
    constexpr int IMG_U4 = KM * 64;
    constexpr int STRIDE_U4 = (1 + SB) * IMG_U4;
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;
    const int hh = lane >> 5;
    if (tid == 0) s_bad = 0;                                   // (published by the barrier of the first close)

    int g, tile0;
    if (!h2s_wg_assignment(a, 1, g, tile0)) return;
    const int blk_begin = a.group_block_begin[g];
    const int blk_end = a.group_block_begin[g + 1];
    const bool has = tile0 < a.n_tiles;
    const TileDesc tile = a.tiles[has ? tile0 : a.n_tiles - 1];
    const bool valid = has && col < tile.count;
    const int64_t row = tile.start + (valid ? col : 0);
    // the quadratic-half frame fragments: the same for the four waves (one tile), used once per 4-5 images -- in LDS (KQF KiB),
    // written by wave 0, not in 4 x KQF registers per lane (with them the kernel spilled 7 dwords: a reload in the image loop
    // waits for the prefetches in flight)
    __shared__ uint4 s_bq[KQF * 64];
    f16x8 bl[KLF];
    float zmax = 0.0f;
    {
        f16x8 bq[KQF];
        h2s_build_b<KQF>(bq, a.X + row * a.dim, a.center, a.scale, a.q_desc, hh, true, zmax);
        if (wave == 0) {
#pragma unroll
            for (int ks = 0; ks < KQF; ks++) s_bq[ks * 64 + lane] = __builtin_bit_cast(uint4, bq[ks]);
        }
    }
    h2s_build_b<KLF>(bl, a.X + row * a.dim, a.center, a.scale, a.l_desc, hh, false, zmax);
    __syncthreads();
    const float off = a.ref_ll[row] * H2S_LOG2E;
    if (zmax >= 255.0f) atomicOr(a.oor_flag, 1);
    const float safe_ll2 = a.clamp ? fmaxf(LSE_MINLOG2 + LSE_NEAR + a.log2_k, a.band_hi * H2S_LOG2E + 1.0f) : -3.0e38f;
    const f32x16 zero1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    for (int blk = blk_begin; blk < blk_end; blk++) {
        const SharedBlock sb = a.blocks[blk];
        const uint4 *base = a.params + sb.offset_u4 + lane;
        float ssum4[4] = {0.0f, 0.0f, 0.0f, 0.0f};             // this wave's models wave, wave + 4, wave + 8, wave + 12
        uint4 fr[2][KM];
        f32x16 qacc = zero1;
        // image j of a mixture tile for this wave: 0 = the quadratic one, j >= 1 = linear image of model wave + 4 (j - 1)
        auto img_at = [&](int t, int j) { return base + (size_t)t * STRIDE_U4 + (size_t)(j == 0 ? 0 : 1 + wave + 4 * (j - 1)) * IMG_U4; };
        auto load = [&](uint4 (&dst)[KM], const uint4 *at, int kn) {
#pragma unroll
            for (int ks = 0; ks < KM; ks++)
                if (ks < kn) dst[ks] = at[ks * 64];
        };
        // PT images per mixture tile (5 for waves 0-2, 4 for wave 3), NT tiles per call: PT * NT steps, step u on register set u & 1
        // (two tiles of five make an even count, so a call always starts on set 0), set u & 1 refilled with step u + 2's image
        auto run = [&](auto pt_c, auto nt_c, int t0) __attribute__((always_inline)) {
            constexpr int PT = decltype(pt_c)::value, NT = decltype(nt_c)::value;
            h2p_for<0, PT * NT>([&](auto uc) {
                constexpr int U = decltype(uc)::value, TT = U / PT, J = U % PT, SET = U & 1;
                f32x16 acc;
                if constexpr (J == 0) {
                    acc = zero1;
#pragma unroll
                    for (int ks = 0; ks < KQF; ks++)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[SET][ks]), __builtin_bit_cast(f16x8, s_bq[ks * 64 + lane]), acc, 0, 0, 0);
                } else {
                    acc = qacc;
#pragma unroll
                    for (int ks = 0; ks < KLF; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[SET][ks]), bl[ks], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // the image two steps on, into the set whose chain has just issued
                    constexpr int U2 = U + 2;
                    if constexpr (U2 < PT * NT) {
                        load(fr[SET], img_at(t0 + U2 / PT, U2 % PT), (U2 % PT) == 0 ? KQF : KLF);
                    } else {
                        constexpr int J2 = U2 - PT * NT;               // 0 or 1: of the tile after this call's
                        if (t0 + NT < a.n_mix_tiles) load(fr[SET], img_at(t0 + NT, J2), J2 == 0 ? KQF : KLF);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (J == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] -= off;
                    qacc = acc;
                } else {
                    float e0 = 0.0f, e1 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        e0 += __builtin_amdgcn_exp2f(acc[r]);
                        e1 += __builtin_amdgcn_exp2f(acc[r + 1]);
                    }
                    ssum4[J - 1] += e0 + e1;
                    asm volatile("" : "+v"(ssum4[J - 1]));
                }
                __builtin_amdgcn_sched_barrier(0);
                (void)TT;
            });
        };
        load(fr[0], img_at(0, 0), KQF);
        load(fr[1], img_at(0, 1), KLF);
        int t = 0;
        if (wave < 3) {
            for (; t + 2 <= a.n_mix_tiles; t += 2) run(std::integral_constant<int, 5>{}, std::integral_constant<int, 2>{}, t);
            if (t < a.n_mix_tiles) run(std::integral_constant<int, 5>{}, std::integral_constant<int, 1>{}, t);
        } else {
            for (; t + 2 <= a.n_mix_tiles; t += 2) run(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, t);
            if (t < a.n_mix_tiles) run(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}, t);
        }
        float ssum[SB];
#pragma unroll
        for (int si = 0; si < SB; si++) ssum[si] = ssum4[si >> 2];          // (its own models: si = wave + 4 k; the others are not read)
        h2s_close_block<true>(a, sb, blk, ssum, off, valid, has, tile0, row, lane, hh, safe_ll2, wave, &s_bad);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gmm_score_h2p_kernel (round 3): the 12-wave form with the image loop software-pipelined INSIDE each wave.  What the
// dispatcher takes for large batches (score_h2s_shape = 3 forces it): 6.5-8.5 % faster than the plain 12-wave kernel above.
//
// Round 2 read scripts/ubench/mfma_lse_inwave.hip as "a wave cannot hide its own vector work behind its own MFMAs" and built
// the loop above on it (chain, fragment reads, epilogue, one after the other).  That microbenchmark left the interleave to
// sched_group_barrier hints which the compiler did not follow (5 MFMAs back to back, the exps in clumps).  With the order
// PINNED (scripts/ubench/mfma_lse_pinned.hip, profiles/r03_ubench_pinned.txt) one wave runs 8 MFMAs + 16 exp + 16 add in 330
// cycles (466 one after the other), three waves per SIMD in 286: an MFMA occupies the matrix pipe for 32 cycles and the same
// wave's independent vector instructions issue in its shadow.  (v_pk_add_f32 does NOT hide: 518 cycles.  Plain adds only.)
//
// What that bought on the real workload, and why the schedule alone is worth so little (profiles/r03_h2p_parts.txt): 1-2 %.
// With its parts switched off one at a time the kernel says where the time is: matrix instructions alone 27.9 ms (1.5 M frames)
// = the pipe at ~98 % -- at the clock the 1.3 kW cap leaves it (~1.6 GHz, not 2.4; sr_mfma_peak_probe measures 1.75-1.8 PFLOP/s
// sustained) -- and every other part adds about what it costs in ENERGY whether or not it overlaps in cycles: the 16 exps +
// 16 adds +5.7 ms (hidden in MFMA shadow or not), the 8 fragment reads per image +2.0, the LDS-DMA stream +1.9, the stage
// barriers +1.5.  The chip trades clock for activity.  The rest of the 6.5-8.5 % came from what the pinned structure made
// affordable: half the barriers (a ring of 2 x 4 images: with the run-ahead DMA the lead stays 4 images), the LDS-DMA issued
// in the middle of an image instead of in front of its first MFMA, no scratch at all (one reload in the loop = one vmcnt(0) =
// a wait for the whole stream in flight), and the stages that hold only phantom models of a set's last block skipped (a scalar
// test per image: this loop's order is pinned, the same test cost the round-2 loop 9 %).  The matrix work itself cannot shrink
// inside the 1e-4 tolerance (one part product instead of three on the MAP shift: 4.7e-4 on the benchmark's own speakers; fp8
// cross terms: ~1e-4; HISTORY.md 2.1).
//
// Slot u of image i (one per MFMA of the chain):
//     s_waitcnt lgkmcnt  fragment u is here (counting only this loop's reads: safe beside the compiler's own, returns are in order)
//     v_mfma             acc[i & 1] <- fr[u] x bl[u] (+ Q for u = 0)
//     ds_read_b128       fr[u] <- fragment u of image i + 1              (the register the MFMA just read)
//     v_add_f32 x n      the exps of slot u - 1 into the running sum of image i - 1's model   (a slot late: gfx950 needs a wait
//     v_exp_f32 x n      of acc[(i - 1) & 1], n = 2-4; none in slot 0     state between a transcendental and its consumer)
// as ONE asm statement per slot for chains of 5+ MFMAs (statement by statement hipcc puts an s_nop between any two that share
// a register), operation by operation for shorter chains.  sched_barrier fences each slot: the IR-level vectoriser had turned
// builtin-based adds into v_pk_add_f32 and sunk them, the post-RA scheduler had clumped the exps.  The fragment reads are asm
// with their own lgkmcnt bookkeeping because a compiler-visible ds_read that may alias an LDS-DMA target makes hipcc drain
// vmcnt in front of it, and the stream here runs two stages ahead through a ring of three.  Hazards the compiler no longer
// sees, handled by hand: MFMA result -> VALU read (slot 0 carries no exps, so a full MFMA lies between the last MFMA of an image
// and the first exp of its values; s_nop 12 in front of the Q - O subtraction and of the final drain), VALU write -> MFMA C
// operand (s_nop 1 after the subtraction), transcendental -> consumer (the adds trail by a slot; s_nop 0 before the last ones).
//
// Barrier B_S sits in front of image (S, 1): stage S + 1 has landed (each wave waits for its own pieces, then the barrier), the
// fragments of (S + 1, 0) are read during (S, 1); the slot of stage S is free (its last reads were issued during (S, 0)) and takes
// the LDS-DMA of stage S + 3.
// exps of a 16-value epilogue issued in slots <= u of a KN-slot chain (none in slot 0; all after the chain when KN = 1)
__host__ __device__ constexpr int h2p_cum(int kn, int u) { return kn <= 1 ? 0 : u <= 0 ? 0 : u >= kn - 1 ? 16 : (16 * u + (kn - 1) / 2) / (kn - 1); }
// (every exp has a slot, the counts never decrease, and a chain of 5+ MFMAs never asks a slot for more than the 4 that the
// one-statement slot form has operands for)
constexpr bool h2p_cum_ok(int kn) {
    if (h2p_cum(kn, 0) != 0 || h2p_cum(kn, kn - 1) != 16) return false;
    for (int u = 1; u < kn; u++) {
        const int n = h2p_cum(kn, u) - h2p_cum(kn, u - 1);
        if (n < 0 || (kn >= 5 && n > 4)) return false;
    }
    return true;
}
static_assert(h2p_cum_ok(2) && h2p_cum_ok(3) && h2p_cum_ok(4) && h2p_cum_ok(5) && h2p_cum_ok(6) && h2p_cum_ok(7) && h2p_cum_ok(8),
              "distribution of a model-tile's 16 exps over the slots of the next chain");

// One slot of the pipelined loop as ONE asm statement (chains of 5+ MFMAs: at most 4 exps per slot).  Statement by statement
// hipcc puts an s_nop between any two asm statements that touch the same register and after each wait in front of the MFMA:
// 11-13 instructions per slot, and a wave issues about one instruction per 4-5 cycles -- more than the 32 cycles of MFMA
// shadow the slot has (measured: the epilogue cost 20 % on top of the bare chains in that form).
//   HEAD 0: first MFMA of the Q image (C = 0)   1: first MFMA of an L image (C = Q)   2: any later one (C = the accumulator)
#define H2P_ADDS0 ""
#define H2P_ADDS1 "v_add_f32 %[s], %[s], %[e0]\n"
#define H2P_ADDS2 H2P_ADDS1 "v_add_f32 %[s], %[s], %[e1]\n"
#define H2P_ADDS3 H2P_ADDS2 "v_add_f32 %[s], %[s], %[e2]\n"
#define H2P_ADDS4 H2P_ADDS3 "v_add_f32 %[s], %[s], %[e3]\n"
#define H2P_EXPS0 ""
#define H2P_EXPS1 "v_exp_f32 %[e0], %[p0]\n"
#define H2P_EXPS2 H2P_EXPS1 "v_exp_f32 %[e1], %[p1]\n"
#define H2P_EXPS3 H2P_EXPS2 "v_exp_f32 %[e2], %[p2]\n"
#define H2P_EXPS4 H2P_EXPS3 "v_exp_f32 %[e3], %[p3]\n"
#define H2P_WAIT "s_waitcnt lgkmcnt(%[w])\n"
#define H2P_MFMA0 "v_mfma_f32_32x32x16_f16 %[cur], %[fu], %[b], 0\n"
#define H2P_MFMA1 "v_mfma_f32_32x32x16_f16 %[cur], %[fu], %[b], %[c]\n"
#define H2P_MFMA2 "v_mfma_f32_32x32x16_f16 %[cur], %[fu], %[b], %[cur]\n"
#define H2P_READ0 ""
#define H2P_READ1 "ds_read_b128 %[fu], %[at] offset:%[o]\n"
#define H2P_SLOT_ASM(H, R, NA_, NX_)                                                                                                     \
    asm volatile(H2P_WAIT H2P_MFMA##H H2P_READ##R H2P_ADDS##NA_ H2P_EXPS##NX_                                                            \
                 : [cur] "+v"(cur), [fu] "+v"(fu), [s] "+v"(s), [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [e3] "+v"(e3)                \
                 : [b] "v"(b), [c] "v"(c), [at] "v"(at), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [w] "n"(W), [o] "n"(O))
#define H2P_SLOT_NX(H, R, NA_)                                  \
    if constexpr (NX == 0) H2P_SLOT_ASM(H, R, NA_, 0);          \
    else if constexpr (NX == 1) H2P_SLOT_ASM(H, R, NA_, 1);     \
    else if constexpr (NX == 2) H2P_SLOT_ASM(H, R, NA_, 2);     \
    else if constexpr (NX == 3) H2P_SLOT_ASM(H, R, NA_, 3);     \
    else H2P_SLOT_ASM(H, R, NA_, 4);
#define H2P_SLOT_NA(H, R)                                \
    if constexpr (NA == 0) { H2P_SLOT_NX(H, R, 0) }      \
    else if constexpr (NA == 1) { H2P_SLOT_NX(H, R, 1) } \
    else if constexpr (NA == 2) { H2P_SLOT_NX(H, R, 2) } \
    else if constexpr (NA == 3) { H2P_SLOT_NX(H, R, 3) } \
    else { H2P_SLOT_NX(H, R, 4) }
template <int HEAD, bool READ, int NA, int NX, int W, int O>
__device__ __forceinline__ void h2p_slot(f32x16 &cur, u32x4 &fu, float &s, float &e0, float &e1, float &e2, float &e3, const f16x8 &b, const f32x16 &c,
                                         unsigned at, float p0, float p1, float p2, float p3) {
    static_assert(NA <= 4 && NX <= 4, "at most four exps per slot");
    if constexpr (HEAD == 0) {
        if constexpr (READ) { H2P_SLOT_NA(0, 1) } else { H2P_SLOT_NA(0, 0) }
    } else if constexpr (HEAD == 1) {
        if constexpr (READ) { H2P_SLOT_NA(1, 1) } else { H2P_SLOT_NA(1, 0) }
    } else {
        if constexpr (READ) { H2P_SLOT_NA(2, 1) } else { H2P_SLOT_NA(2, 0) }
    }
}

template <int KQF, int KLF>
__global__ __launch_bounds__(12 * 64, 3)
void gmm_score_h2p_kernel(const H2sArgs a) {
    // G images per stage, a ring of NB stages, the LDS-DMA of a stage issued in slot DMA_AT of the image behind the barrier.
    // Measured on the configs[2]-shaped pass (profiles/r03_h2p_parts.txt): ring 3 x 2 images with the DMA right behind the barrier
    // 0.988 of the round-2 kernel's time; DMA in slot 4 0.973-0.977; ring 2 x 4 images (half the barriers, the same lead of 4 images,
    // 64 KiB + the 96 KiB of quadratic-half fragments = all 160 KiB of the CU) 0.945.
    constexpr int SB = SHARED_SB, WAVES = 12, G = 4, NB = 2, DMA_AT = KLF / 2;
    constexpr int KM = KQF > KLF ? KQF : KLF;
    static_assert(KQF <= KLF, "the linear half is never the shorter one");
    constexpr int IMG_U4 = KM * 64;
    constexpr int STAGE_U4 = G * IMG_U4;
    constexpr int N_IMG = 1 + SB;
    constexpr int N_STAGES = N_IMG / G;
    static_assert(N_IMG % G == 0 && N_STAGES >= 1, "stages tile the images of a mixture tile");
    extern __shared__ uint4 h2s_bq_lds[];                                  // [WAVES][KQF][64]
    __shared__ uint4 ring[NB * STAGE_U4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;
    const int hh = lane >> 5;

    constexpr int N_PIECES = STAGE_U4 / 64;                    // 1 KiB wave-instructions per stage
    constexpr int P_LO = N_PIECES / WAVES, P_HI = (N_PIECES + WAVES - 1) / WAVES, N_HI = N_PIECES % WAVES;
    auto stage_load = [&](int slot, const uint4 *src) {
#pragma unroll
        for (int i = 0; i < P_HI; i++) {
            const int piece = i * WAVES + wave;
            if (piece < N_PIECES)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + piece * 64 + lane),
                    (__attribute__((address_space(3))) void *)(ring + slot * STAGE_U4 + piece * 64), 16, 0, 0);
        }
    };
    // wait until at most `keep` stages' worth of THIS wave's pieces are in flight (they land in issue order)
    auto wait_stages = [&](int keep) {
        if (keep <= 0 || P_HI == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (keep == 1) {
            if (N_HI != 0 && wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_HI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_LO) : "memory");
        } else {
            if (N_HI != 0 && wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P_HI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P_LO) : "memory");
        }
    };
    auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };

    int g, tile0;                                  // this workgroup's model group and first 32-frame tile
    if (!h2s_wg_assignment(a, WAVES, g, tile0)) return;
    const int blk_begin = a.group_block_begin[g];
    const int blk_end = a.group_block_begin[g + 1];

    // ---- resident B fragments of this lane's frame ----
    f16x8 bl[KLF];
    float zmax = 0.0f;
    // this wave's work item (h2p_unit).  What describes it is read here and AGAIN at every close: ten more scalar registers held
    // across the image loop spilled 35 of them into vector lanes.
    const int unit = tile0 + wave;
    bool has;
    TileDesc tile;
    int my_col = col;
    {
        int tids[4], first_col[4], n_seg;
        h2p_unit(a, unit, tids, first_col, n_seg);
        has = tids[0] >= 0;
        tile = a.tiles[has ? tids[0] : 0];
#pragma unroll
        for (int p = 1; p < 4; p++)
            if (p < n_seg && col >= first_col[p]) {
                my_col = col - first_col[p];
                tile = a.tiles[tids[p]];
            }
    }
    const bool valid = has && my_col < tile.count;
    const int64_t row = tile.start + (valid ? my_col : 0);
    {
        f16x8 bq[KQF];
        h2s_build_b<KQF>(bq, a.X + row * a.dim, a.center, a.scale, a.q_desc, hh, true, zmax);
#pragma unroll
        for (int ks = 0; ks < KQF; ks++) h2s_bq_lds[(wave * KQF + ks) * 64 + lane] = __builtin_bit_cast(uint4, bq[ks]);
    }
    h2s_build_b<KLF>(bl, a.X + row * a.dim, a.center, a.scale, a.l_desc, hh, false, zmax);
    const float off = a.ref_ll[row] * H2S_LOG2E;
    if (zmax >= 255.0f) atomicOr(a.oor_flag, 1);
    const float safe_ll2 = a.clamp ? fmaxf(LSE_MINLOG2 + LSE_NEAR + a.log2_k, a.band_hi * H2S_LOG2E + 1.0f) : -3.0e38f;
    const f32x16 zero1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned ring_lane = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)ring + (unsigned)lane * 16u;   // LDS byte address

    for (int blk = blk_begin; blk < blk_end; blk++) {
        const SharedBlock sb = a.blocks[blk];
        const uint4 *stream = a.params + sb.offset_u4;
        // ssum[SB]: the LAST executed image of a mixture tile has its epilogue carried by the next tile's Q image (and by the drain
        // behind the loop); which model that is depends on the block (below), so its sum is kept apart until the close
        float ssum[SB + 1];
#pragma unroll
        for (int si = 0; si <= SB; si++) ssum[si] = 0.0f;
        // A block of fewer than SB models (the last one of a set: 201 = 13 x 15 + 6) runs only the stages that hold a model: its
        // images are there in the stream (zeros), but 8 of configs[2]'s 224 images per mixture tile are not worth 3.6 % of the pass.
        // Stage by stage -- a scalar test per image, which this loop can afford (its order is pinned; the same test cost the
        // round-2 loop 9 %, profiles/r02_h2s_stalls.txt section 7).
        const int n_st = (1 + sb.n_models + G - 1) / G;       // stages per mixture tile with a model in them, 1 .. N_STAGES
        const int n_stage_total = a.n_mix_tiles * n_st;
        int ld_tile = 0, ld_st = 0;                           // the next stage to load: mixture tile, stage within it
        auto stage_load_next = [&](int slot) {
            stage_load(slot, stream + (size_t)ld_tile * (N_IMG * IMG_U4) + (size_t)ld_st * STAGE_U4);
            if (++ld_st == n_st) {
                ld_st = 0;
                ld_tile++;
            }
        };
        u32x4 fr[KM];                         // (a native vector: asm operands cannot be HIP's struct vectors)
        // fragment `U` of the image at LDS byte address `at`
        auto frag_read = [&](auto U, unsigned at) {
            constexpr int u = decltype(U)::value;
            u32x4 &dst = fr[u];               // (an asm operand alone does not capture)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(at), "n"(u * 1024));
        };

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads the previous block's last image issued for an image that is not there
        barrier();                                            // previous block's readers are done with the ring
        stage_load_next(0);
        if (n_stage_total > 1) stage_load_next(1);
        if (NB > 2 && n_stage_total > 2) stage_load_next(2);
        wait_stages(n_stage_total > NB - 1 ? NB - 1 : n_stage_total - 1);
        barrier();
        h2p_for<0, KQF>([&](auto U) { frag_read(U, ring_lane); });

        f32x16 acc[2];
        acc[0] = zero1;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[1][r] = -1.0e30f;    // "the image before the first": its epilogue adds 16 zeros to ssum[SB]
        int S = 0, slot_cur = 0, slot_nxt = 1;                // stage counter of the block and its ring slots
        static_assert(NB == 2 || NB == 3, "ring of two or three stages");
        f32x16 qacc = zero1;
        float e[16] = {0.0f, 0.0f, 0.0f, 0.0f};               // exps on their way from the slot that made them to the slot that adds them up
        for (int t = 0; t < a.n_mix_tiles; t++) {
            h2p_for<0, N_IMG>([&](auto IMG) {
                constexpr int img = decltype(IMG)::value;
                constexpr int gi = img % G;
                constexpr int kn = img == 0 ? KQF : KLF;                   // this image's chain
                constexpr int kn_next = img + 1 == N_IMG ? KQF : KLF;      // fragments of the next one
                constexpr bool carry = img != 1;                           // image 1 follows Q: nothing to add up
                constexpr int prev_model = img == 0 ? SB : img - 2;        // sum of the image before (when carry); SB: see ssum
                if (img / G >= n_st) return;                               // a stage of phantom models only
                if constexpr (gi == G - 1) {
                    wait_stages(NB > 2 && S + 2 < n_stage_total ? 1 : 0);
                    barrier();
                    if constexpr (DMA_AT == 0)
                        if (S + NB < n_stage_total) stage_load_next(slot_cur);
                }
                const unsigned next_at = ring_lane + (unsigned)((gi == G - 1 ? slot_nxt * STAGE_U4 : slot_cur * STAGE_U4 + (gi + 1) * IMG_U4) * 16);
                f32x16 &cur = img == 0 ? qacc : acc[img & 1];
                const f32x16 &prev = acc[(img & 1) ^ 1];                   // (image 0 follows image 15: acc[1])
                // the wave's own quadratic-half fragments, four at a time (all KQF of them beside fr, bl and three accumulators do
                // not fit 168 registers: hipcc spilled two bl fragments, and a scratch reload waits with vmcnt(0) -- for the whole
                // LDS-DMA stream in flight, not only for itself)
                constexpr int BW = KQF < 4 ? KQF : 4;
                f16x8 bqt[BW];
                auto bq_load = [&](int ks) { return __builtin_bit_cast(f16x8, h2s_bq_lds[(wave * KQF + ks) * 64 + lane]); };
                if constexpr (img == 0) {
#pragma unroll
                    for (int ks = 0; ks < BW; ks++) bqt[ks] = bq_load(ks);
                }
                constexpr bool ONE_ASM = kn >= 5;              // (shorter chains: more than 4 exps per slot)
                constexpr int sum_model = carry ? prev_model : 0;          // (an operand has to name something)
                if constexpr (ONE_ASM) {
                    h2p_for<0, kn>([&](auto U) {
                        constexpr int u = decltype(U)::value;
                        if constexpr (gi == G - 1 && DMA_AT != 0 && u == DMA_AT)
                            if (S + NB < n_stage_total) stage_load_next(slot_cur);
                        constexpr int younger = (kn - 1 - u) + (u < kn_next ? u : kn_next);
                        static_assert(younger <= 15, "lgkmcnt is a 4-bit field");
                        constexpr int e0 = carry ? h2p_cum(kn, u - 2) : 0, e1 = carry ? h2p_cum(kn, u - 1) : 0, e2 = carry ? h2p_cum(kn, u) : 0;
                        constexpr int NA = e1 - e0, NX = e2 - e1;
                        const float p0 = prev[e1 < 16 ? e1 : 15], p1 = prev[e1 + 1 < 16 ? e1 + 1 : 15], p2 = prev[e1 + 2 < 16 ? e1 + 2 : 15],
                                    p3 = prev[e1 + 3 < 16 ? e1 + 3 : 15];
                        if constexpr (img == 0)
                        {
                            h2p_slot<u == 0 ? 0 : 2, (u < kn_next), NA, NX, younger, u * 1024>(cur, fr[u], ssum[sum_model], e[0], e[1], e[2], e[3], bqt[u % BW],
                                                                                               qacc, next_at, p0, p1, p2, p3);
                            if constexpr (u + BW < KQF) bqt[u % BW] = bq_load(u + BW);
                        }
                        else
                            h2p_slot<u == 0 ? 1 : 2, (u < kn_next), NA, NX, younger, u * 1024>(cur, fr[u], ssum[sum_model], e[0], e[1], e[2], e[3], bl[u], qacc,
                                                                                               next_at, p0, p1, p2, p3);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                } else {
                    h2p_for<0, kn>([&](auto U) {
                        constexpr int u = decltype(U)::value;
                        if constexpr (gi == G - 1 && DMA_AT != 0 && u == DMA_AT)
                            if (S + NB < n_stage_total) stage_load_next(slot_cur);
                        // our reads issued after fragment u of this image: the rest of this image's, then the next image's first u
                        constexpr int younger = (kn - 1 - u) + (u < kn_next ? u : kn_next);
                        u32x4 &fu = fr[u];
                        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fu) : "n"(younger));
                        if constexpr (img == 0)
                            cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[u]), bqt[u % BW], u == 0 ? zero1 : cur, 0, 0, 0);
                        else
                            cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[u]), bl[u], u == 0 ? qacc : cur, 0, 0, 0);
                        f32x16 &cpin = cur;
                        asm volatile("" : "+v"(cpin));                         // the MFMA stays in front of this slot's vector work
                        if constexpr (u < kn_next) frag_read(U, next_at);
                        if constexpr (carry) {
                            // slot u issues exps [cum(u - 1), cum(u)) and adds up the slot before's [cum(u - 2), cum(u - 1))
                            constexpr int e0 = h2p_cum(kn, u - 2), e1 = h2p_cum(kn, u - 1), e2 = h2p_cum(kn, u);
                            h2p_for<e0, e1>([&](auto R) {
                                float &sv = ssum[prev_model];
                                const float ev = e[decltype(R)::value - e0];
                                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sv) : "v"(ev));
                            });
                            h2p_for<e1, e2>([&](auto R) {
                                float &ev = e[decltype(R)::value - e1];
                                const float pv = prev[decltype(R)::value];
                                asm volatile("v_exp_f32 %0, %1" : "=v"(ev) : "v"(pv));
                            });
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                h2p_for<kn, kn_next>([&](auto U) { frag_read(U, next_at); });          // (a Q image shorter than the L image after it)
                if constexpr (carry) {
                    constexpr int e0 = h2p_cum(kn, kn - 2), e1 = h2p_cum(kn, kn - 1);
                    if constexpr (kn <= 1) {
#pragma unroll
                        for (int r = 0; r < 16; r++) ssum[prev_model] += __builtin_amdgcn_exp2f(prev[r]);
                    } else {
                        asm volatile("s_nop 0");
                        h2p_for<e0, e1>([&](auto R) {
                            float &sv = ssum[prev_model];
                            const float ev = e[decltype(R)::value - e0];
                            asm volatile("v_add_f32 %0, %0, %1" : "+v"(sv) : "v"(ev));
                        });
                    }
                }
                if constexpr (img == 0) {
                    // Q - O.  As asm: written as vector code hipcc builds splat(O) as 16-register tuples -- several of them, hoisted
                    // out of the tile loop and spilled (708 bytes of scratch per lane, 44 reloads per mixture tile).  An asm reader of
                    // an MFMA result gets no hazard handling: 8 passes + 3 wait states after the chain's last MFMA (s_nop 12 = 13).
                    asm volatile("s_nop 12");
                    h2p_for<0, 16>([&](auto R) {
                        float x = qacc[decltype(R)::value];
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(off));
                        qacc[decltype(R)::value] = x;
                    });
                    asm volatile("s_nop 1");
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (gi == G - 1) {
                    S++;
                    slot_cur = slot_nxt;
                    slot_nxt = slot_nxt == NB - 1 ? 0 : slot_nxt + 1;
                }
            });
        }
        // the last image's epilogue has no chain to ride on (and its MFMAs were asm: no hazard handling from the compiler)
        asm volatile("s_nop 12");
#pragma unroll
        for (int r = 0; r < 16; r++) ssum[SB] += __builtin_amdgcn_exp2f(acc[(N_IMG - 1) & 1][r]);     // (the last image of a stage is an odd one)
        float fin[SB];
        const int last_model = n_st * G - 2;                  // image n_st * G - 1
#pragma unroll
        for (int si = 0; si < SB; si++) fin[si] = ssum[si] + (si == last_model ? ssum[SB] : 0.0f);
        {
            int tids[4], first_col[4], n_seg;
            h2p_unit(a, unit, tids, first_col, n_seg);
            if (n_seg == 1) h2s_close_block(a, sb, blk, fin, off, valid, has, tids[0], row, lane, hh, safe_ll2);
            else h2s_close_block_packed(a, sb, blk, fin, off, valid, has, tids, first_col, n_seg, row, lane, hh, safe_ll2);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// The exception pass (gmm.cc:34-38, :237-244 semantics: lse.hpp): the FRAMES the main pass listed, classic online log-sum-exp with
// the reference's underflow rule.  Round 4: a 4-wave workgroup streams ONE block's images through LDS once for its four waves, as
// the main kernel does (through round 3 a lone wave fetched its images straight from L2 and the pass ran at the L2's bandwidth).
// Round 6: the list holds {tile, column mask} and a wave's 32 columns take the listed frames of SEVERAL tiles side by side
// (h2s_plan_kernel cuts a block's list into items of <= 32 frames; a tile's frames never straddle two items) -- configs[3]'s
// shard lists 12 200 tiles per block with one or two frames each: 390 items instead of 12 200 whole tiles.
//
// What a (tile, model) partial is, whatever the packing: the main pass's fixed-order sum over the tile's unlisted frames, plus the
// sum of the listed frames' values taken one by one in ascending column order (formed by the lane that owns the entry).  A (frame,
// model) value is formed by one lane over the mixture tiles in order.  Neither depends on what a tile is packed with or where.
__device__ __forceinline__ int *h2s_plan_starts(const H2sArgs &a, int blk) {
    return reinterpret_cast<int *>(a.exc_list + (size_t)a.n_blocks * a.n_tiles) + (size_t)blk * (a.n_tiles + 1);
}

constexpr int H2S_PLAN_INLINE_MAX_TILES = 1024;
// One wave per block: walks the block's entries in list order and opens a new item whenever the next tile's frames do not fit the
// current item's 32 columns.  starts[i] = first entry of item i, starts[n_items] = count; returns n_items (wave-uniform).
__device__ __forceinline__ int h2s_plan_block(const H2sArgs &a, int blk, int lane) {
    const int count = min(a.exc_count[blk], a.n_tiles);
    int *starts = h2s_plan_starts(a, blk);
    const int2 *list = a.exc_list + (size_t)blk * a.n_tiles;
    int n_items = 0, fill = 32;                                 // (the first entry opens item 0)
    for (int base = 0; base < count; base += 64) {
        const int pop = base + lane < count ? __builtin_popcount((unsigned)list[base + lane].y) : 0;
        const int n = min(64, count - base);
        for (int i = 0; i < n; i++) {                           // wave-uniform: scalar arithmetic
            const int p = __builtin_amdgcn_readlane(pop, i);
            if (fill + p > 32) {
                if (lane == 0) starts[n_items] = base + i;
                n_items++;
                fill = 0;
            }
            fill += p;
        }
    }
    if (lane == 0) starts[n_items] = count;
    return n_items;
}

// ... as a kernel of its own in front of the exception pass: n_items behind the counts.  (Small batches -- a serving decision is ten
// tiles -- skip this launch: `plan_inline`, every workgroup of the exception pass then forms its block's plan itself, all of them
// the same values.)
__global__ __launch_bounds__(64)
void h2s_plan_kernel(const H2sArgs a) {
    const int n_items = h2s_plan_block(a, blockIdx.x, threadIdx.x);
    if (threadIdx.x == 0) a.exc_count[a.n_blocks + blockIdx.x] = n_items;
}

__device__ __forceinline__ double shfl_f64(double v, int src_lane) {
    union { double d; int i[2]; } x, y;
    x.d = v;
    y.i[0] = __shfl(x.i[0], src_lane);
    y.i[1] = __shfl(x.i[1], src_lane);
    return y.d;
}

template <int KQF, int KLF>
__global__ __launch_bounds__(256, 2)
void gmm_score_h2s_online_kernel(const H2sArgs a) {
    constexpr int SB = SHARED_SB, WAVES = 4, G = 2;
    constexpr int Q_U4 = KQF * 64, L_U4 = KLF * 64;
    constexpr int IMG_U4 = Q_U4 > L_U4 ? Q_U4 : L_U4;
    constexpr int STRIDE_U4 = (1 + SB) * IMG_U4;
    constexpr int N_STAGES = (1 + SB) / G;
    constexpr int KM = KQF > KLF ? KQF : KLF;
    static_assert((1 + SB) % G == 0 && (N_STAGES % 2) == 0, "stages must tile the 16 images and alternate buffers");
    __shared__ uint4 lds_a[G * IMG_U4];
    __shared__ uint4 lds_b[G * IMG_U4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;
    const int hh = lane >> 5;
    const int gy = (int)gridDim.x / a.n_blocks;            // workgroups per block
    const int blk = (int)blockIdx.x / gy, y = (int)blockIdx.x - blk * gy;
    int n_items;
    if (a.plan_inline) {
        if (a.exc_count[blk] == 0) return;                  // (the usual case: nothing listed)
        __shared__ int s_items;
        if (wave == 0) {
            const int n = h2s_plan_block(a, blk, lane);
            if (lane == 0) s_items = n;
        }
        __threadfence();                                    // (this workgroup reads the starts it wrote itself)
        __syncthreads();
        n_items = s_items;
    } else {
        n_items = a.exc_count[a.n_blocks + blk];
    }
    if (y * WAVES >= n_items) return;                       // (the usual case: nothing listed)
    const int *starts = h2s_plan_starts(a, blk);
    const int2 *list = a.exc_list + (size_t)blk * a.n_tiles;
    const float near_thr = lse_near_threshold(a.clamp);
    const f32x16 zero1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const SharedBlock sb = a.blocks[blk];
    const uint4 *stream = a.params + sb.offset_u4;
    constexpr int N_PIECES = G * IMG_U4 / 64;
    auto stage_load = [&](uint4 *dst, const uint4 *src) {
#pragma unroll
        for (int i = 0; i < (N_PIECES + WAVES - 1) / WAVES; i++) {
            const int piece = i * WAVES + wave;
            if (piece < N_PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 64), 16, 0, 0);
        }
    };
    auto publish_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const int n_stage_total = a.n_mix_tiles * N_STAGES;
    for (int chunk = y; chunk * WAVES < n_items; chunk += gy) {
        const int item = chunk * WAVES + wave;
        const bool has = item < n_items;                    // (a wave beyond the plan shadows its last item and stores nothing)
        const int it = has ? item : n_items - 1;
        const int e0 = starts[it], n_ent = starts[it + 1] - e0;            // 1 .. 32 entries, <= 32 frames between them
        // lane i < n_ent owns entry i: its tile, its listed columns, where they start among the wave's 32 columns
        const int2 ent = lane < n_ent ? list[e0 + lane] : make_int2(0, 0);
        const int pop = __builtin_popcount((unsigned)ent.y);
        int incl = pop;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int excl = incl - pop;
        const int total = __shfl(incl, n_ent - 1);
        // column c of the wave: the entry it belongs to, and which of that entry's listed frames it is
        int my_e = 0;
        for (int i = 1; i < n_ent; i++)                                     // wave-uniform trip count
            if (col >= __shfl(excl, i)) my_e = i;
        const bool valid = col < total;
        const int my_tile = __shfl(ent.x, my_e);
        unsigned m = (unsigned)__shfl(ent.y, my_e);
        const int rank = col - __shfl(excl, my_e);
        for (int r = 0; r < rank; r++) m &= m - 1;                          // drop the `rank` lowest listed columns
        const int tcol = valid ? __builtin_ctz(m | 0x80000000u) : 0;
        const TileDesc tile = a.tiles[my_tile];
        const int64_t row = tile.start + tcol;
        f16x8 bq[1][KQF], bl[1][KLF];
        float zmax = 0.0f;
        h2s_build_b<KQF>(bq[0], a.X + row * a.dim, a.center, a.scale, a.q_desc, hh, true, zmax);
        h2s_build_b<KLF>(bl[0], a.X + row * a.dim, a.center, a.scale, a.l_desc, hh, false, zmax);
        if (zmax >= 255.0f) atomicOr(a.oor_flag, 1);
        float mx[SB], ssum[SB];
#pragma unroll
        for (int si = 0; si < SB; si++) {
            mx[si] = NEG_BIG;
            ssum[si] = 0.0f;
        }
        uint4 fr[KM];
        auto load_frags = [&](const uint4 *at, int kn) {
#pragma unroll
            for (int ks = 0; ks < KM; ks++)
                if (ks < kn) fr[ks] = at[ks * 64];
        };
        __syncthreads();                      // the previous chunk's readers are done with both buffers
        stage_load(lds_a, stream);
        if (n_stage_total > 1) stage_load(lds_b, stream + (size_t)G * IMG_U4);
        publish_barrier();
        load_frags(lds_a + lane, KQF);
        for (int t = 0; t < a.n_mix_tiles; t++) {
            const uint4 *tsrc = stream + (size_t)t * STRIDE_U4;
            const bool more_tiles = t + 1 < a.n_mix_tiles;
            f32x16 qacc[1];
#pragma unroll
            for (int st = 0; st < N_STAGES; st++) {
                uint4 *cur = (st & 1) ? lds_b : lds_a;
                const uint4 *nxt = (st & 1) ? lds_a : lds_b;
#pragma unroll
                for (int gi = 0; gi < G; gi++) {
                    const int img = st * G + gi;
                    f32x16 acc[1];
                    if (img == 0) h2s_chain_regs<KQF, KM, 1>(qacc, {zero1}, fr, bq);
                    else h2s_chain_regs<KLF, KM, 1>(acc, qacc, fr, bl);
                    __builtin_amdgcn_sched_barrier(0);
                    if (gi == G - 1) {
                        // every wave holds its fragments of this stage: `cur` may be refilled, and the stage after this one has landed
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
                    if (img > 0) {
                        lse_update16(acc[0], mx[img - 1], ssum[img - 1], near_thr);
                        asm volatile("" : "+v"(mx[img - 1]), "+v"(ssum[img - 1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // most listed frames a tile of this item has (wave-uniform)
        int max_pop = pop;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) max_pop = max(max_pop, __shfl_xor(max_pop, o));
        const unsigned own_cols = pop >= 32 ? 0xffffffffu : ((1u << pop) - 1u);
#pragma unroll
        for (int si = 0; si < SB; si++) {
            const float ll = lse_close2(mx[si], ssum[si], other_half(mx[si]), other_half(ssum[si]), a.clamp);
            const bool live = has && valid && hh == 0 && si < sb.n_models;
            const double v = live ? (double)ll : 0.0;
            if (live && a.frame_ll) a.frame_ll[(int64_t)(sb.first_model + si) * a.n_frames + row] = ll;
            const unsigned hot_cols = (unsigned)__builtin_amdgcn_ballot_w64(live && ll < a.band_hi);
            // the owner of an entry adds its frames' values one by one, in ascending column order
            double sum = 0.0;
            for (int j = 0; j < max_pop; j++) {
                const double t = shfl_f64(v, min(excl + j, 31));
                if (j < pop) sum += t;
            }
            if (has && lane < n_ent && si < sb.n_models) {
                double *p = a.partial + (int64_t)ent.x * a.n_models + sb.first_model + si;
                // (a frame in the band where the reference's partial products decide: the whole (tile, model) goes to gmm_flush.hip)
                *p = ((hot_cols >> excl) & own_cols) != 0 ? SR_FLUSH_POISON : *p + sum;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// LDS the pipelined kernel takes: its ring of two stages of four images plus the 12 waves' quadratic-half fragments
__host__ __device__ constexpr bool h2p_fits(int kqf, int klf) { return klf >= 2 && kqf <= klf && (2 * 4 * klf + 12 * kqf) * 1024 <= 160 * 1024; }


template <int KQF, int KLF, int COLS, int WAVES, bool PIN = false, bool MS = false>
static int launch_h2s(const H2sLaunch &l) {
    constexpr bool BQ_LDS = h2s_bq_in_lds(KQF, KLF, WAVES);
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
    a.n_blocks = l.n_blocks;
    a.n_frames = l.n_frames;
    a.dim = l.dim;
    a.n_models = l.n_models;
    a.n_mix_tiles = l.n_mix_tiles;
    a.clamp = l.clamp;
    a.n_groups = l.n_groups;
    a.n_tiles = l.n_tiles;
    a.log2_k = l.log2_k;
    a.force_exc = l.force_exc;
    a.plan_inline = l.n_tiles <= H2S_PLAN_INLINE_MAX_TILES;
    a.band_hi = l.band_hi;
    // long grids in launches of ~H2S_ROUNDS_PER_LAUNCH rounds of resident workgroups (see the kernel)
    constexpr int TILES_WG = MS ? 1 : WAVES * COLS;
    const int resident = ctx().n_cu * (WAVES > 4 ? 1 : h2s_waves_per_eu(KQF, KLF, COLS, WAVES, MS));
    if (PIN && l.n_work <= 0) fail("the pipelined shared-sigma kernel needs its work table behind the tile table (ensure_work_table)");
    const int n_wg = ((PIN ? l.n_work : l.n_tiles) + TILES_WG - 1) / TILES_WG;
    a.n_units = PIN ? (l.n_work + H2P_ROUND_ITEMS - 1) / H2P_ROUND_ITEMS * H2P_ROUND_ITEMS : l.n_tiles;
    static_assert(!PIN || H2P_ROUND_ITEMS % TILES_WG == 0, "a workgroup's work items never straddle the table's padding");
    // (the 12-wave form has one workgroup per CU sweeping the stream: nothing drifts apart, 1 / 3 / 5 / 9 / 18 launches
    // per configs[2] pass all take 0.281-0.283 s -- one launch)
    int wg_per_launch = WAVES > 4 ? std::max(8, (n_wg + 7) / 8 * 8)
                        : std::max(8, (H2S_ROUNDS_PER_LAUNCH * resident / std::max(1, l.n_groups)) / 8 * 8);
    int n_launches = 0;
    for (int base = 0; base < n_wg; base += wg_per_launch) {
        n_launches++;
        a.tile_base = base * TILES_WG;
        const int n = std::min(wg_per_launch, n_wg - base);
        // A serving decision (gmm_score_h2m_kernel, every workgroup resident at once): a group's workgroups on ONE XCD, so that a
        // block's images come into one L2, once.  Spread over the XCDs as above, every L2 fetched all 28 MiB of configs[2]'s images
        // for its share of the ten tiles and the fabric bounded the kernel: 300 frames 0.072 -> 0.057 ms, a decision 0.126 -> 0.105
        // (600 / 900 frames: 0.094 -> 0.088 / 0.091).  The LDS-staged 4-wave shapes wait for their stages' round trips, not for
        // bytes (0.110 either way), and larger grids ran slower XCD-major (h2s_wg_assignment).
        const int per_xcd = ((l.n_groups + 7) / 8) * n;
        const bool xcd_groups = MS && KLF <= H2M_MAX_KLF && l.n_groups > 1 && n == n_wg && per_xcd <= 2 * (ctx().n_cu / 8);
        dim3 grid(xcd_groups ? (unsigned)(8 * ((l.n_groups + 7) / 8) * n) : (unsigned)((int64_t)l.n_groups * ((n + 7) / 8) * 8));
        // Launch order.  With ONE group every workgroup sweeps all blocks from block 0 on, in phase with its XCD's others (see the
        // kernel).  With several, workgroups of different groups stream different blocks: group-fastest order (rounds 2-3) put one
        // workgroup of EVERY block on each XCD at a time -- no reuse in its L2, every stage a trip to HBM, the loop bound by that
        // latency (64 utterances x 300 frames: ~780 cycles per image against 381 in a full-chip pass).  Group-major order runs a
        // block's workgroups side by side.
        a.rows8 = (n + 7) / 8;
        a.n_wg = n;
        a.group_major = xcd_groups ? 2 : l.n_groups > 1;
        // (the model-split shape: gmm_score_h2m_kernel, fragments straight into registers, where two sets of them fit; the LDS form
        // of round 4 for the longest chains -- D = 46 .. 48 -- whose two sets spill)
        constexpr bool M_DIRECT = MS && KLF <= H2M_MAX_KLF;
        constexpr size_t dyn = M_DIRECT ? 0 : (BQ_LDS ? (size_t)WAVES * KQF * 64 * sizeof(uint4) : 0) + (MS ? sizeof(uint4) : 0);
        if constexpr (dyn > 0) {
            static bool attr_set[MAX_DEVICES] = {};
            if (!attr_set[ctx().device]) {
                if constexpr (PIN)
                    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gmm_score_h2p_kernel<KQF, KLF>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
                else
                    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gmm_score_h2s_kernel<KQF, KLF, COLS, WAVES, MS>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
                attr_set[ctx().device] = true;
            }
        }
        if constexpr (PIN)
            hipLaunchKernelGGL((gmm_score_h2p_kernel<KQF, KLF>), grid, dim3(WAVES * 64), dyn, ctx().stream, a);
        else if constexpr (M_DIRECT)
            hipLaunchKernelGGL((gmm_score_h2m_kernel<KQF, KLF>), grid, dim3(WAVES * 64), 0, ctx().stream, a);
        else
            hipLaunchKernelGGL((gmm_score_h2s_kernel<KQF, KLF, COLS, WAVES, MS>), grid, dim3(WAVES * 64), dyn, ctx().stream, a);
    }
    a.tile_base = 0;
    // the exception pass: the plan (one wave per block cuts its list into items of <= 32 listed frames), then 4-wave workgroups over
    // the items (a couple of resident ones per block and CU's worth of the chip; with nothing listed -- the usual case -- they
    // leave at once)
    if (!a.plan_inline) hipLaunchKernelGGL(h2s_plan_kernel, dim3((unsigned)l.n_blocks), dim3(64), 0, ctx().stream, a);
    const int gy = std::max(1, (2 * ctx().n_cu) / std::max(1, l.n_blocks));
    hipLaunchKernelGGL((gmm_score_h2s_online_kernel<KQF, KLF>), dim3((unsigned)(l.n_blocks * gy)), dim3(256), 0, ctx().stream, a);
    return n_launches;
}

// workgroups resident per CU, and 32-frame tiles per workgroup, of shape `shape` (0: 4 waves; 1: 12 waves; 2: 12 waves, pipelined;
// 3: 4 waves on one tile, the block's models split between them)
int h2s_resident_per_cu(int kqf, int klf, int shape) { return (shape == 0 || shape == 3) ? h2s_waves_per_eu(kqf, klf, 1, 4, shape == 3) : 1; }
int h2s_tiles_per_wg(int shape) { return shape == 0 ? 4 : shape == 3 ? 1 : 12; }
bool h2s_pipelined_available(int kqf, int klf) { return h2p_fits(kqf, klf); }
bool h2s_msplit_direct(int kqf, int klf) { (void)kqf; return klf <= H2M_MAX_KLF; }

// returns the number of launches of the main kernel the pass was cut into
int launch_score_h2_shared(const H2sLaunch &l, int KQF, int KLF) {
#define SR_H2S_CASE(Q, L)                                       \
    if (KQF == Q && KLF == L) {                                 \
        if constexpr (h2p_fits(Q, L))                           \
            if (l.shape == 2) return launch_h2s<Q, L, 1, 12, true>(l); \
        if (l.shape == 3) return launch_h2s<Q, L, 1, 4, false, true>(l); \
        if (l.shape >= 1) return launch_h2s<Q, L, 1, 12>(l);    \
        return launch_h2s<Q, L, 1, 4>(l);                       \
    }
    // KQF = ceil(3D/16), KLF = ceil((3D+2)/16): equal, or one apart at D = 5, 16, 21, 32, 37, 48
    SR_H2S_CASE(1, 1) SR_H2S_CASE(1, 2) SR_H2S_CASE(2, 2) SR_H2S_CASE(3, 3) SR_H2S_CASE(3, 4) SR_H2S_CASE(4, 4)
    SR_H2S_CASE(4, 5) SR_H2S_CASE(5, 5) SR_H2S_CASE(6, 6) SR_H2S_CASE(6, 7) SR_H2S_CASE(7, 7)
    SR_H2S_CASE(7, 8) SR_H2S_CASE(8, 8) SR_H2S_CASE(9, 9) SR_H2S_CASE(9, 10)
#undef SR_H2S_CASE
    fail("no split-fp16 shared-sigma scoring kernel for %d + %d contraction steps", KQF, KLF);
}

}  // namespace sr
