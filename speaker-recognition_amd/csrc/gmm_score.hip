// gmm_score.hip -- diagonal-GMM log-likelihood scoring on gfx950 (CDNA4), the hot loop of
// the path.  Replaces Gaussian::probability_of_fast_exp (src/gmm/src/gmm.cc:176-202),
// GMM::log_probability_of_fast_exp (:237-244) and threaded_log_probability_of (:533-569), and
// -- by looping all S speaker models over a resident frame tile -- the per-speaker ABI loop
// of GMMSet.predict_one (src/testbench/gmmset.py:59-64, 95-99).
//
// Formulation:  LL(x) = ln2 * log2sum_k 2^(c_k - sum_d (x_d s_kd + m_kd)^2)
// with the tables of gmm_model.hpp; fp32, online max; optional reference-compat clamp.
//
// Mapping: a workgroup (256 threads = 4 wave64) owns one tile of 256*F frames of one
// utterance; every lane keeps F frames (F*D floats) in VGPRs for the whole kernel, so X is
// read from HBM exactly once.  Mixture parameters stream through LDS in chunks (double
// buffered by LDS-DMA); all lanes read the same LDS address (broadcast),
// two ds_read_b128 feed 8*F FMAs.  The log-sum-exp is online per lane, in the log2 domain
// (v_exp_f32 / v_log_f32 are base-2).  No MFMA: the 2-FMA distance form is not a contraction.
#include "lse.hpp"
#include "score.hpp"
#include "wave_ops.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace sr {

struct ScoreArgs {
    const float *X;            // [n_frames][dim] row-major fp32
    const TileDesc *tiles;
    const float4 *params;
    const float *center;       // [DP] subtracted from every frame (PackedModels::center)
    const ChunkDesc *chunks;
    const int *group_chunk_begin;  // [G+1]
    double *partial;           // [n_tiles][S][4]  per-wave partial sums
    float *frame_ll;           // [S][n_frames] or nullptr
    int64_t n_frames;
    int dim;
    int n_models;
    int clamp;
    float band_hi;             // below it a frame goes to the partial-product path (lse.hpp); -inf: never
};


__host__ __device__ constexpr int score_waves_per_eu(int dp, int f) {
    return dp > 64 ? 2 : (dp * f + 44 <= 128) ? 4 : (dp * f + 44 <= 168) ? 3 : 2;     // wide rows: LDS chunks of 20-33 KB
}

typedef float v2f __attribute__((ext_vector_type(2)));

// Lane-private arithmetic on either one frame (float) or a packed pair of frames (v2f ->
// v_pk_fma_f32, the mixture constants broadcast to both halves through op_sel).
template <bool PK> struct Lanes;
template <> struct Lanes<false> {
    using T = float;
    static constexpr int W = 1;
    static __device__ __forceinline__ T zero() { return 0.0f; }
    static __device__ __forceinline__ T fma_bcast(T x, float s, float m) { return fmaf(x, s, m); }
    static __device__ __forceinline__ T fma_sq(T t, T acc) { return fmaf(t, t, acc); }
    static __device__ __forceinline__ float get(T v, int) { return v; }
    static __device__ __forceinline__ void set(T &v, int, float s) { v = s; }
};
template <> struct Lanes<true> {
    using T = v2f;
    static constexpr int W = 2;
    static __device__ __forceinline__ T zero() { return (v2f){0.0f, 0.0f}; }
    static __device__ __forceinline__ T fma_bcast(T x, float s, float m) {
        return __builtin_elementwise_fma(x, (v2f){s, s}, (v2f){m, m});
    }
    static __device__ __forceinline__ T fma_sq(T t, T acc) { return __builtin_elementwise_fma(t, t, acc); }
    static __device__ __forceinline__ float get(T v, int e) { return e ? v.y : v.x; }
    static __device__ __forceinline__ void set(T &v, int e, float s) { if (e) v.y = s; else v.x = s; }
};

template <int DP, int F, bool PK>
__global__ __launch_bounds__(256, score_waves_per_eu(DP, F))
void gmm_score_kernel(const float *__restrict__ X, const TileDesc *__restrict__ tiles,
                      const float4 *__restrict__ params, const float *__restrict__ center,
                      const ChunkDesc *__restrict__ chunks,
                      const int *__restrict__ group_chunk_begin, double *__restrict__ partial,
                      float *__restrict__ frame_ll, int64_t n_frames, int dim, int n_models,
                      int clamp, int n_groups, int n_tiles, float band_hi) {
    using L = Lanes<PK>;
    using XT = typename L::T;
    constexpr int W = L::W;
    constexpr int NV = F / W;                // lane-private vectors per dim
    static_assert(F % W == 0, "packed variant needs an even number of frames per lane");
    constexpr int REC = 2 * DP + 1;          // float4 per record
    constexpr int CHUNK_F4 = CB * REC;       // float4 per LDS buffer
    constexpr int PF = (CHUNK_F4 + 255) / 256;
    // Two separately named LDS objects (not one [2][..] array): the buffer a ds_read touches
    // is then statically distinct from the one an in-flight LDS-DMA writes, which lets hipcc
    // keep the DMA outstanding across the compute instead of draining vmcnt(0) first.
    __shared__ float4 lds_a[CHUNK_F4];
    __shared__ float4 lds_b[CHUNK_F4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware 1-D grid: workgroup b runs on XCD b % 8 (observed dispatch order, speed only), so
    // the G workgroups that share a frame tile are given ids 8 apart: same XCD, same L2, adjacent
    // in dispatch order -> the tile's rows leave HBM / the fabric once instead of G times.
    const int tile_lo = blockIdx.x & 7;
    const int q = blockIdx.x >> 3;
    const int g = q % n_groups;
    const int tile_id = (q / n_groups) * 8 + tile_lo;
    if (tile_id >= n_tiles) return;              // padding workgroups (whole workgroup, before any barrier)
    const TileDesc tile = tiles[tile_id];
    const int chunk_begin = group_chunk_begin[g];
    const int chunk_end = group_chunk_begin[g + 1];

    // LDS-DMA (global_load_lds_dwordx4): LDS destination = wave-uniform base + lane*16, so a
    // chunk (a linear run of float4) lands as a linear image; no VGPR round trip.
    auto stage = [&](float4 *dst, const ChunkDesc cd) {
        const float4 *src = params + cd.offset_f4;
        const int n4 = cd.n_records * REC;
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const int base = (i * 4 + wave) * 64;
            if (base + lane < n4)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + base + lane),
                    (__attribute__((address_space(3))) void *)(dst + base), 16, 0, 0);
        }
    };
    stage(lds_a, chunks[chunk_begin]);   // in flight while the frames are fetched

    // ---- resident frames: frame f of this lane = tile.start + f*256 + tid ----
    XT x[NV][DP];
    bool valid[F];
    int64_t row[F];
#pragma unroll
    for (int f = 0; f < F; f++) {
        const int local = f * 256 + tid;
        valid[f] = local < tile.count;
        row[f] = tile.start + (valid[f] ? local : 0);
        const float *src = X + row[f] * dim;
        if (dim == DP) {
#pragma unroll
            for (int d = 0; d < DP; d++) L::set(x[f / W][d], f % W, src[d] - center[d]);
        } else {
#pragma unroll
            for (int d = 0; d < DP; d++) L::set(x[f / W][d], f % W, (d < dim) ? src[d] - center[d] : 0.0f);
        }
    }

    float m[F], ssum[F];
#pragma unroll
    for (int f = 0; f < F; f++) {
        m[f] = NEG_BIG;
        ssum[f] = 0.0f;
    }
    const float drop_thr = clamp ? LSE_MINLOG2 : -3.0e38f;      // wave-uniform
    dma_publish_barrier();   // drains the LDS-DMA of chunk 0 (hipcc emits vmcnt(0) before the barrier)

    // One chunk: stage the next one into `other`, run all records of `cur`, close the model
    // if the chunk is its last, then barrier (next chunk landed; everyone is done with `cur`).
    auto do_chunk = [&](const float4 *cur, float4 *other, int c) {
        const ChunkDesc cd = chunks[c];
        if (c + 1 < chunk_end) stage(other, chunks[c + 1]);

        for (int r = 0; r < cd.n_records; r++) {
            const float4 *rec = cur + r * REC;
            XT acc[NV][KB];
#pragma unroll
            for (int h = 0; h < NV; h++)
#pragma unroll
                for (int j = 0; j < KB; j++) acc[h][j] = L::zero();
#pragma unroll
            for (int d = 0; d < DP; d++) {
                const float4 p0 = rec[2 * d];
                const float4 p1 = rec[2 * d + 1];
#pragma unroll
                for (int h = 0; h < NV; h++) {
                    const XT xv = x[h][d];
                    const XT t0 = L::fma_bcast(xv, p0.x, p0.y);
                    const XT t1 = L::fma_bcast(xv, p0.z, p0.w);
                    const XT t2 = L::fma_bcast(xv, p1.x, p1.y);
                    const XT t3 = L::fma_bcast(xv, p1.z, p1.w);
                    acc[h][0] = L::fma_sq(t0, acc[h][0]);
                    acc[h][1] = L::fma_sq(t1, acc[h][1]);
                    acc[h][2] = L::fma_sq(t2, acc[h][2]);
                    acc[h][3] = L::fma_sq(t3, acc[h][3]);
                }
            }
            const float4 cc = rec[2 * DP];
#pragma unroll
            for (int f = 0; f < F; f++) {
                const float v0 = cc.x - L::get(acc[f / W][0], f % W);
                const float v1 = cc.y - L::get(acc[f / W][1], f % W);
                const float v2 = cc.z - L::get(acc[f / W][2], f % W);
                const float v3 = cc.w - L::get(acc[f / W][3], f % W);
                const float mx = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                const float mn = fmaxf(m[f], mx);
                // the reference's sub-DBL_MIN terms are exactly 0 (lse.hpp).  Branch-free on purpose: a wave-uniform "only
                // next to the boundary" branch here, inside the unrolled frame loop, cost this kernel 300-800 dwords of
                // scratch per lane (1240 B at D = 39, F = 4: 10x slower); four selects per frame and record are ~4 %.
                // (a running maximum below the boundary means every earlier term was dropped: ssum is already 0)
                // (the select sits on exp2's ARGUMENT -- 2^-1e30 = 0 -- so that the compiler has nothing expensive to branch around)
                const float e0 = __builtin_amdgcn_exp2f(v0 >= drop_thr ? v0 - mn : LSE_NEG_BIG);
                const float e1 = __builtin_amdgcn_exp2f(v1 >= drop_thr ? v1 - mn : LSE_NEG_BIG);
                const float e2 = __builtin_amdgcn_exp2f(v2 >= drop_thr ? v2 - mn : LSE_NEG_BIG);
                const float e3 = __builtin_amdgcn_exp2f(v3 >= drop_thr ? v3 - mn : LSE_NEG_BIG);
                ssum[f] = fmaf(ssum[f], __builtin_amdgcn_exp2f(m[f] - mn), (e0 + e1) + (e2 + e3));
                m[f] = mn;
            }
        }

        if (cd.model_done >= 0) {   // wave-uniform: close the model, start the next one
            const int s = cd.model_done;
            double mine = 0.0;
            bool hot = false;              // a frame in the band of the reference's partial-product flushes (lse.hpp)
#pragma unroll
            for (int f = 0; f < F; f++) {
                // the reference's underflow behaviour (safe_log -> ln 1e-15, gmm.cc:34-38, :237-244): lse.hpp
                const float ll = lse_close1(m[f], ssum[f], clamp);
                if (valid[f]) {
                    mine += (double)ll;
                    if (frame_ll) frame_ll[(int64_t)s * n_frames + row[f]] = ll;
                    hot |= ll < band_hi;
                }
                m[f] = NEG_BIG;
                ssum[f] = 0.0f;
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

// Rows wider than MAX_REG_DIM (the reference has no limit, gmm.cc:40-51): the same direct form, same records, same lane = frame,
// with the D loop cut into slices of WIDE_DC dimensions.  A step = (chunk of <= CB records of one model, slice): the slice's
// parameters -- n_records runs of 2 WIDE_DC float4 inside the record-major layout -- land in LDS by LDS-DMA one step ahead, the
// lane fetches its row's slice (256 B of its own row; the tile's rows stay in L2 between steps), and the CB x KB running
// distances stay in registers across the chunk's slices; constants, log-sum-exp update and the model close follow the last
// slice exactly as above.  Any dim: the slice count is a run-time value.
__global__ __launch_bounds__(256, 3)
void gmm_score_wide_kernel(const float *__restrict__ X, const TileDesc *__restrict__ tiles,
                           const float4 *__restrict__ params, const float *__restrict__ center,
                           const ChunkDesc *__restrict__ chunks, const int *__restrict__ group_chunk_begin,
                           double *__restrict__ partial, float *__restrict__ frame_ll, int64_t n_frames, int dim, int dp,
                           int n_models, int clamp, int n_groups, int n_tiles, float band_hi) {
    constexpr int DC = WIDE_DC;
    constexpr int RUN = 2 * DC;              // float4 per record and slice
    constexpr int SLICE_F4 = CB * RUN;
    constexpr int PF = SLICE_F4 / 256;
    static_assert(RUN % 64 == 0 && SLICE_F4 % 256 == 0, "an LDS-DMA instruction stays inside one record's run");
    __shared__ float4 lds_a[SLICE_F4];
    __shared__ float4 lds_b[SLICE_F4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_lo = blockIdx.x & 7;      // XCD-aware order, as gmm_score_kernel
    const int q = blockIdx.x >> 3;
    const int g = q % n_groups;
    const int tile_id = (q / n_groups) * 8 + tile_lo;
    if (tile_id >= n_tiles) return;
    const TileDesc tile = tiles[tile_id];
    const int chunk_begin = group_chunk_begin[g];
    const int chunk_end = group_chunk_begin[g + 1];
    const int rec_f4 = 2 * dp + 1;
    const int n_dc = dp / DC;

    auto stage = [&](float4 *dst, const ChunkDesc cd, int dc) {
        const float4 *src = params + cd.offset_f4 + dc * RUN;
        const int n4 = cd.n_records * RUN;
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const int base = (i * 4 + wave) * 64;      // wave-uniform; record base / RUN, offset base % RUN + lane inside its run
            if (base < n4)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + (size_t)(base / RUN) * rec_f4 + (base % RUN) + lane),
                    (__attribute__((address_space(3))) void *)(dst + base), 16, 0, 0);
        }
    };
    stage(lds_a, chunks[chunk_begin], 0);

    const bool valid = tid < tile.count;
    const int64_t row = tile.start + (valid ? tid : 0);
    const float *xrow = X + row * dim;
    float m = NEG_BIG, ssum = 0.0f;
    const float drop_thr = clamp ? LSE_MINLOG2 : -3.0e38f;
    float acc[CB][KB];
    dma_publish_barrier();

    auto do_step = [&](const float4 *cur, float4 *other, int c, int dc) __attribute__((always_inline)) {
        const ChunkDesc cd = chunks[c];
        {
            const int dcn = dc + 1 < n_dc ? dc + 1 : 0;
            const int cn = dcn ? c : c + 1;
            if (cn < chunk_end) stage(other, chunks[cn], dcn);
        }
        float x[DC];
        const int d0 = dc * DC;
        if (d0 + DC <= dim) {                // wave-uniform
#pragma unroll
            for (int d = 0; d < DC; d++) x[d] = xrow[d0 + d] - center[d0 + d];
        } else {
#pragma unroll
            for (int d = 0; d < DC; d++) x[d] = (d0 + d < dim) ? xrow[d0 + d] - center[d0 + d] : 0.0f;
        }
        if (dc == 0) {
#pragma unroll
            for (int r = 0; r < CB; r++)
#pragma unroll
                for (int j = 0; j < KB; j++) acc[r][j] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < CB; r++) {
            if (r < cd.n_records) {          // wave-uniform
                const float4 *rec = cur + r * RUN;
#pragma unroll
                for (int d = 0; d < DC; d++) {
                    const float4 p0 = rec[2 * d];
                    const float4 p1 = rec[2 * d + 1];
                    const float t0 = fmaf(x[d], p0.x, p0.y);
                    const float t1 = fmaf(x[d], p0.z, p0.w);
                    const float t2 = fmaf(x[d], p1.x, p1.y);
                    const float t3 = fmaf(x[d], p1.z, p1.w);
                    acc[r][0] = fmaf(t0, t0, acc[r][0]);
                    acc[r][1] = fmaf(t1, t1, acc[r][1]);
                    acc[r][2] = fmaf(t2, t2, acc[r][2]);
                    acc[r][3] = fmaf(t3, t3, acc[r][3]);
                }
            }
        }
        if (dc == n_dc - 1) {
#pragma unroll
            for (int r = 0; r < CB; r++) {
                if (r < cd.n_records) {
                    const float4 cc = params[cd.offset_f4 + (size_t)r * rec_f4 + 2 * dp];      // wave-uniform address
                    const float v0 = cc.x - acc[r][0], v1 = cc.y - acc[r][1], v2 = cc.z - acc[r][2], v3 = cc.w - acc[r][3];
                    const float mx = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                    const float mn = fmaxf(m, mx);
                    // (the reference's sub-DBL_MIN terms are exactly 0: lse.hpp; select on exp2's argument as gmm_score_kernel)
                    const float e0 = __builtin_amdgcn_exp2f(v0 >= drop_thr ? v0 - mn : LSE_NEG_BIG);
                    const float e1 = __builtin_amdgcn_exp2f(v1 >= drop_thr ? v1 - mn : LSE_NEG_BIG);
                    const float e2 = __builtin_amdgcn_exp2f(v2 >= drop_thr ? v2 - mn : LSE_NEG_BIG);
                    const float e3 = __builtin_amdgcn_exp2f(v3 >= drop_thr ? v3 - mn : LSE_NEG_BIG);
                    ssum = fmaf(ssum, __builtin_amdgcn_exp2f(m - mn), (e0 + e1) + (e2 + e3));
                    m = mn;
                }
            }
            if (cd.model_done >= 0) {
                const int s = cd.model_done;
                const float ll = lse_close1(m, ssum, clamp);
                double mine = 0.0;
                bool hot = false;
                if (valid) {
                    mine = (double)ll;
                    if (frame_ll) frame_ll[(int64_t)s * n_frames + row] = ll;
                    hot = ll < band_hi;
                }
                m = NEG_BIG;
                ssum = 0.0f;
                mine = wave_sum_f64(mine);
                if (__builtin_amdgcn_ballot_w64(hot) != 0) mine = SR_FLUSH_POISON;
                if (lane == 0) partial[((int64_t)tile_id * n_models + s) * 4 + wave] = mine;
            }
        }
        dma_publish_barrier();
    };

    int c = chunk_begin, dc = 0;
    while (c < chunk_end) {
        do_step(lds_a, lds_b, c, dc);
        if (++dc == n_dc) { dc = 0; c++; }
        if (c >= chunk_end) break;
        do_step(lds_b, lds_a, c, dc);
        if (++dc == n_dc) { dc = 0; c++; }
    }
}

// Per utterance: add the tile/wave partials in a fixed order (deterministic), then the
// reference's argmax -- first maximum wins (gmmset.py:62-64, `max(enumerate(scores), key=...)`).
// A (tile, model) whose partial is SR_FLUSH_POISON holds a frame in the band where the reference's flushes of partial
// products decide (lse.hpp): it is left out of the sum and noted for gmm_flush.hip, which adds the tile's sum later.
//
// The order (round 4): the utterance's tiles in segments of `seg` consecutive tiles -- 32, or more for utterances of more
// than 32 k tiles, so that there are at most 1024 segments -- each summed front to back, then the segment sums front to back.
// It depends on the utterance's own tile count only, never on the batch around it.  One thread per (segment, model): until
// round 4 one thread per model walked ALL the tiles, which for the one long utterance of an E-step (400 k frames = 12 500
// tiles, one model) was 1.7 ms of dependent loads behind a 0.6 ms scoring kernel.
constexpr int FIN_LDS_DOUBLES = 4096;
struct FinalizeDelivery {      // SCORE_HOST_DELIVER (score.hpp); host == nullptr: off
    DeliverHeader *host;
    int *counters;             // the pass's counters: [0] saturation flag, [1] flush count, [2] this kernel's ticket, [4 ...]
    int n_counters;
    unsigned seq;
};
__global__ __launch_bounds__(256)
void gmm_finalize_kernel(const double *partial, const int *utt_tile_begin, int n_models,
                         int per_tile, double *sums, int *argmax, int2 *flush_list, int *flush_count, int flush_cap,
                         FinalizeDelivery dl) {
    __shared__ double seg_sum[FIN_LDS_DOUBLES];
    const int u = blockIdx.x;
    const int tb = utt_tile_begin[u], te = utt_tile_begin[u + 1];
    const int n_t = te - tb;
    const int seg = max(32, (n_t + 1023) / 1024);
    const int n_seg = (n_t + seg - 1) / seg;                   // <= 1024
    const int mb = max(1, min(n_models, FIN_LDS_DOUBLES / max(1, n_seg)));      // models per pass
    double best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int s0 = 0; s0 < n_models; s0 += mb) {
        const int m = min(mb, n_models - s0);
        for (int item = threadIdx.x; item < n_seg * m; item += 256) {
            const int sg = item / m, s = s0 + item - sg * m;   // (consecutive threads: consecutive models of one segment)
            const int t0 = tb + sg * seg, t1 = min(te, t0 + seg);
            double acc = 0.0;
            for (int t = t0; t < t1; t++) {
                // per_tile = 4: one double per wave of the tile's workgroup; 1: already combined
                const double *p = partial + ((int64_t)t * n_models + s) * per_tile;
                double tile_sum = 0.0;
                bool poisoned = false;
                for (int i = 0; i < per_tile; i++) {
                    poisoned |= flush_poisoned(p[i]);
                    tile_sum += p[i];
                }
                if (__builtin_expect(poisoned && flush_count != nullptr, 0)) {
                    const int idx = atomicAdd(flush_count, 1);
                    if (idx < flush_cap) flush_list[idx] = make_int2(t, s);
                    continue;
                }
                acc += tile_sum;
            }
            seg_sum[item] = acc;
        }
        __syncthreads();
        for (int sl = threadIdx.x; sl < m; sl += 256) {
            double acc = 0.0;
            for (int sg = 0; sg < n_seg; sg++) acc += seg_sum[sg * m + sl];
            const int s = s0 + sl;
            sums[(int64_t)u * n_models + s] = acc;
            if (acc > best) {                                  // (a thread meets its models in increasing order)
                best = acc;
                best_i = s;
            }
        }
        __syncthreads();
    }
    __shared__ double sv[256];
    __shared__ int si[256];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = best_i;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            const double ov = sv[threadIdx.x + w];
            const int oi = si[threadIdx.x + w];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) argmax[u] = (te > tb && si[0] != 0x7fffffff) ? si[0] : -1;
    if (dl.host == nullptr) return;
    // ---- the last workgroup to get here delivers the pass: results and counters to host memory, counters cleared ----
    __shared__ int s_last;
    __threadfence();                                       // this workgroup's sums / argmax / list entries before its ticket
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&dl.counters[2], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const int U = (int)gridDim.x;
    double *h_sums = reinterpret_cast<double *>(dl.host + 1);
    int *h_arg = reinterpret_cast<int *>(h_sums + (size_t)U * n_models);
    for (int i = threadIdx.x; i < U * n_models; i += 256)
        h_sums[i] = __hip_atomic_load(&sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = threadIdx.x; i < U; i += 256) h_arg[i] = __hip_atomic_load(&argmax[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
        dl.host->oor = __hip_atomic_load(&dl.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dl.host->n_flush = __hip_atomic_load(&dl.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                       // (the counters are read: clear them for the next pass)
    for (int i = threadIdx.x; i < dl.n_counters; i += 256) dl.counters[i] = 0;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&dl.host->seq, dl.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Hybrid sets: LL = ln(exp(LL_a) + exp(LL_b)) per frame and model, the per-tile float64 sums for gmm_finalize_kernel, and
// (optionally) the merged per-frame values.  With the reference's clamp on, a part whose mixtures all fell below
// DBL_MIN reports ln(1e-15) (lse.hpp): both -> ln(1e-15); one -> the other part alone, as the reference's linear-domain
// sum of the surviving terms.
__global__ __launch_bounds__(256)
void gmm_merge_kernel(const float *__restrict__ A, const float *__restrict__ B, const TileDesc *__restrict__ tiles,
                      int n_models, int64_t n_frames, int clamp, double *__restrict__ partial, float *__restrict__ out,
                      float band_hi) {
    __shared__ double part[4];
    const TileDesc tile = tiles[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool valid = tid < tile.count;
    const int64_t row = tile.start + (valid ? tid : 0);
    for (int s = 0; s < n_models; s++) {
        double mine = 0.0;
        bool hot = false;                             // a frame in the band of the reference's partial-product flushes (lse.hpp)
        if (valid) {
            const float a = A[(int64_t)s * n_frames + row], b = B[(int64_t)s * n_frames + row];
            float ll;
            // with the clamp on, a half whose terms all underflowed reports -inf (lse.hpp, clamp 2): out of band
            const bool sa = clamp && a == -INFINITY, sb = clamp && b == -INFINITY;
            if (sa || sb) {
                ll = sa ? (sb ? LSE_LN_1E_15 : b) : a;        // (both: the reference's ln 1e-15)
            } else {
                const float hi = fmaxf(a, b), lo = fminf(a, b);
                ll = hi + LSE_LN2 * log2f(1.0f + __builtin_amdgcn_exp2f((lo - hi) * 1.4426950408889634f));
            }
            if (out) out[(int64_t)s * n_frames + row] = ll;
            mine = (double)ll;
            hot = ll < band_hi;
        }
        mine = wave_sum_f64(mine);
        if (__builtin_amdgcn_ballot_w64(hot) != 0) mine = SR_FLUSH_POISON;     // (inf + anything finite stays inf below)
        __syncthreads();                              // the previous model's reader is done with part[]
        if (lane == 0) part[wave] = mine;
        __syncthreads();
        if (tid == 0) partial[(int64_t)blockIdx.x * n_models + s] = ((part[0] + part[1]) + part[2]) + part[3];
    }
}

// ---------------- host side ----------------

struct LastKernel {
    char name[256] = "";
};
#define g_last_kernel (per_device<LastKernel>().name)      // threads on different devices launch concurrently
const char *last_score_kernel() { return g_last_kernel; }

ScoreOptions &score_options() {
    static ScoreOptions o;
    return o;
}

struct ScoreWorkspace {
    DevBuf<double> partial;
    // Results of a pass: [U x S] sums with the U argmax values right behind them, and the pass's three counters side by side --
    // one host-bound copy and one clear each instead of two and three (a copy or a fill is ~4.5 us on the stream: 18 us of a
    // 240 us single-utterance decision, round 4).
    DevBuf<double> results;
    DevBuf<int> counters;                // [0] saturation flag of the fp16 engines, [1] pairs in the partial-product band, [4 ...] the shared-sigma
                                         // engine's exception tiles per model block
    double *sums_p(size_t n_sums) { (void)n_sums; return results.p; }
    int *argmax_p(size_t n_sums) { return reinterpret_cast<int *>(results.p + n_sums); }
    void ensure_results(size_t n_utt, size_t n_models) { results.ensure(n_utt * n_models + (n_utt + 1) / 2 + 1); }
    int *oor_p() { return counters.p; }
    int *flush_count_p() { return counters.p + 1; }
    int *exc_count_p() { return counters.p + 4; }
    DevBuf<float> frame_ll;
    DevBuf<float> ref_ll;                // split-fp16 shared-sigma engine: the reference model's per-frame LL
    DevBuf<double> ref_partial;
    DevBuf<int> exc_list;                // ... and its exception lists ({tile, listed frames} per block) + the exception pass's plan
    DevBuf<float> hy_a, hy_b;            // hybrid sets: per-frame LL of the two sub-sets
    DevBuf<int2> flush_list;             // (tile, model) pairs in the partial-product band (lse.hpp, gmm_flush.hip)
    size_t flush_min_cap = 0;            // set after an overflow: the next pass gets a list of that length
    // SCORE_HOST_DELIVER: the page-locked landing area, the last sequence number handed out, and whether the last pass's finalize
    // left `clean_n` counters at `clean_p` cleared (nothing else writes them between passes)
    PinnedBuf<char> deliver;
    void *deliver_dev = nullptr;         // the device's view of it
    unsigned deliver_seq = 0;
    const int *clean_p = nullptr;
    size_t clean_n = 0;
};
static ScoreWorkspace &ws() { return per_device<ScoreWorkspace>(); }   // one per device, leaked on purpose

// What a scoring pass needs for the partial-product band (lse.hpp): the threshold its engine compares per-frame values
// with, and the list gmm_finalize_kernel notes poisoned (tile, model) pairs in.  Nothing when the reference's clamp is off.
struct FlushPass {
    float band_hi = -INFINITY;
    int2 *list = nullptr;
    int *count = nullptr;
    int cap = 0;
};
static FlushPass prepare_flush(const SRModelSet &set, int n_tiles, int flags) {
    FlushPass fp;
    if (!(flags & 1) || (flags & SCORE_NO_FLUSH)) return fp;
    auto &w = ws();
    const size_t pairs = (size_t)std::max(1, n_tiles) * (size_t)set.host.n_models;
    size_t cap = std::min<size_t>(pairs, (size_t)1 << 20);
    if (score_options().flush_list_cap > 0) cap = (size_t)score_options().flush_list_cap;     // (testing the overflow path)
    cap = std::min<size_t>(std::max(cap, w.flush_min_cap), 0x7fffffff);
    w.flush_list.ensure(cap);
    fp.list = w.flush_list.p;
    fp.count = w.flush_count_p();          // (cleared with the pass's other counters by score_device)
    // (the capacity the pass is told is a function of ITS size, not of what the workspace happens to hold: a caller that sets a
    // piece's list aside -- multi.cpp -- then sizes its copy once; told the workspace's size, the small pieces of sr_multi's SECOND
    // call found a list grown by the first call's large piece, reallocated theirs under the pipeline and cost configs[2]'s second
    // from-host call 60 ms, round 6)
    fp.cap = (int)std::min<size_t>(cap, 0x7fffffff);
    fp.band_hi = (float)(-708.396418532264 + set.host.flush_band);
    return fp;
}

template <int DP, int F, bool PK>
static void launch_score(const ScoreArgs &a, int n_tiles, int n_groups) {
    dim3 grid((unsigned)((int64_t)n_groups * ((n_tiles + 7) / 8) * 8));   // 1-D, XCD-aware order
    hipLaunchKernelGGL((gmm_score_kernel<DP, F, PK>), grid, dim3(256), 0, ctx().stream, a.X, a.tiles,
                       a.params, a.center, a.chunks, a.group_chunk_begin, a.partial, a.frame_ll, a.n_frames,
                       a.dim, a.n_models, a.clamp, n_groups, n_tiles, a.band_hi);
}

template <int DP>
static void dispatch_f(const ScoreArgs &a, int F, bool pk, int n_tiles, int n_groups) {
    if constexpr (DP > 64) {
        return launch_score<DP, 1, false>(a, n_tiles, n_groups);     // wide rows: one frame per lane
    } else {
        if (F == 1) return launch_score<DP, 1, false>(a, n_tiles, n_groups);
        if (F == 2) return pk ? launch_score<DP, 2, true>(a, n_tiles, n_groups)
                              : launch_score<DP, 2, false>(a, n_tiles, n_groups);
        if constexpr (DP <= 40) {
            if (F == 4) return pk ? launch_score<DP, 4, true>(a, n_tiles, n_groups)
                                  : launch_score<DP, 4, false>(a, n_tiles, n_groups);
        }
        fail("frames_per_lane=%d not instantiated for dim %d", F, DP);
    }
}

static void dispatch(const ScoreArgs &a, int DP, int F, bool pk, int n_tiles, int n_groups) {
    switch (DP) {
        case 8: dispatch_f<8>(a, F, pk, n_tiles, n_groups); break;
        case 13: dispatch_f<13>(a, F, pk, n_tiles, n_groups); break;
        case 16: dispatch_f<16>(a, F, pk, n_tiles, n_groups); break;
        case 24: dispatch_f<24>(a, F, pk, n_tiles, n_groups); break;
        case 26: dispatch_f<26>(a, F, pk, n_tiles, n_groups); break;
        case 32: dispatch_f<32>(a, F, pk, n_tiles, n_groups); break;
        case 34: dispatch_f<34>(a, F, pk, n_tiles, n_groups); break;
        case 39: dispatch_f<39>(a, F, pk, n_tiles, n_groups); break;
        case 40: dispatch_f<40>(a, F, pk, n_tiles, n_groups); break;
        case 48: dispatch_f<48>(a, F, pk, n_tiles, n_groups); break;
        case 56: dispatch_f<56>(a, F, pk, n_tiles, n_groups); break;
        case 64: dispatch_f<64>(a, F, pk, n_tiles, n_groups); break;
        case 80: dispatch_f<80>(a, F, pk, n_tiles, n_groups); break;
        case 96: dispatch_f<96>(a, F, pk, n_tiles, n_groups); break;
        default: fail("no scoring kernel for padded dim %d", DP);
    }
}

static int auto_frames_per_lane(const SRBatch &b, int dp) {
    const int fmax = dp <= 40 ? 4 : dp <= 64 ? 2 : 1;
    if (b.n_utt == 0) return 1;
    const double mean_len = (double)b.n_rows / b.n_utt;
    // pick the largest F whose tiles are mostly full
    for (int f = fmax; f > 1; f >>= 1) {
        const double tile = 256.0 * f;
        const double tiles = std::ceil(mean_len / tile);
        if (mean_len / (tiles * tile) >= 0.80) return f;
    }
    return 1;
}

void upload_model_set(SRModelSet &s) {
    ensure_device();
    s.d_params.upload(s.host.params.data(), s.host.params.size());
    s.d_center0.upload(s.host.center.data(), s.host.center.size());
    s.d_chunks.upload(s.host.chunks.data(), s.host.chunks.size());
    sync_stream();
    s.device = ctx().device;
    if (s.hy_good) upload_model_set(*s.hy_good);
    if (s.hy_bad) upload_model_set(*s.hy_bad);
}

// is the set inside the split-fp16 shared-sigma engine's range?
static bool h2s_ok(const PackedH2Shared &p) {
    return !p.params.empty() && p.amp <= F16_MAX_AMP && p.pad_waste <= MFMA_MAX_PAD_WASTE &&
           p.sigma_ratio <= F16_MAX_SIGMA_RATIO && p.coef_max <= F16_MAX_COEF;
}

static bool f16_ok(const PackedSplit &p);
static bool f16_ok_fwd(const PackedSplit &p) { return f16_ok(p); }

// amp_k = sum_d ((mu_kd - centre_d) / sigma_kd)^2 with the centre the matrix-core layouts use (mean of all means)
static std::vector<std::vector<double>> mixture_amps(const std::vector<const GMM *> &models) {
    const int dim = models[0]->dim;
    std::vector<double> centre(dim, 0.0);
    size_t cnt = 0;
    for (const GMM *g : models) {
        for (int k = 0; k < g->nr_mixtures; k++)
            for (int d = 0; d < dim; d++) centre[d] += g->mean[(size_t)k * dim + d];
        cnt += (size_t)g->nr_mixtures;
    }
    for (int d = 0; d < dim; d++) centre[d] = (double)(float)(centre[d] / (double)cnt);
    std::vector<std::vector<double>> amp(models.size());
    for (size_t s = 0; s < models.size(); s++) {
        const GMM &g = *models[s];
        amp[s].assign(g.nr_mixtures, 0.0);
        for (int k = 0; k < g.nr_mixtures; k++)
            for (int d = 0; d < dim; d++) {
                const double v = (g.mean[(size_t)k * dim + d] - centre[d]) / g.sigma[(size_t)k * dim + d];
                amp[s][k] += v * v;
            }
    }
    return amp;
}

static void pack_model_set_plain(SRModelSet &s, const std::vector<const GMM *> &models);

// true when the dispatcher would send the (plainly packed) set to the vector engine because of its conditioning alone
static bool ill_conditioned_only(const SRModelSet &s) {
    const bool any_ok = (h2s_ok(s.h2s)) ||
                        (!s.shared.params.empty() && s.shared.amp <= MFMA_MAX_AMP && s.shared.pad_waste <= MFMA_MAX_PAD_WASTE) ||
                        f16_ok_fwd(s.h2) ||
                        (!s.bx3.params.empty() && s.bx3.amp <= MFMA_MAX_AMP && s.bx3.pad_waste <= MFMA_MAX_PAD_WASTE);
    if (any_ok) return false;
    const double amp = !s.bx3.params.empty() ? s.bx3.amp : !s.shared.params.empty() ? s.shared.amp : 0.0;
    const double waste = !s.bx3.params.empty() ? s.bx3.pad_waste : !s.shared.params.empty() ? s.shared.pad_waste : 1.0;
    return amp > MFMA_MAX_AMP && waste <= MFMA_MAX_PAD_WASTE;
}

void pack_model_set(SRModelSet &s, const std::vector<const GMM *> &models) {
    pack_model_set_plain(s, models);
    if (score_options().engine != 0 || s.host.dim > MAX_MATRIX_DIM || !ill_conditioned_only(s)) return;
    // ---- hybrid form: the few offending mixtures on the vector engine, the rest on the matrix cores ----
    const int dim = models[0]->dim;
    const auto amp = mixture_amps(models);
    const bool shared = models.size() > 1 && models_share_sigma_and_weights(models);
    std::vector<std::vector<char>> bad(models.size());
    for (size_t m = 0; m < models.size(); m++) {
        bad[m].assign(models[m]->nr_mixtures, 0);
        for (int k = 0; k < models[m]->nr_mixtures; k++) bad[m][k] = amp[m][k] > 0.5 * F16_MAX_AMP;     // margin: the centre moves
    }
    if (shared)     // keep the sub-sets shared-sigma: the same mixtures leave every model
        for (int k = 0; k < models[0]->nr_mixtures; k++) {
            char any = 0;
            for (size_t m = 0; m < models.size(); m++) any |= bad[m][k];
            for (size_t m = 0; m < models.size(); m++) bad[m][k] = any;
        }
    size_t n_bad = 0, n_all = 0;
    int worst = 0;
    for (size_t m = 0; m < models.size(); m++) {
        int b = 0;
        for (char c : bad[m]) b += c;
        if (b == models[m]->nr_mixtures) return;             // a model made of such mixtures only: nothing to gain
        n_bad += (size_t)b;
        n_all += (size_t)models[m]->nr_mixtures;
        worst = std::max(worst, b);
    }
    if (n_bad == 0 || (double)n_bad > HYBRID_MAX_BAD_FRACTION * (double)n_all) return;
    std::vector<GMM> good_m(models.size()), bad_m(models.size());
    for (size_t m = 0; m < models.size(); m++) {
        const GMM &g = *models[m];
        for (int side = 0; side < 2; side++) {
            GMM &o = side ? bad_m[m] : good_m[m];
            o.dim = dim;
            for (int k = 0; k < g.nr_mixtures; k++) {
                if ((bad[m][k] != 0) != (side != 0)) continue;
                o.weights.push_back(g.weights[k]);            // un-normalised on purpose: the two parts add up to the model
                o.mean.insert(o.mean.end(), g.mean.begin() + (size_t)k * dim, g.mean.begin() + (size_t)(k + 1) * dim);
                o.sigma.insert(o.sigma.end(), g.sigma.begin() + (size_t)k * dim, g.sigma.begin() + (size_t)(k + 1) * dim);
            }
            if (o.weights.empty()) {                          // a model without such mixtures: one dead mixture (weight 0 adds nothing)
                o.weights.push_back(0.0);
                o.mean.insert(o.mean.end(), g.mean.begin(), g.mean.begin() + dim);
                o.sigma.insert(o.sigma.end(), g.sigma.begin(), g.sigma.begin() + dim);
            }
            o.nr_mixtures = (int)o.weights.size();
        }
    }
    std::vector<const GMM *> gp, bp;
    for (size_t m = 0; m < models.size(); m++) {
        gp.push_back(&good_m[m]);
        bp.push_back(&bad_m[m]);
    }
    auto good = std::make_unique<SRModelSet>();
    pack_model_set_plain(*good, gp);
    if (ill_conditioned_only(*good)) return;                  // still ill conditioned without them: stay on the vector engine
    auto badset = std::make_unique<SRModelSet>();
    badset->host = pack_models(bp);                           // vector layout only
    s.hy_good = std::move(good);
    s.hy_bad = std::move(badset);
    s.hy_bad_mixtures = worst;
}

static void pack_model_set_plain(SRModelSet &s, const std::vector<const GMM *> &models) {
    s.host = pack_models(models);
    size_t n_mix = 0;
    for (const GMM *g : models) n_mix += (size_t)g->nr_mixtures;
    if (s.host.dim > MAX_MATRIX_DIM) return;                // wide rows: the vector-ALU engine only
    const bool small = n_mix <= ((size_t)1 << 16);          // every layout is a few MB at most
    const int forced = score_options().engine;
    const bool shared_ok = (int)models.size() >= SHARED_MIN_MODELS && models[0]->dim <= 48 &&   // <= 3 + 4 contraction steps: no scratch
                           models_share_sigma_and_weights(models);
    // the shared-sigma forms: the split-fp16 one when the set is within its range, else split-bf16
    // (small sets carry both, so that either can be forced and the precise re-run has its layout)
    bool h2s_fits = false;
    if (shared_ok && (small || forced == 0 || forced == 6)) {
        s.h2s = pack_models_h2_shared(models);
        h2s_fits = h2s_ok(s.h2s);
        if (!small && forced == 0 && !h2s_fits) s.h2s = PackedH2Shared();
    }
    if (shared_ok && (small || forced == 4 || (forced == 0 && !h2s_fits))) s.shared = pack_models_bx3_shared(models);
    if (small || forced == 3 || (forced == 0 && !shared_ok)) s.bx3 = pack_models_split(models, SPLIT_BF16X3);
    if (small || forced == 5 || (forced == 0 && !shared_ok)) s.h2 = pack_models_split(models, SPLIT_F16X2);
}

static void ensure_shared_layout(SRModelSet &s) {
    if (s.d_shared_params.p) return;
    s.d_shared_params.upload(s.shared.params.data(), s.shared.params.size());
    s.d_shared_blocks.upload(s.shared.blocks.data(), s.shared.blocks.size());
    s.d_shared_center.upload(s.shared.center.data(), s.shared.center.size());
    sync_stream();
}

static void ensure_h2s_layout(SRModelSet &s) {
    if (s.d_h2s_params.p) return;
    const PackedH2Shared &h = s.h2s;
    s.d_h2s_params.upload(h.params.data(), h.params.size());
    s.d_h2s_blocks.upload(h.blocks.data(), h.blocks.size());
    s.d_h2s_center.upload(h.center.data(), h.center.size());
    s.d_h2s_scale.upload(h.scale.data(), h.scale.size());
    s.d_h2s_qdesc.upload(h.q_desc.data(), h.q_desc.size());
    s.d_h2s_ldesc.upload(h.l_desc.data(), h.l_desc.size());
    s.d_h2s_ref_params.upload(h.ref.params.data(), h.ref.params.size());
    s.d_h2s_ref_chunks.upload(h.ref.chunks.data(), h.ref.chunks.size());
    // the reference pre-pass: one model, one group; its center and scale follow the set's (appended)
    const int gcb[2] = {0, (int)h.ref.chunks.size()};
    s.d_h2s_ref_gcb.upload(gcb, 2);
    s.d_h2s_ref_center.upload(h.ref.center.data(), h.ref.center.size());
    s.d_h2s_ref_scale.upload(h.ref.scale.data(), h.ref.scale.size());
    sync_stream();
}

static void ensure_h2_layout(SRModelSet &s) {
    if (s.d_h2_params.p) return;
    s.d_h2_params.upload(s.h2.params.data(), s.h2.params.size());
    s.d_h2_chunks.upload(s.h2.chunks.data(), s.h2.chunks.size());
    s.d_h2_center.upload(s.h2.center.data(), s.h2.center.size());
    s.d_h2_scale.upload(s.h2.scale.data(), s.h2.scale.size());
    sync_stream();
}

static bool f16_ok(const PackedSplit &p) {
    return !p.params.empty() && p.amp <= F16_MAX_AMP && p.pad_waste <= MFMA_MAX_PAD_WASTE &&
           p.sigma_ratio <= F16_MAX_SIGMA_RATIO && p.coef_max <= F16_MAX_COEF;
}

bool split_bf16_in_range(const SRModelSet &s) {
    return !s.bx3.params.empty() && s.bx3.amp <= MFMA_MAX_AMP && s.bx3.pad_waste <= MFMA_MAX_PAD_WASTE;
}

void ensure_bx3_layout(SRModelSet &s) {
    if (s.d_bx3_params.p && !s.bx3_stale) return;
    s.bx3_stale = false;
    s.d_bx3_params.upload(s.bx3.params.data(), s.bx3.params.size());
    s.d_bx3_chunks.upload(s.bx3.chunks.data(), s.bx3.chunks.size());
    s.d_bx3_center.upload(s.bx3.center.data(), s.bx3.center.size());
    sync_stream();
}

static ScoreResult score_hybrid(SRModelSet &set, SRBatch &feat, bool want_frame_ll, int flags, float *frame_ll_dst);

ScoreResult score_device(SRModelSet &set, SRBatch &feat, bool want_frame_ll, int flags, float *frame_ll_dst) {
    ensure_device();
    if (feat.kind != SRBatch::FEATURES) fail("scoring needs a feature batch");
    feat.bind_device();
    if (set.device != ctx().device)
        fail("model set lives on device %d, the calling thread is on device %d", set.device, ctx().device);
    if (feat.dim != set.host.dim)
        fail("feature dim %d != model dim %d", feat.dim, set.host.dim);
    if (set.hy_good && score_options().engine == 0) return score_hybrid(set, feat, want_frame_ll, flags, frame_ll_dst);
    const int S = set.host.n_models;
    const int DP = set.host.dp;
    const ScoreOptions &opt = score_options();
    // engine choice (see score.hpp): the matrix-core kernel when its layout exists, is well
    // conditioned and not mostly padding; the vector-ALU kernel otherwise or when forced
    const bool bx3_ok = !set.bx3.params.empty();
    const bool shared_ok = !set.shared.params.empty();
    const bool h2_ok = !set.h2.params.empty();
    const bool h2s_present = !set.h2s.params.empty();
    const bool precise = (flags & SCORE_PRECISE) != 0;
    bool use_bx3 = false, use_shared = false, use_h2 = false, use_h2s = false;
    auto precise_fallback = [&]() {      // best fp32-grade engine whose layout this set carries
        use_shared = shared_ok && set.shared.amp <= MFMA_MAX_AMP && set.shared.pad_waste <= MFMA_MAX_PAD_WASTE;
        if (!use_shared) use_bx3 = bx3_ok && set.bx3.amp <= MFMA_MAX_AMP && set.bx3.pad_waste <= MFMA_MAX_PAD_WASTE;
    };
    if (opt.engine == 6 && !precise) {
        if (!h2s_present) fail("split-fp16 shared-sigma engine requested but the set does not qualify (>= %d models with "
                               "identical sigma and weights, packed with that engine available)", SHARED_MIN_MODELS);
        use_h2s = true;
    } else if (opt.engine == 6) {
        precise_fallback();
    } else if (opt.engine == 5 && !precise) {
        if (!h2_ok) fail("split-fp16 engine requested but the set has no fp16 layout (sets of more than 65536 mixtures pack "
                        "only the layouts selected by score_engine when they are created)");
        use_h2 = true;
    } else if (opt.engine == 5) {
        use_bx3 = bx3_ok;             // the precise re-run of a forced fp16 engine
    } else if (opt.engine == 4) {
        if (!shared_ok) fail("shared-sigma engine requested but the set does not qualify (>= %d models with "
                             "identical sigma and weights, packed with that engine available)", SHARED_MIN_MODELS);
        use_shared = true;
    } else if (opt.engine == 3) {
        if (!bx3_ok) fail("split-bf16 engine requested but the set has no bf16x3 layout (sets of more than 65536 mixtures pack "
                         "only the layouts selected by score_engine when they are created)");
        use_bx3 = true;
    } else if (opt.engine == 0) {
        use_h2s = !precise && h2s_ok(set.h2s);
        if (!use_h2s) use_shared = shared_ok && set.shared.amp <= MFMA_MAX_AMP && set.shared.pad_waste <= MFMA_MAX_PAD_WASTE;
        if (!use_h2s && !use_shared) use_h2 = !precise && f16_ok(set.h2);
        if (!use_h2s && !use_shared && !use_h2)
            use_bx3 = bx3_ok && set.bx3.amp <= MFMA_MAX_AMP && set.bx3.pad_waste <= MFMA_MAX_PAD_WASTE;
    }
    const bool use_split = use_bx3 || use_h2;
    const PackedSplit &split = use_h2 ? set.h2 : set.bx3;
    const bool use_mat = use_split || use_shared || use_h2s;
    int F = opt.frames_per_lane ? opt.frames_per_lane : auto_frames_per_lane(feat, DP);
    if (DP > 40 && F > 2) F = 2;
    if (DP > 64) F = 1;
    // few workgroups (one utterance against one model: every E-step of a MAP enrolment): a lane that holds F frames runs
    // F times as long, so the frames go to more workgroups first (3000 frames x 1 model: 3 workgroups at F = 4, 12 at F = 1)
    if (!opt.frames_per_lane)
        while (F > 1 && ((feat.n_rows + 256 * F - 1) / (256 * F)) * (int64_t)std::max(1, S) < 2 * (int64_t)ctx().n_cu) F >>= 1;
    int FT = 1;
    if (use_split) FT = opt.mfma_ft ? std::min(opt.mfma_ft, split_max_ft(split.ks)) : 1;   // one column tile per wave won or tied every sweep
    int h2s_shape = 0;      // 0: 4-wave workgroups; 1: 12-wave workgroups (launch_score_h2_shared)
    if (use_h2s) {
        // one wide workgroup per CU (one copy of the parameter stream in LDS for all its waves) once its workgroups --
        // tile groups x model blocks, the most the grid can be cut into -- fill the chip six times over; three
        // 4-wave workgroups per CU below that (measured crossover on 201 models x 512 mixtures: 30-50 k frames;
        // at 250 k frames x 1001 models x 2048 mixtures the wide form is 25 % faster, 0.128 s against 0.169 s)
        const int64_t n32 = (feat.n_rows + 31) / 32 + feat.n_utt;     // upper bound of the 32-frame tiles
        // Round 4 (scripts/ab_h2s_small.py, 201 x 512 x 39, utterances of 300 frames): the pipelined 12-wave shape wins from
        // ~2000 frames up -- 8 utterances 0.177 against 0.196 ms, 64 utterances 0.76 against 0.95, 256 utterances 2.31 against
        // 3.09 -- and loses below (4 utterances 0.173 against 0.159: a few workgroups, latency-bound); round 3's rule ("fills
        // the chip six times over") kept the 4-wave shape up to 30-50 k frames.
        const bool wide = n32 >= 64 + feat.n_utt;
        // ... and the smallest ones -- one serving utterance: ten tiles -- take the model-split shape: a workgroup per (tile, block)
        // with the block's models dealt to its four waves (gmm_score_h2_shared.hip).  Every such workgroup streams its block's
        // images for ONE tile, so beyond one workgroup per CU the stream (L2 / fabric, 6 TB/s measured) bounds it: 300 frames
        // 0.094 against 0.115 ms, 600 frames 0.123 against 0.114, 1200 frames 0.197 against 0.115 (scripts/ab_h2s_small.py)
        // Round 6: with the images fetched straight into registers (gmm_score_h2m_kernel: no LDS stage to wait out) two such
        // workgroups per CU run side by side -- 300 frames 0.074 ms, 600 and 900 frames 0.095 against 0.112 for the 4-wave shape,
        // 1200 frames (a third workgroup per CU: a second round) 0.141 against 0.111.
        const int64_t ms_wgs = n32 * (int64_t)set.h2s.blocks.size();
        const bool tiny = ms_wgs <= (int64_t)ctx().n_cu * (h2s_msplit_direct(set.h2s.kqf, set.h2s.klf) ? 2 : 1);
        h2s_shape = opt.h2s_shape ? opt.h2s_shape - 1 : (tiny ? H2S_MSPLIT_SHAPE : wide ? H2S_PIPELINED_SHAPE : 0);
        if (h2s_shape == H2S_PIPELINED_SHAPE && !h2s_pipelined_available(set.h2s.kqf, set.h2s.klf)) h2s_shape = H2S_WIDE_SHAPE;
    }
    // the generic split-fp16 engine as ONE wide workgroup per CU (gmm_score_splitp.hip) once the batch fills the chip: the 4-wave
    // kernel re-streams every chunk per 128 frames, and the LDS-DMA that takes is what bounds it on large batches
    int splitp_w = 0, split_cpm = 0;
    if (use_h2 && FT == 1 && opt.split_shape != 1) {
        const std::vector<int> &mcb = split.model_chunk_begin;
        split_cpm = S > 0 ? mcb[1] - mcb[0] : 0;
        for (int s = 1; s < S; s++)
            if (mcb[s + 1] - mcb[s] != split_cpm) split_cpm = 0;       // models of different orders: the 4-wave kernel
        if (split_cpm > 0) {
            const int64_t n32 = (feat.n_rows + 31) / 32 + feat.n_utt;  // upper bound of the 32-frame tiles
            // Measured (profiles/r04_splitp.txt): every shape of this engine delivers the same MFMAs per second on real data -- the
            // socket's power cap sets the clock by the kernel's activity (zero-filled operands: 1.46x faster, same instructions) --
            // so the shapes differ by single percents: 8 waves (two workgroups per CU, one's frame prologue under the other's
            // chains) wins or ties from ~32 chunks per prologue up (configs[1]: 2.69 against 2.78-2.99 ms), the 4-wave kernel keeps
            // the short streams (one 256-mixture model: 0.33 against 0.38 ms) and the small batches
            const int want = opt.split_shape ? opt.split_shape : 8;
            const int w = splitp_waves(SPLIT_F16X2, split.ks, want);
            // ... and the long contractions only: with fewer than 5 steps (D < 32) a chunk is 6-12 MFMAs against the same ~60-instruction
            // update and the 4-wave kernel wins or ties (100 x 64 mixtures, 1 M frames: D = 26 2.40 against 2.45 ms, D = 20 2.02 / 2.17,
            // D = 13 1.74 / 1.82; configs[4]'s tick of 1024 windows, 20 x 256 x 13: 0.090 against 0.143 -- scripts/ab_split_shape.py)
            if (w > 0 && (opt.split_shape || (split.ks >= 5 && (int64_t)S * split_cpm >= 32 &&
                                              (n32 / w) * (int64_t)std::min(S, 16) >= (int64_t)6 * ctx().n_cu * splitp_resident_per_cu(w))))
                splitp_w = w;
        }
    }
    TileTable &tt = feat.tiles_for((use_h2s || use_split) ? 32 : use_mat ? 128 * FT : 256 * F);
    const int U = feat.n_utt;

    auto &w = ws();
    bool used_oor = false;
    const size_t n_counters = 4 + 2 * set.h2s.blocks.size();      // (shared-sigma engine: exception entries and items per block)
    w.counters.ensure(n_counters);
    // (a delivering finalize cleared them behind itself; only a delivering pass -- never one being captured into a graph -- relies on it)
    if (!((flags & SCORE_HOST_DELIVER) && w.clean_p == w.counters.p && w.clean_n >= n_counters))
        SR_HIP(hipMemsetAsync(w.counters.p, 0, n_counters * sizeof(int), ctx().stream));     // the pass's counters, all at once
    w.clean_p = nullptr;
    FinalizeDelivery dl{nullptr, nullptr, 0, 0u};
    if ((flags & SCORE_HOST_DELIVER) && !want_frame_ll && !frame_ll_dst && host_deliverable((size_t)feat.n_utt, (size_t)S)) {
        if (!w.deliver.p) {
            w.deliver.ensure(sizeof(DeliverHeader) + HOST_DELIVER_MAX_BYTES + 64, hipHostMallocCoherent | hipHostMallocMapped);
            std::memset(w.deliver.p, 0, w.deliver.n);
            SR_HIP(hipHostGetDevicePointer(&w.deliver_dev, w.deliver.p, 0));
        }
        dl.host = reinterpret_cast<DeliverHeader *>(w.deliver_dev);
        dl.counters = w.counters.p;
        dl.n_counters = (int)n_counters;
        dl.seq = ++w.deliver_seq ? w.deliver_seq : ++w.deliver_seq;      // (0 is "nothing yet")
    }
    const FlushPass fp = prepare_flush(set, tt.n_tiles, flags);
    // 0 off, 1 the reference's clamp, 2 the same with "all terms underflowed" reported as -inf (a half of a hybrid set)
    const int clamp_mode = (flags & 1) ? ((flags & SCORE_NO_FLUSH) ? 2 : 1) : 0;
    if (frame_ll_dst) want_frame_ll = true;
    w.ensure_results((size_t)std::max(1, U), (size_t)S);
    const size_t n_sums = (size_t)std::max(1, U) * S;
    if (tt.n_tiles > 0) {
        // model groups: enough workgroups to fill the chip several times over
        int G = opt.model_groups;
        if (G <= 0) {
            // enough workgroups for a short tail: >= ~16 rounds of resident ones for the vector and
            // fp32 matrix kernels; the split-bf16 kernel's workgroups are short, and every extra
            // group re-reads the frame tile, so ~6 rounds (4 resident per CU) are enough there
            const int target = use_h2s ? ctx().n_cu * h2s_resident_per_cu(set.h2s.kqf, set.h2s.klf, h2s_shape) * 6 : use_shared ? ctx().n_cu * 2 * 6
                               : splitp_w ? ctx().n_cu * splitp_resident_per_cu(splitp_w) * 8
                               : use_split ? ctx().n_cu * 4 * 6 : ctx().n_cu * 3 * 16;
            // (the split-fp16 shared-sigma engine's workgroups, and the wide generic ones, take several 32-frame tiles each)
            const int n_wg_tiles = use_h2s ? (tt.n_tiles + h2s_tiles_per_wg(h2s_shape) - 1) / h2s_tiles_per_wg(h2s_shape)
                                   : splitp_w ? (tt.n_tiles + splitp_w - 1) / splitp_w
                                   : use_split ? (tt.n_tiles + 4 * FT - 1) / (4 * FT) : tt.n_tiles;
            G = (target + n_wg_tiles - 1) / n_wg_tiles;
            if (use_h2s) {
                // Round 4: when the grid is a handful of rounds, WHICH handful matters more than having many workgroups: a
                // workgroup is a frame prologue (about three (block, mixture tile) steps' worth; scripts/debug/h2s_small_one.py
                // with a round-4 build that left the kernel after the prologue: 0.12 of 0.78 ms at 64 utterances x 300 frames) plus its blocks, and the chip runs
                // ceil(workgroups / resident) rounds of the longest one.  64 x 300 frames against 14 blocks: 14 groups = 700
                // workgroups = 3 rounds of (prologue + 1 block); 5 groups = 250 workgroups = 1 round of (prologue + 3 blocks).
                const int n_blocks = (int)set.h2s.blocks.size();
                const int64_t resident = (int64_t)ctx().n_cu * h2s_resident_per_cu(set.h2s.kqf, set.h2s.klf, h2s_shape);
                const double prologue = 3.0 / std::max(1, set.h2s.n_tiles);      // in units of one block
                double best = 0.0;
                int best_g = 1;
                for (int g = 1; g <= std::min(G, n_blocks); g++) {
                    const int64_t rounds = ((int64_t)n_wg_tiles * g + resident - 1) / resident;
                    const double cost = (double)rounds * (prologue + (double)((n_blocks + g - 1) / g));
                    if (g == 1 || cost < best * 0.999) {
                        best = cost;
                        best_g = g;
                    }
                }
                G = best_g;
            }
        }
        const int n_units = use_h2s ? (int)set.h2s.blocks.size()
                            : use_shared ? (int)set.shared.blocks.size() : S;     // what a group is a range of
        G = std::max(1, std::min(G, n_units));
        std::vector<int> gcb(G + 1);
        if (use_shared || use_h2s) {
            for (int g = 0; g <= G; g++) gcb[g] = (int)(((int64_t)g * n_units) / G);
        } else {
            const std::vector<int> &mcb = use_split ? split.model_chunk_begin : set.host.model_chunk_begin;
            for (int g = 0; g <= G; g++) {
                const int model = (int)(((int64_t)g * S) / G);
                gcb[g] = mcb[model];
            }
        }
        bool uploaded = false;
        // the group table lives with the SET (a hybrid set's two halves, or sets scored in turn, each keep theirs: no
        // re-upload -- and no stream synchronisation, which a captured serving tick could not take -- in steady state)
        const int *d_gcb = nullptr;
        for (auto &gt : set.group_tables)
            if (gt->host == gcb) d_gcb = gt->dev.p;
        if (!d_gcb) {
            constexpr size_t MAX_GROUP_TABLES = 8;
            if (set.group_tables.size() < MAX_GROUP_TABLES) {
                set.group_tables.push_back(std::make_unique<SRModelSet::GroupTable>());
                set.group_table_next = set.group_tables.size() - 1;
            }
            auto &gt = *set.group_tables[set.group_table_next];
            set.group_table_next = (set.group_table_next + 1) % MAX_GROUP_TABLES;
            sync_stream();                        // (a replaced table may still be read by a launch in flight)
            gt.host = gcb;
            gt.dev.upload(gt.host.data(), gt.host.size());
            d_gcb = gt.dev.p;
            uploaded = true;
        }
        w.partial.ensure((size_t)tt.n_tiles * S * ((use_split || use_shared || use_h2s) ? 1 : 4));
        if (want_frame_ll && !frame_ll_dst) w.frame_ll.ensure((size_t)S * feat.n_rows);
        float *const fll = !want_frame_ll ? nullptr : frame_ll_dst ? frame_ll_dst : w.frame_ll.p;

        if (use_h2s) {
            ensure_h2s_layout(set);
            const PackedH2Shared &h = set.h2s;
            used_oor = true;
            // pre-pass: the reference model's per-frame LL (natural log, no clamp) = the offset
            w.ref_ll.ensure((size_t)std::max<int64_t>(1, feat.n_rows));
            TileTable &tt_ref = feat.tiles_for(32);      // (the generic split kernel's unit: a 32-frame tile per wave)
            w.ref_partial.ensure((size_t)tt_ref.n_tiles);
            {
                MfmaLaunch r;
                r.X = feat.data.p;
                r.tiles = tt_ref.d_tiles.p;
                r.params = reinterpret_cast<const float4 *>(set.d_h2s_ref_params.p);
                r.chunks = set.d_h2s_ref_chunks.p;
                r.group_chunk_begin = set.d_h2s_ref_gcb.p;
                r.center = set.d_h2s_ref_center.p;
                r.scale = set.d_h2s_ref_scale.p;
                r.partial = w.ref_partial.p;
                r.frame_ll = w.ref_ll.p;
                r.oor_flag = w.oor_p();
                r.n_frames = feat.n_rows;
                r.dim = feat.dim;
                r.n_models = 1;
                r.clamp = 0;
                r.n_groups = 1;
                r.n_tiles = tt_ref.n_tiles;
                ScopedKernelTimer t(T_SCORE_REF);
                // (high parts only: a third of the MFMAs; an offset a few nats off is as good as an exact one, gmm_score_split.hip)
                launch_score_split(r, SPLIT_F16X1, h.ref.ks, 1);
            }
            const int n_blocks = (int)h.blocks.size();
            w.exc_list.ensure((size_t)std::max(1, tt.n_tiles) * n_blocks * 2 + (size_t)(std::max(1, tt.n_tiles) + 1) * n_blocks);
            H2sLaunch a;
            a.X = feat.data.p;
            a.tiles = tt.d_tiles.p;
            a.params = set.d_h2s_params.p;
            a.blocks = set.d_h2s_blocks.p;
            a.group_block_begin = d_gcb;
            a.center = set.d_h2s_center.p;
            a.scale = set.d_h2s_scale.p;
            a.q_desc = set.d_h2s_qdesc.p;
            a.l_desc = set.d_h2s_ldesc.p;
            a.ref_ll = w.ref_ll.p;
            a.partial = w.partial.p;
            a.frame_ll = fll;
            a.oor_flag = w.oor_p();
            a.exc_list = w.exc_list.p;
            a.exc_count = w.exc_count_p();
            a.n_blocks = n_blocks;
            a.n_frames = feat.n_rows;
            a.dim = feat.dim;
            a.n_models = S;
            a.n_mix_tiles = h.n_tiles;
            a.clamp = clamp_mode;
            a.n_groups = G;
            a.n_tiles = tt.n_tiles;
            a.log2_k = (float)std::log2((double)h.n_tiles * MT);
            a.force_exc = opt.h2s_force_exc;
            a.shape = h2s_shape;
            if (h2s_shape == 2) {      // the pipelined kernel walks work items: ragged tail tiles share a wave
                ensure_work_table(tt, opt.h2s_pack_tails != 0);
                a.tiles = tt.d_tiles_work.p;
                a.n_work = tt.n_work;
            }
            a.band_hi = fp.band_hi;
            snprintf(g_last_kernel, sizeof(LastKernel::name),
                     "%s<%d,%d,%s> (shared sigma: quadratic half once per %d models; split-fp16 MFMA, "
                     "3 products as one contraction; reference-offset log-sum-exp)", h2s_shape == 2 ? "gmm_score_h2p_kernel" : (h2s_shape == 3 && h2s_msplit_direct(h.kqf, h.klf)) ? "gmm_score_h2m_kernel" : "gmm_score_h2s_kernel",
                     h.kqf, h.klf, h2s_shape == 2 ? "waves=12, pipelined in the wave" : h2s_shape == 1 ? "waves=12" : h2s_shape == 3 ? "waves=4 on one tile, models split" : "waves=4", SHARED_SB);
            ScopedKernelTimer t(T_SCORE);
            const int n_launches = launch_score_h2_shared(a, h.kqf, h.klf);
            const size_t len = strlen(g_last_kernel);
            snprintf(g_last_kernel + len, sizeof(LastKernel::name) - len, " [%d launches per pass]", n_launches);
        } else if (use_shared) {
            ensure_shared_layout(set);
            SharedLaunch a;
            a.X = feat.data.p;
            a.tiles = tt.d_tiles.p;
            a.params = set.d_shared_params.p;
            a.blocks = set.d_shared_blocks.p;
            a.group_block_begin = d_gcb;
            a.center = set.d_shared_center.p;
            a.partial = w.partial.p;
            a.frame_ll = fll;
            a.n_frames = feat.n_rows;
            a.dim = feat.dim;
            a.n_models = S;
            a.n_mix_tiles = set.shared.n_tiles;
            a.clamp = clamp_mode;
            a.n_groups = G;
            a.n_tiles = tt.n_tiles;
            a.band_hi = fp.band_hi;
            snprintf(g_last_kernel, sizeof(LastKernel::name),
                     "gmm_score_bx3_shared_kernel<%d,%d> (shared sigma: quadratic half once per %d models; split-bf16 MFMA)",
                     set.shared.kq, set.shared.kl, SHARED_SB);
            ScopedKernelTimer t(T_SCORE);
            launch_score_bx3_shared(a, set.shared.kq, set.shared.kl);
        } else if (use_mat) {
            if (use_h2) ensure_h2_layout(set); else ensure_bx3_layout(set);
            MfmaLaunch a;
            a.X = feat.data.p;
            a.tiles = tt.d_tiles.p;
            a.params = use_h2 ? reinterpret_cast<const float4 *>(set.d_h2_params.p) : reinterpret_cast<const float4 *>(set.d_bx3_params.p);
            a.chunks = use_h2 ? set.d_h2_chunks.p : set.d_bx3_chunks.p;
            a.group_chunk_begin = d_gcb;
            a.center = use_h2 ? set.d_h2_center.p : set.d_bx3_center.p;
            if (use_h2) {
                a.scale = set.d_h2_scale.p;
                a.oor_flag = w.oor_p();
                used_oor = true;
            }
            a.partial = w.partial.p;
            a.frame_ll = fll;
            a.n_frames = feat.n_rows;
            a.dim = feat.dim;
            a.n_models = S;
            a.clamp = clamp_mode;
            a.n_groups = G;
            a.n_tiles = tt.n_tiles;
            a.band_hi = fp.band_hi;
            ScopedKernelTimer t(T_SCORE);
            if (use_h2 && splitp_w) {
                snprintf(g_last_kernel, sizeof(LastKernel::name),
                         "gmm_score_splitp_kernel<f16x2,%d,waves=%d> (3 x v_mfma_f32_32x32x16_f16 per fp32 product; log-sum-exp pipelined "
                         "under the next chunk's MFMAs)", split.ks, splitp_w);
                if (!launch_score_splitp(a, SPLIT_F16X2, split.ks, splitp_w, split_cpm))
                    fail("no wide split-fp16 kernel for %d contraction steps and %d waves", split.ks, splitp_w);
            } else if (use_h2) {
                snprintf(g_last_kernel, sizeof(LastKernel::name),
                         "gmm_score_split_kernel<f16x2,%d,%d> (3 x v_mfma_f32_32x32x16_f16 per fp32 product)", split.ks, FT);
                launch_score_split(a, SPLIT_F16X2, split.ks, FT);
            } else {
                snprintf(g_last_kernel, sizeof(LastKernel::name),
                         "gmm_score_split_kernel<bf16x3,%d,%d> (6 x v_mfma_f32_32x32x16_bf16 per fp32 product)", split.ks, FT);
                launch_score_split(a, SPLIT_BF16X3, split.ks, FT);
            }
        } else {
            ScoreArgs a;
            a.X = feat.data.p;
            a.tiles = tt.d_tiles.p;
            a.params = reinterpret_cast<const float4 *>(set.d_params.p);
            a.center = set.d_center0.p;
            a.chunks = set.d_chunks.p;
            a.group_chunk_begin = d_gcb;
            a.partial = w.partial.p;
            a.frame_ll = fll;
            a.n_frames = feat.n_rows;
            a.dim = feat.dim;
            a.n_models = S;
            a.clamp = clamp_mode;
            a.band_hi = fp.band_hi;
            ScopedKernelTimer t(T_SCORE);
            if (DP > MAX_REG_DIM) {
                snprintf(g_last_kernel, sizeof(LastKernel::name), "gmm_score_wide_kernel (vector ALU, %d slices of %d dims)", DP / WIDE_DC, WIDE_DC);
                dim3 grid((unsigned)((int64_t)G * ((tt.n_tiles + 7) / 8) * 8));
                hipLaunchKernelGGL(gmm_score_wide_kernel, grid, dim3(256), 0, ctx().stream, a.X, a.tiles, a.params, a.center, a.chunks,
                                   a.group_chunk_begin, a.partial, a.frame_ll, a.n_frames, a.dim, DP, a.n_models, a.clamp, G, tt.n_tiles,
                                   a.band_hi);
            } else {
                snprintf(g_last_kernel, sizeof(LastKernel::name), "gmm_score_kernel<%d,%d,%s> (vector ALU)", DP, F,
                         (opt.packed >= 0 && F >= 2) ? "packed" : "scalar");
                dispatch(a, DP, F, opt.packed >= 0 && F >= 2, tt.n_tiles, G);
            }
        }
        SR_HIP(hipGetLastError());
        if (uploaded) sync_stream();   // first call with this grouping only; the copy source is the workspace's own vector
    }
    if (U > 0) {
        ScopedKernelTimer t(T_FINALIZE);
        hipLaunchKernelGGL(gmm_finalize_kernel, dim3((unsigned)U), dim3(256), 0, ctx().stream,
                           w.partial.p, tt.d_utt_tile_begin.p, S, (use_split || use_shared || use_h2s) ? 1 : 4, w.sums_p(n_sums), w.argmax_p(n_sums),
                           fp.list, fp.count, fp.cap, dl);
    }
    SR_HIP(hipGetLastError());
    ScoreResult r;
    if (dl.host && U > 0) {
        r.h_deliver = reinterpret_cast<const volatile DeliverHeader *>(w.deliver.p);
        r.deliver_seq = dl.seq;
        w.clean_p = w.counters.p;
        w.clean_n = n_counters;
    }
    r.d_sums = w.sums_p(n_sums);
    r.d_argmax = w.argmax_p(n_sums);
    r.d_frame_ll = (want_frame_ll && tt.n_tiles > 0) ? (frame_ll_dst ? frame_ll_dst : w.frame_ll.p) : nullptr;
    r.d_oor = used_oor ? w.oor_p() : nullptr;
    r.d_flush_count = fp.count;
    r.d_flush_list = fp.list;
    r.flush_cap = fp.cap;
    r.tiles = &tt;
    return r;
}

// The two sub-sets of a hybrid set, then the merge (gmm_merge_kernel) and the usual finalize.
static ScoreResult score_hybrid(SRModelSet &set, SRBatch &feat, bool want_frame_ll, int flags, float *frame_ll_dst) {
    auto &w = ws();
    const int S = set.host.n_models;
    const int U = feat.n_utt;
    const size_t n = (size_t)S * (size_t)std::max<int64_t>(1, feat.n_rows);
    w.hy_a.ensure(n);
    w.hy_b.ensure(n);
    // the ill-conditioned mixtures first (vector engine: the only layout that sub-set carries), then the rest -- so
    // that the fp16 engines' saturation flag of the second call is the one left in the workspace
    // (the partial-product band is the merge's business: the merged value, against the WHOLE model's parameters)
    score_device(*set.hy_bad, feat, true, flags | SCORE_NO_FLUSH, w.hy_b.p);
    const ScoreResult good = score_device(*set.hy_good, feat, true, flags | SCORE_NO_FLUSH, w.hy_a.p);
    char good_name[sizeof(LastKernel::name)];
    snprintf(good_name, sizeof(good_name), "%s", g_last_kernel);
    TileTable &tt = feat.tiles_for(256);
    const FlushPass fp = prepare_flush(set, tt.n_tiles, flags);
    w.ensure_results((size_t)std::max(1, U), (size_t)S);
    const size_t n_sums = (size_t)std::max(1, U) * S;
    float *out = nullptr;
    if (want_frame_ll || frame_ll_dst) {
        if (!frame_ll_dst) w.frame_ll.ensure(n);
        out = frame_ll_dst ? frame_ll_dst : w.frame_ll.p;
    }
    if (tt.n_tiles > 0) {
        w.partial.ensure((size_t)tt.n_tiles * S);
        ScopedKernelTimer t(T_SCORE);
        hipLaunchKernelGGL(gmm_merge_kernel, dim3((unsigned)tt.n_tiles), dim3(256), 0, ctx().stream, w.hy_a.p, w.hy_b.p,
                           tt.d_tiles.p, S, feat.n_rows, (flags & 1) ? 1 : 0, w.partial.p, out, fp.band_hi);
        SR_HIP(hipGetLastError());
    }
    if (U > 0) {
        ScopedKernelTimer t(T_FINALIZE);
        hipLaunchKernelGGL(gmm_finalize_kernel, dim3((unsigned)U), dim3(256), 0, ctx().stream, w.partial.p,
                           tt.d_utt_tile_begin.p, S, 1, w.sums_p(n_sums), w.argmax_p(n_sums), fp.list, fp.count, fp.cap,
                           FinalizeDelivery{nullptr, nullptr, 0, 0u});
        SR_HIP(hipGetLastError());
    }
    snprintf(g_last_kernel, sizeof(LastKernel::name), "hybrid: %d ill-conditioned mixtures on the vector ALU + %.150s", set.hy_bad_mixtures, good_name);
    ScoreResult r;
    r.d_sums = w.sums_p(n_sums);
    r.d_argmax = w.argmax_p(n_sums);
    r.d_frame_ll = (out && tt.n_tiles > 0) ? out : nullptr;
    r.d_oor = good.d_oor;
    r.d_flush_count = fp.count;
    r.d_flush_list = fp.list;
    r.flush_cap = fp.cap;
    r.tiles = &tt;
    return r;
}

struct ResultStaging {
    PinnedBuf<double> sums;
    PinnedBuf<int> argmax;
    PinnedBuf<float> frame_ll;
    PinnedBuf<int> oor;
};
static ResultStaging &staging() { return per_device<ResultStaging>(); }   // leaked on purpose (no hipHostFree at exit)

// Copies the last scoring call's results to host memory through pinned staging.  With the reference's clamp on, the
// (tile, model) pairs gmm_finalize_kernel left out because a frame of theirs sits in the partial-product band (lse.hpp)
// are resolved first (gmm_flush.hip patches the device results; nothing to do, and nothing extra copied but one int,
// when there are none -- the case of real data).
bool fetch_results(SRModelSet &set, SRBatch &feat, int flags, const ScoreResult &r_in, double *sums_out, int *argmax_out,
                   float *frame_ll_out) {
    auto &st = staging();
    ScoreResult r = r_in;
    const size_t U = (size_t)feat.n_utt, S = (size_t)set.host.n_models, n_frames = (size_t)feat.n_rows;
    struct ResetCap {                       // an enlarged band list is for this batch only
        bool armed = false;
        ~ResetCap() { if (armed) ws().flush_min_cap = 0; }
    } reset_cap;
    if (r.h_deliver) {
        // SCORE_HOST_DELIVER: the pass's last workgroup wrote everything into page-locked host memory and released `seq`.
        // Poll for it (a wake-up from hipStreamSynchronize costs more than the kernels' tail); now and then ask the stream --
        // a faulted queue must not leave this thread spinning.
        const volatile DeliverHeader *h = r.h_deliver;
        for (unsigned spins = 1;; spins++) {
            if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) == r.deliver_seq) break;
            if ((spins & 0xfff) == 0) {
                const hipError_t e = hipStreamQuery(ctx().stream);
                if (e == hipSuccess) {
                    if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) == r.deliver_seq) break;
                    fail("scoring pass finished without delivering its results (sequence %u, found %u)", r.deliver_seq, h->seq);
                }
                if (e != hipErrorNotReady) SR_HIP(e);
            }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
            __builtin_ia32_pause();
#endif
        }
        if (r.d_oor && h->oor != 0) return false;
        const int n_flush = r.d_flush_count ? h->n_flush : 0;
        double *h_sums = const_cast<double *>(reinterpret_cast<const volatile double *>(h + 1));
        int *h_arg = reinterpret_cast<int *>(h_sums + U * S);
        if (n_flush == 0 || n_flush <= r.flush_cap) {
            if (n_flush) flush_resolve_host(set, feat, *r.tiles, r.d_flush_list, n_flush, h_sums, h_arg);
            if (sums_out) std::memcpy(sums_out, h_sums, U * S * sizeof(double));
            if (argmax_out) std::memcpy(argmax_out, h_arg, U * sizeof(int));
            return true;
        }
        // more pairs than the list holds: the pass again with a list of that length, through the general path below
        ws().flush_min_cap = (size_t)n_flush;
        reset_cap.armed = true;
        r = score_device(set, feat, false, flags & ~SCORE_HOST_DELIVER);
    }
    st.oor.ensure(2);
    const size_t fll_n = (frame_ll_out && r.d_frame_ll) ? S * n_frames : 0;
    const bool stage_fll = fll_n > 0 && fll_n * sizeof(float) <= ((size_t)64 << 20);
    const int *h_argmax = nullptr;            // where the staged argmax values are (behind the sums when they came in one copy)
    bool rescored = false;
    for (;;) {
        st.oor.p[0] = st.oor.p[1] = 0;
        // (the workspace keeps the two counters, and the argmax values behind the sums, side by side: one copy each)
        if (r.d_oor && r.d_flush_count == r.d_oor + 1) {
            SR_HIP(hipMemcpyAsync(st.oor.p, r.d_oor, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
        } else {
            if (r.d_oor) SR_HIP(hipMemcpyAsync(st.oor.p, r.d_oor, sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
            if (r.d_flush_count) SR_HIP(hipMemcpyAsync(st.oor.p + 1, r.d_flush_count, sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
        }
        const bool together = sums_out && argmax_out && U && (const void *)r.d_argmax == (const void *)(r.d_sums + U * S);
        h_argmax = nullptr;
        if (together) {
            st.sums.ensure(U * S + (U + 1) / 2);
            SR_HIP(hipMemcpyAsync(st.sums.p, r.d_sums, U * S * sizeof(double) + U * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
            h_argmax = reinterpret_cast<const int *>(st.sums.p + U * S);
        } else {
            if (sums_out && U) {
                st.sums.ensure(U * S);
                SR_HIP(hipMemcpyAsync(st.sums.p, r.d_sums, U * S * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
            }
            if (argmax_out && U) {
                st.argmax.ensure(U);
                SR_HIP(hipMemcpyAsync(st.argmax.p, r.d_argmax, U * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
                h_argmax = st.argmax.p;
            }
        }
        if (fll_n) {
            if (stage_fll) {
                st.frame_ll.ensure(fll_n);
                SR_HIP(hipMemcpyAsync(st.frame_ll.p, r.d_frame_ll, fll_n * sizeof(float), hipMemcpyDeviceToHost, ctx().stream));
            } else {
                SR_HIP(hipMemcpyAsync(frame_ll_out, r.d_frame_ll, fll_n * sizeof(float), hipMemcpyDeviceToHost, ctx().stream));
            }
        }
        sync_stream();
        if (r.d_oor && st.oor.p[0] != 0) return false;
        int n_flush = st.oor.p[1];
        if (n_flush == 0) break;
        if (n_flush > r.flush_cap) {
            // more pairs than the list holds (the counter kept counting): the pass again with a list of that length -- and ITS
            // sums, argmax and count staged afresh (the loop's top), so that nothing below depends on the two passes having
            // left the same bits in the same places
            if (rescored) fail("partial-product band: %d (tile, model) pairs noted, list of %d", n_flush, r.flush_cap);
            rescored = true;
            ws().flush_min_cap = (size_t)n_flush;
            reset_cap.armed = true;
            const bool own = r.d_frame_ll && r.d_frame_ll != ws().frame_ll.p;
            r = score_device(set, feat, r.d_frame_ll != nullptr, flags, own ? const_cast<float *>(r.d_frame_ll) : nullptr);
            continue;
        }
        if (!fll_n && sums_out && argmax_out && U) {
            // Sums and argmax are already here: complete the HOST copies (one more wait for the tiles' exact sums; what sr_multi's
            // pieces do) instead of patching the device's and copying everything a second time -- two waits, two uploads, two
            // kernels and a copy of all U x S sums less per call.
            // Invariant: the staged sums, the staged argmax and the list all come from the SAME pass `r` (an overflow re-score
            // restarts the loop and stages its own).  The device-resident d_sums / d_argmax stay UNPATCHED on this branch --
            // they are the workspace's, valid until the next scoring call, and nothing reads them after this one returns.
            flush_resolve_host(set, feat, *r.tiles, r.d_flush_list, n_flush, st.sums.p, const_cast<int *>(h_argmax));
            break;
        }
        flush_resolve(set, feat, *r.tiles, r.d_flush_list, n_flush, const_cast<double *>(r.d_sums),
                      const_cast<int *>(r.d_argmax), const_cast<float *>(r.d_frame_ll));
        r.d_flush_count = nullptr;          // resolved: copy the patched results out
    }
    if (sums_out && U) std::memcpy(sums_out, st.sums.p, U * S * sizeof(double));
    if (argmax_out && U) std::memcpy(argmax_out, h_argmax, U * sizeof(int));
    if (stage_fll) std::memcpy(frame_ll_out, st.frame_ll.p, fll_n * sizeof(float));
    return true;
}

void score_batch_set(SRModelSet &set, SRBatch &feat, double *sums_out, int *argmax_out,
                     float *frame_ll_out, int flags) {
    // (small result sets land in host memory by themselves: SCORE_HOST_DELIVER, score.hpp)
    // (either result alone too: the legacy ABI's score_all wants one sum, pygmm.cc:98-104, and so does every second EM iteration)
    const int deliver = (!frame_ll_out && (sums_out || argmax_out) && host_deliverable((size_t)feat.n_utt, (size_t)set.host.n_models)) ? SCORE_HOST_DELIVER : 0;
    const ScoreResult r = score_device(set, feat, frame_ll_out != nullptr, flags | deliver);
    if (fetch_results(set, feat, flags, r, sums_out, argmax_out, frame_ll_out)) return;
    // a frame left the fp16 engine's range: the whole batch again on the fp32-grade engines
    const ScoreResult r2 = score_device(set, feat, frame_ll_out != nullptr, flags | SCORE_PRECISE);
    fetch_results(set, feat, flags | SCORE_PRECISE, r2, sums_out, argmax_out, frame_ll_out);
}

}  // namespace sr

// Work items of the pipelined shared-sigma kernel over a 32-frame tile table: tiles in order, every full one an item of its own, the
// ragged tails (1000-frame utterances leave 8 of 32 columns: 2.3 % of the pass's MFMAs on dead frames) packed greedily, in order,
// up to four and up to 32 columns to an item, all packed items together at the END of the list (pack_tail_tiles, gmm_model.hpp).
// The table is padded with empty items to whole rounds of H2P_ROUND_ITEMS = 8 workgroups x 12 waves: the kernel reads
// (tiles + n_tiles)[unit] for every unit of a launched round without a bound of its own.
namespace sr {
void ensure_work_table(TileTable &tt, bool pack_tails) {
    if ((tt.n_work > 0 && tt.work_packed == pack_tails) || tt.n_tiles == 0) return;
    static_assert(sizeof(TileDesc) == sizeof(int4), "work items travel in the tile table's buffer");
    // (pack_tail_tiles, gmm_model.cpp: full tiles in order, the packed items together at the end of the list)
    std::vector<int> counts(tt.h_tiles.size());
    for (size_t t = 0; t < counts.size(); t++) counts[t] = tt.h_tiles[t].count;
    const std::vector<WorkItem> items = pack_tail_tiles(counts, tt.frames_per_tile, pack_tails);
    std::vector<int4> work(items.size());
    for (size_t i = 0; i < items.size(); i++) work[i] = make_int4(items[i].t[0], items[i].t[1], items[i].t[2], items[i].t[3]);
    tt.n_work = (int)work.size();
    tt.work_packed = pack_tails;
    work.resize(((work.size() + H2P_ROUND_ITEMS - 1) / H2P_ROUND_ITEMS) * H2P_ROUND_ITEMS, make_int4(-1, -1, -1, -1));
    std::vector<TileDesc> both(tt.h_tiles);
    both.resize(tt.h_tiles.size() + work.size());
    std::memcpy(both.data() + tt.h_tiles.size(), work.data(), work.size() * sizeof(int4));
    if (tt.stage_tiles.h.p && both.size() * sizeof(TileDesc) <= STAGED_TABLE_MAX_BYTES) {      // a rebuilt table of a reused batch
        tt.stage_work.send(tt.d_tiles_work, both.data(), both.size());
        return;
    }
    tt.d_tiles_work.upload(both.data(), both.size());
    sync_stream();
}
}  // namespace sr

sr::TileTable &SRBatch::tiles_for(int frames_per_tile) {
    sr::TileTable *found = nullptr;
    for (auto &t : tile_tables)
        if (t->frames_per_tile == frames_per_tile) {
            if (!t->stale) return *t;
            found = t.get();
        }
    // (a stale table is rebuilt where it stands: the uploads below are on the stream the kernels that read the old contents
    // were launched on, so they run behind them)
    std::unique_ptr<sr::TileTable> fresh;
    if (!found) {
        fresh = std::make_unique<sr::TileTable>();
        fresh->frames_per_tile = frames_per_tile;
    }
    sr::TileTable *const tt = found ? found : fresh.get();
    tt->stale = false;
    tt->n_work = 0;
    tt->work_packed = false;
    std::vector<sr::TileDesc> tiles;
    std::vector<int> begin(n_utt + 1, 0);
    for (int u = 0; u < n_utt; u++) {
        begin[u] = (int)tiles.size();
        for (int64_t s = offsets[u]; s < offsets[u + 1]; s += frames_per_tile) {
            sr::TileDesc td;
            td.start = s;
            td.count = (int32_t)std::min<int64_t>(frames_per_tile, offsets[u + 1] - s);
            td.utt = u;
            tiles.push_back(td);
        }
    }
    begin[n_utt] = (int)tiles.size();
    tt->n_tiles = (int)tiles.size();
    tt->h_tiles = tiles;
    if (found && tiles.size() * sizeof(sr::TileDesc) <= sr::STAGED_TABLE_MAX_BYTES && begin.size() * sizeof(int) <= sr::STAGED_TABLE_MAX_BYTES) {
        // the table of a batch that is being reused: no host wait (common.hpp: StagedUpload)
        tt->stage_tiles.send(tt->d_tiles, tiles.data(), tiles.size());
        tt->stage_begin.send(tt->d_utt_tile_begin, begin.data(), begin.size());
        return *tt;
    }
    tt->d_tiles.upload(tiles.data(), tiles.size());
    tt->d_utt_tile_begin.upload(begin.data(), begin.size());
    sr::sync_stream();
    if (fresh) tile_tables.push_back(std::move(fresh));
    return *tt;
}
