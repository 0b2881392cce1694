// em.hip -- EM / MAP training on the GPU: the engine behind train_model and
// train_model_from_ubm (src/gmm/src/pygmm.cc:63-96).  Restates
// GMMTrainerBaseline::train / ::iteration (src/gmm/src/gmm.cc:581-653, :439-531),
// init_gaussians (:306-361) with both of its initialisers -- random frames and k-means|| -- in
// kmeans_init.hip, drawing the reference's own random numbers,
// and the MAP variant GMMUBMTrainerBaseline (src/gmm/src/gmmubm.cc:29-81: means only,
// relevance 16, weights and sigmas copied from the UBM).
//
// E-step on the device in two passes over the resident frames: (1) the scoring kernel gives the
// per-frame log-likelihood; (2) em_stats_kernel recomputes the per-mixture log densities with
// the same 2-FMA form, turns them into responsibilities and accumulates, per mixture,
// N_k, sum g (x-mu_old), sum g (x-mu_old)^2 (centred on the current mean, so the variance
// update does not cancel).  The M-step is O(K*D) and runs on the host in float64 with the
// reference's formulas.
#include "score.hpp"
#include "split_prologue.hpp"
#include "split_schemes.hpp"
#include "wave_ops.hpp"

#include "../../include/pygmm_hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <future>
#include <memory>
#include <system_error>
#include <thread>

namespace sr {

constexpr float EM_MINLOG = -708.396418532264f;   // fastexp.cc:93,105: below this the reference's
                                                  // linear-domain sum is 0 -> MIN_PROB_SUM path
constexpr float LOG2E_F = 1.4426950408889634f;

// One workgroup = one tile of 256 frames (lane = frame) walked over ALL mixtures, record by
// record.  Phase A (lane = frame): log2 densities of the record's 4 mixtures -> responsibilities
// into LDS.  Phase B (thread = (mixture j, dim d)): sweep the 256 frames of the tile out of LDS.
// Per-workgroup slabs in global memory take the running sums (only this workgroup touches its
// slab, in a fixed order -> deterministic); a second kernel adds the slabs in float64.
// Few tiles (a MAP enrolment is one utterance, ~12 tiles): gridDim.y cuts the records into ranges, so that the chip is
// busy anyway -- the workgroups of one tile then write disjoint mixture ranges of the same slab.
template <int DP>
__global__ __launch_bounds__(256)
void em_stats_kernel(const float *__restrict__ X, int64_t n_frames, int dim,
                     const float4 *__restrict__ params, const float *__restrict__ center /* [DP], of the packed set */,
                     int n_records,
                     const float *__restrict__ mean_f32 /* [K_pad][DP] */,
                     const float *__restrict__ frame_ll /* natural log, no clamp */,
                     float *__restrict__ slabs /* [grid][K_pad][2*DP+1] */, int n_tiles) {
    constexpr int REC = 2 * DP + 1;
    constexpr int XS = DP + 1;                 // padded row stride: conflict-free column sweeps
    __shared__ float xs[256 * XS];
    __shared__ float gs[KB][256];
    __shared__ float4 rec_s[REC];
    const int tid = threadIdx.x;
    const int K_pad = n_records * KB;
    float *slab = slabs + (size_t)blockIdx.x * K_pad * REC;
    const int rec_per = (n_records + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_begin = (int)blockIdx.y * rec_per, r_end = min(n_records, r_begin + rec_per);
    if (r_begin >= r_end) return;              // (whole workgroup, before any barrier)
    // phase-B roles: thread -> (mixture j of the record, dim d); KB * DP of them (one per thread up to DP = 64,
    // a short loop for the wide rows)

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t frame = (int64_t)tile * 256 + tid;
        const bool valid = frame < n_frames;
        float x[DP];
        float lse2 = 0.f;
        bool live = false;
        {
            const float *src = X + (valid ? frame : 0) * dim;
#pragma unroll
            for (int dd = 0; dd < DP; dd++) {
                const float raw = (dd < dim) ? src[dd] : 0.f;
                x[dd] = raw - center[dd];                  // the densities work on x - center (gmm_model.hpp); the sums on x itself
                xs[tid * XS + dd] = valid ? raw : 0.f;
            }
            if (valid) {
                const float ll = frame_ll[frame];
                live = ll >= EM_MINLOG;        // underflowed frames carry no responsibility (gmm.cc:482-498)
                lse2 = ll * LOG2E_F;
            }
        }
        for (int r = r_begin; r < r_end; r++) {
            __syncthreads();                   // previous phase B done with gs / rec_s (and xs on the first record)
            for (int i = tid; i < REC; i += 256) rec_s[i] = params[(size_t)r * REC + i];
            __syncthreads();
            float acc[KB] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dd = 0; dd < DP; dd++) {
                const float4 p0 = rec_s[2 * dd];
                const float4 p1 = rec_s[2 * dd + 1];
                const float t0 = fmaf(x[dd], p0.x, p0.y);
                const float t1 = fmaf(x[dd], p0.z, p0.w);
                const float t2 = fmaf(x[dd], p1.x, p1.y);
                const float t3 = fmaf(x[dd], p1.z, p1.w);
                acc[0] = fmaf(t0, t0, acc[0]);
                acc[1] = fmaf(t1, t1, acc[1]);
                acc[2] = fmaf(t2, t2, acc[2]);
                acc[3] = fmaf(t3, t3, acc[3]);
            }
            const float4 cc = rec_s[2 * DP];
            gs[0][tid] = live ? __builtin_amdgcn_exp2f(cc.x - acc[0] - lse2) : 0.f;
            gs[1][tid] = live ? __builtin_amdgcn_exp2f(cc.y - acc[1] - lse2) : 0.f;
            gs[2][tid] = live ? __builtin_amdgcn_exp2f(cc.z - acc[2] - lse2) : 0.f;
            gs[3][tid] = live ? __builtin_amdgcn_exp2f(cc.w - acc[3] - lse2) : 0.f;
            __syncthreads();
            for (int role = tid; role < KB * DP; role += 256) {
                const int j = role / DP;
                const int d = role - j * DP;
                const int k = r * KB + j;
                const float mu = mean_f32[(size_t)k * DP + d];
                float sd = 0.f, sdd = 0.f, sn = 0.f;
                for (int i = 0; i < 256; i++) {
                    const float gam = gs[j][i];
                    const float dv = xs[i * XS + d] - mu;
                    const float gd = gam * dv;
                    sd += gd;
                    sdd = fmaf(gd, dv, sdd);
                    sn += gam;
                }
                float *dst = slab + (size_t)k * REC;
                dst[d] += sd;
                dst[DP + d] += sdd;
                if (d == 0) dst[2 * DP] += sn;
            }
        }
        __syncthreads();                       // xs is rewritten by the next tile
    }
}

// Rows wider than MAX_REG_DIM (the reference has no limit, gmm.cc:40-51): the same two phases with the D loop cut into slices of
// WIDE_DC dimensions, CB records (32 mixtures) at a time.  Phase A (lane = frame): the records' slice in LDS, the lane's slice of
// its row from global memory, CB x KB running distances in registers across the slices -> 32 responsibilities per frame into LDS.
// Phase B (thread = dimension d, 8 mixtures at a time): sweeps the tile's frames -- x[i][d] straight from global memory,
// consecutive threads consecutive addresses -- in frame order, as em_stats_kernel does.  Same slabs, same reduction.
constexpr int EMW_JB = 8;
__global__ __launch_bounds__(256, 2)
void em_stats_wide_kernel(const float *__restrict__ X, int64_t n_frames, int dim, int dp,
                          const float4 *__restrict__ params, const float *__restrict__ center, int n_records,
                          const float *__restrict__ mean_f32 /* [K_pad][dp] */, const float *__restrict__ frame_ll,
                          float *__restrict__ slabs /* [grid][K_pad][2*dp+1] */, int n_tiles) {
    constexpr int DC = WIDE_DC, RUN = 2 * DC;
    __shared__ float4 rec_s[CB * RUN];
    __shared__ float gs[CB * KB][256];
    const int tid = threadIdx.x;
    const int REC = 2 * dp + 1;
    const int n_dc = dp / DC;
    const int K_pad = n_records * KB;
    float *slab = slabs + (size_t)blockIdx.x * K_pad * REC;
    const int rec_per = ((n_records + (int)gridDim.y - 1) / (int)gridDim.y + CB - 1) / CB * CB;
    const int r_begin = (int)blockIdx.y * rec_per, r_end = min(n_records, r_begin + rec_per);
    if (r_begin >= r_end) return;              // (whole workgroup, before any barrier)

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t frame = (int64_t)tile * 256 + tid;
        const bool valid = frame < n_frames;
        const int cnt = (int)min((int64_t)256, n_frames - (int64_t)tile * 256);
        const float *xrow = X + (valid ? frame : 0) * dim;
        float lse2 = 0.f;
        bool live = false;
        if (valid) {
            const float ll = frame_ll[frame];
            live = ll >= EM_MINLOG;            // underflowed frames carry no responsibility (gmm.cc:482-498)
            lse2 = ll * LOG2E_F;
        }
        for (int r0 = r_begin; r0 < r_end; r0 += CB) {
            const int nr = min(CB, r_end - r0);
            float acc[CB][KB];
#pragma unroll
            for (int r = 0; r < CB; r++)
#pragma unroll
                for (int j = 0; j < KB; j++) acc[r][j] = 0.f;
            for (int dc = 0; dc < n_dc; dc++) {
                __syncthreads();               // previous slice's readers / previous phase B done
                for (int i = tid; i < nr * RUN; i += 256)
                    rec_s[i] = params[(size_t)(r0 + i / RUN) * REC + dc * RUN + (i % RUN)];
                float x[DC];
                const int d0 = dc * DC;
#pragma unroll
                for (int d = 0; d < DC; d++) x[d] = (d0 + d < dim) ? xrow[d0 + d] - center[d0 + d] : 0.f;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < CB; r++) {
                    if (r < nr) {
#pragma unroll
                        for (int d = 0; d < DC; d++) {
                            const float4 p0 = rec_s[r * RUN + 2 * d];
                            const float4 p1 = rec_s[r * RUN + 2 * d + 1];
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
            }
#pragma unroll
            for (int r = 0; r < CB; r++) {
                if (r < nr) {
                    const float4 cc = params[(size_t)(r0 + r) * REC + 2 * dp];
                    gs[r * KB + 0][tid] = live ? __builtin_amdgcn_exp2f(cc.x - acc[r][0] - lse2) : 0.f;
                    gs[r * KB + 1][tid] = live ? __builtin_amdgcn_exp2f(cc.y - acc[r][1] - lse2) : 0.f;
                    gs[r * KB + 2][tid] = live ? __builtin_amdgcn_exp2f(cc.z - acc[r][2] - lse2) : 0.f;
                    gs[r * KB + 3][tid] = live ? __builtin_amdgcn_exp2f(cc.w - acc[r][3] - lse2) : 0.f;
                }
            }
            __syncthreads();
            const float *xt = X + (int64_t)tile * 256 * dim;
            for (int d = tid; d < dim; d += 256) {
                for (int j0 = 0; j0 < nr * KB; j0 += EMW_JB) {
                    float mu[EMW_JB], sd[EMW_JB], sdd[EMW_JB];
#pragma unroll
                    for (int jj = 0; jj < EMW_JB; jj++) {
                        mu[jj] = j0 + jj < nr * KB ? mean_f32[(size_t)(r0 * KB + j0 + jj) * dp + d] : 0.f;
                        sd[jj] = 0.f;
                        sdd[jj] = 0.f;
                    }
                    for (int i = 0; i < cnt; i++) {
                        const float xv = xt[(size_t)i * dim + d];
#pragma unroll
                        for (int jj = 0; jj < EMW_JB; jj++) {
                            const float gam = gs[j0 + jj][i];
                            const float dv = xv - mu[jj];
                            const float gd = gam * dv;
                            sd[jj] += gd;
                            sdd[jj] = fmaf(gd, dv, sdd[jj]);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < EMW_JB; jj++) {
                        if (j0 + jj < nr * KB) {
                            float *dst = slab + (size_t)(r0 * KB + j0 + jj) * REC;
                            dst[d] += sd[jj];
                            dst[dp + d] += sdd[jj];
                        }
                    }
                }
            }
            if (tid < nr * KB) {
                float sn = 0.f;
                for (int i = 0; i < cnt; i++) sn += gs[tid][i];
                slab[(size_t)(r0 * KB + tid) * REC + 2 * dp] += sn;
            }
        }
        __syncthreads();                       // gs is rewritten by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The statistics on the fp64 matrix cores (round 3).  sum_i g_ik [x_i | x_i^2 | 1] is a product
// Gamma^T [K x N] . Y [N x (2D+1)], and v_mfma_f64_16x16x4_f64 accumulates it in float64 from operands that are exact
// in float64 (g: the fp32 responsibility; x, x^2: the fp32 feature and its exact square), so the sums can be taken
// about the ORIGIN -- no per-mixture centring, which is what kept the old form off the matrix cores; the host's float64
// M-step re-centres (E[x^2] - E[x]^2 costs (mu/sigma)^2 eps_64).  M = 16 mixtures, N = 16 statistic columns,
// K = 4 frames per instruction: lane l supplies A[mixture l & 15][frame l >> 4] = g and B[frame l >> 4][column l & 15] = y
// and receives D[mixture (l >> 4) + 4 r][column l & 15], r = 0..3 (the f64 layout, NOT the f32 one).
//
// A workgroup = 4 waves on ONE 64-frame tile at a time and 64 mixtures (16 per wave) for its whole life: the 16
// parameter records sit in LDS from the start; the tile's raw rows arrive transposed ([d][frame], stride 130: conflict-free
// for both access patterns below) by LDS-DMA one tile ahead.  Per tile a wave (lane = frame) evaluates its 16 mixtures'
// responsibilities in the 2-FMA form of the scoring kernel (same arithmetic as before), passes them through LDS into
// the A layout, and issues 16 frame groups x NCB column blocks of MFMAs into NCB x 4 float64 accumulators that live in
// registers until the workgroup's frame range is done.  Columns: [x, 16 per block][x^2, 16 per block][the rest of x, of
// x^2, the count] -- only the last block(s) mix kinds, so a B operand is one cvt (x) or cvt + mul (x^2).
// Per-(frame chunk) slabs in float64, added in a fixed order by em_reduce64_kernel: deterministic.
//
// Tried and measured slower (round 3, scripts/ab_em.py, K = 2048 x 400 k frames x 39 dims): an 8-wave workgroup as two teams half a
// tile apart -- one team's waves evaluating responsibilities on the vector ALU while the other's run their MFMAs, a barrier per
// phase, the next tile's rows staged through registers a phase ahead; bit-identical sums, 5.6 ms against 4.45 ms.  A tile costs this
// form 2 x max(A, B) where the form below pays A + B: the two phases do not overlap on this chip.  v_mfma_f64_16x16x4_f64 runs at
// exactly the vector ALU's float64 FMA rate (78.6 TFLOP/s both) and a wave doing only those MFMAs slows its SIMD neighbour's vector
// phase down by about as much as it gains: A + B (11 k + 13 k cycles per 128 frames x 128 mixtures) is what both forms measure,
// 23.6 k here, 29.8 k there.  What shortens this kernel is less arithmetic in phase A, not a different interleaving.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int EMM_WAVES = 4, EMM_MB = 16, EMM_WG_MIX = EMM_WAVES * EMM_MB;
constexpr int EMM_F = 2;                       // frames per lane in the responsibility phase: a parameter read serves both
constexpr int EMM_FT = 64 * EMM_F;             // frames per tile
// Row strides = 2 mod 32.  The operand reads are ds_read_b32: two 32-lane groups, bank = dword address mod 32; a group holds
// (mixture or column j = 0..15) x (frame fl = 0..1 or 2..3) at j * stride + fl + const, i.e. banks 2 j + fl: all different.
// (Stride 4 mod 64 -- the ds_read_b64 / b128 bank map -- made j and j + 8 collide: SQ_LDS_BANK_CONFLICT 55 % of the LDS cycles.)
constexpr int EMM_XS = EMM_FT + 2, EMM_GS = EMM_FT + 2;
__host__ __device__ constexpr int emm_ncb(int dp) { return 2 * (dp / 16) + (2 * (dp % 16) + 1 + 15) / 16; }

template <int DP>
__global__ __launch_bounds__(EMM_WAVES * 64, 2)
void em_stats_mfma_kernel(const float *__restrict__ X, int64_t n_frames, int dim, const float4 *__restrict__ params,
                          const float *__restrict__ center, int n_records, const float *__restrict__ frame_ll,
                          double *__restrict__ slabs /* [gridDim.y][gridDim.x * 64][NCB * 16] */, int n_tiles) {
    constexpr int REC = 2 * DP + 1;
    constexpr int NFULL = DP / 16, REM = DP % 16, NCB = emm_ncb(DP);
    constexpr int RECS_WG = EMM_WG_MIX / KB;                       // 16 records per workgroup
    extern __shared__ float4 em_lds[];
    float4 *par = em_lds;                                          // [RECS_WG * REC]
    float *xt = reinterpret_cast<float *>(par + RECS_WG * REC);    // raw rows of the tile, [d][frame], stride EMM_XS
    float *gs_all = xt + DP * EMM_XS;                              // responsibilities, per wave [mixture][frame], stride EMM_GS
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rec0 = blockIdx.x * RECS_WG;                         // first record of this workgroup
    // ---- this workgroup's parameter records -> LDS (dead records beyond the model: c = -1e30, everything else 0)
    for (int i = tid; i < RECS_WG * REC; i += EMM_WAVES * 64) {
        const int r = rec0 + i / REC;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n_records) v = params[(size_t)r * REC + (i % REC)];
        else if (i % REC == 2 * DP) v = make_float4(NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG);
        par[i] = v;
    }
    // frame range of this workgroup
    const int tiles_per = (n_tiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_begin = (int)blockIdx.y * tiles_per, t_end = min(n_tiles, t_begin + tiles_per);
    // the tile's rows -> xt[d][frame]: one LDS-DMA wave-instruction per dimension and 64 frames (lane = frame supplies its
    // own global address, LDS takes the lanes side by side), the four waves take every fourth dimension
    auto fetch_tile = [&](int tile) {
#pragma unroll
        for (int h = 0; h < EMM_F; h++) {
            int64_t f = (int64_t)tile * EMM_FT + 64 * h + lane;
            if (f >= n_frames) f = n_frames - 1;                    // (a valid address; the lane's responsibilities are 0)
            const float *src = X + f * dim;
            for (int d = wave; d < DP; d += EMM_WAVES)
                if (d < dim)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + d),
                                                     (__attribute__((address_space(3))) void *)(&xt[d * EMM_XS + 64 * h]), 4, 0, 0);
        }
    };
    if (DP != dim)                                                 // padded dimensions read as 0
        for (int i = tid; i < DP * EMM_XS; i += EMM_WAVES * 64) xt[i] = 0.f;

    // What column (16 cb + j) of this lane's B operand is.  Blocks 0 .. NFULL-1 are x_d and blocks NFULL .. 2 NFULL-1 x_d^2 of
    // the SAME d = 16 cb + j: one LDS read and one conversion serve both.  The remaining block(s) mix kinds per lane:
    // y = v (alpha + beta v) + gamma with (alpha, beta, gamma) = (1,0,0) for x, (0,1,0) for x^2, (0,0,1) for the count,
    // (0,0,0) for padding -- two fp64 FMAs, no selects.
    const int j = lane & 15, fl = lane >> 4;
    constexpr int NMIX = NCB - 2 * NFULL;
    int boff[NFULL > 0 ? NFULL : 1], moff[NMIX];
    double al[NMIX], be[NMIX], ga[NMIX];
#pragma unroll
    for (int cb = 0; cb < NFULL; cb++) boff[cb] = (16 * cb + j) * EMM_XS + fl;
#pragma unroll
    for (int mb = 0; mb < NMIX; mb++) {
        const int q = 16 * mb + j;
        int d = 0;
        al[mb] = be[mb] = ga[mb] = 0.0;
        if (q < REM) { d = 16 * NFULL + q; al[mb] = 1.0; }
        else if (q < 2 * REM) { d = 16 * NFULL + q - REM; be[mb] = 1.0; }
        else if (q == 2 * REM) ga[mb] = 1.0;
        moff[mb] = d * EMM_XS + fl;
    }
    f64x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) acc[cb] = (f64x4){0.0, 0.0, 0.0, 0.0};
    float *g = gs_all + (size_t)wave * EMM_MB * EMM_GS;
    const float4 *mypar = par + (size_t)wave * (EMM_MB / KB) * REC;

    for (int tile = t_begin; tile < t_end; tile++) {
        __syncthreads();                       // every wave is done with the previous tile's rows (and `par` is filled)
        fetch_tile(tile);
        dma_publish_barrier();                 // this tile's rows have landed for every wave
        // ---- responsibilities of this wave's 16 mixtures, lane = frames l and l + 64 (the 2-FMA form on x - centre, as
        //      the scoring kernel; one parameter read from LDS serves both frames)
        // underflowed frames (and the lanes beyond the data) carry no responsibility (gmm.cc:482-498): their "log-likelihood"
        // is +1e30, so every 2^(c - q - lse2) below is 2^-1e30 = 0 without a select for the compiler to branch around
        float lse2[EMM_F];
        float x[EMM_F][DP];
#pragma unroll
        for (int h = 0; h < EMM_F; h++) {
            const int64_t frame = (int64_t)tile * EMM_FT + 64 * h + lane;
            lse2[h] = 1.0e30f;
            if (frame < n_frames) {
                const float ll = frame_ll[frame];
                lse2[h] = ll >= EM_MINLOG ? ll * LOG2E_F : 1.0e30f;
            }
#pragma unroll
            for (int d = 0; d < DP; d++) x[h][d] = xt[d * EMM_XS + 64 * h + lane] - center[d];
        }
        // (the records never change: without an offset the compiler cannot see through, it hoists every parameter read of all
        // four records out of the TILE loop -- 1264 registers' worth -- and spills the frame)
        int roff = 0;
        asm volatile("" : "+s"(roff));
#pragma unroll 1
        for (int r = 0; r < EMM_MB / KB; r++) {
            const float4 *rec = mypar + roff + r * REC;
            float a4[EMM_F][KB];
#pragma unroll
            for (int h = 0; h < EMM_F; h++)
#pragma unroll
                for (int m = 0; m < KB; m++) a4[h][m] = 0.f;
#pragma unroll
            for (int d = 0; d < DP; d++) {
                const float4 p0 = rec[2 * d];
                const float4 p1 = rec[2 * d + 1];
#pragma unroll
                for (int h = 0; h < EMM_F; h++) {
                    const float t0 = fmaf(x[h][d], p0.x, p0.y);
                    const float t1 = fmaf(x[h][d], p0.z, p0.w);
                    const float t2 = fmaf(x[h][d], p1.x, p1.y);
                    const float t3 = fmaf(x[h][d], p1.z, p1.w);
                    a4[h][0] = fmaf(t0, t0, a4[h][0]);
                    a4[h][1] = fmaf(t1, t1, a4[h][1]);
                    a4[h][2] = fmaf(t2, t2, a4[h][2]);
                    a4[h][3] = fmaf(t3, t3, a4[h][3]);
                }
            }
            const float4 cc = rec[2 * DP];
#pragma unroll
            for (int h = 0; h < EMM_F; h++) {
                g[(r * KB + 0) * EMM_GS + 64 * h + lane] = __builtin_amdgcn_exp2f(cc.x - a4[h][0] - lse2[h]);
                g[(r * KB + 1) * EMM_GS + 64 * h + lane] = __builtin_amdgcn_exp2f(cc.y - a4[h][1] - lse2[h]);
                g[(r * KB + 2) * EMM_GS + 64 * h + lane] = __builtin_amdgcn_exp2f(cc.z - a4[h][2] - lse2[h]);
                g[(r * KB + 3) * EMM_GS + 64 * h + lane] = __builtin_amdgcn_exp2f(cc.w - a4[h][3] - lse2[h]);
            }
        }
        wave_sync();
        // ---- EMM_FT / 4 frame groups x NCB column blocks on the fp64 matrix cores
#pragma unroll 4
        for (int fg = 0; fg < EMM_FT / 4; fg++) {
            const double av = (double)g[j * EMM_GS + 4 * fg + fl];             // A[mixture j][frame 4 fg + fl]
#pragma unroll
            for (int cb = 0; cb < NFULL; cb++) {
                const double v = (double)xt[boff[cb] + 4 * fg];
                acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, v, acc[cb], 0, 0, 0);
                acc[NFULL + cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, v * v, acc[NFULL + cb], 0, 0, 0);
            }
#pragma unroll
            for (int mb = 0; mb < NMIX; mb++) {
                const double v = (double)xt[moff[mb] + 4 * fg];
                const double y = __builtin_fma(v, __builtin_fma(be[mb], v, al[mb]), ga[mb]);
                acc[2 * NFULL + mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, y, acc[2 * NFULL + mb], 0, 0, 0);
            }
        }
        wave_sync();                           // (gs is rewritten by this wave's next tile)
    }
    // ---- this workgroup's sums -> its slab: D[mixture (l >> 4) + 4 r][column l & 15]
    double *slab = slabs + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * EMM_WG_MIX * (NCB * 16);
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            slab[(size_t)(wave * EMM_MB + fl + 4 * r) * (NCB * 16) + 16 * cb + j] = acc[cb][r];
}

// dynamic LDS of em_stats_mfma_kernel<DP>: 16 parameter records, one transposed tile of rows, four waves' responsibilities
static size_t emm_lds_bytes(int DP) {
    return (size_t)(EMM_WG_MIX / KB) * (2 * DP + 1) * sizeof(float4) + (size_t)DP * EMM_XS * sizeof(float) +
           (size_t)EMM_WAVES * EMM_MB * EMM_GS * sizeof(float);
}

// ---- round 4: the responsibilities on the 16-bit matrix cores -------------------------------------------------------------------
// em_stats_mfma_kernel's phase A is 2 D fused multiply-adds per (frame, mixture) on the vector ALU, and v_mfma_f64 shares that
// ALU's rate (above): 11 k of the tile's 24 k cycles.  Here the log2 densities come from the scoring engine's own contraction
// (gmm_score_split.hip, scheme bf16x3: three bf16 parts per operand, six part products, fp32-grade without range conditions)
// against the set's split-bf16 layout -- the one the pass that produced frame_ll has just run on -- and only exp2(. - lse) stays on
// the vector ALU.  One 8-wave workgroup owns 128 mixtures (four 32-mixture tiles, A fragments resident in LDS) and walks its
// frame chunk in tiles of 128 frames:
//   phase A': wave w takes the 32 frames (w & 3) of the tile and the mixture tiles 2 (w >> 2), + 1: builds the frames' B
//             fragments from the transposed rows in LDS (split_prologue.hpp), 2 x 6 KS MFMAs, exp2, responsibilities -> LDS
//             [mixture][frame];
//   phase B : as em_stats_mfma_kernel -- wave w owns mixtures 16 w .. 16 w + 15 over all 128 frames, fp64 MFMAs from fp32-exact
//             operands into accumulators that live in registers until the chunk is done.
// Same slabs, same reduction; the statistics themselves are float64 as before.
constexpr int EMS_WAVES = 8, EMS_WG_MIX = EMS_WAVES * EMM_MB;       // 128 mixtures = 4 tiles of the split layout
__host__ __device__ constexpr int ems_lds_bytes(int ks) {
    return 4 * ks * 3 * 64 * 16 + 8 * ks * EMM_XS * 4 + EMS_WG_MIX * EMM_GS * 4;
}

template <int DP, int KS>
__global__ __launch_bounds__(EMS_WAVES * 64, 2)
void em_stats_split_kernel(const float *__restrict__ X, int64_t n_frames, int dim, const uint4 *__restrict__ frags /* the model's split-bf16 tiles */,
                           int n_mix_tiles, const float *__restrict__ center /* [8 KS], of that layout */,
                           const float *__restrict__ frame_ll,
                           double *__restrict__ slabs /* [gridDim.y][gridDim.x * 128][NCB * 16] */, int n_tiles) {
    typedef bf16x3 SC;
    constexpr int P = SC::PARTS, TILE_U4 = KS * P * 64;
    constexpr int NFULL = DP / 16, REM = DP % 16, NCB = emm_ncb(DP);
    extern __shared__ uint4 ems_lds[];
    uint4 *afr = ems_lds;                                          // [4 tiles][KS * P][64]
    float *xt = reinterpret_cast<float *>(afr + 4 * TILE_U4);      // raw rows of the tile, [slot feature][frame], stride EMM_XS
    float *gs = xt + 8 * KS * EMM_XS;                              // responsibilities [mixture][frame], stride EMM_GS
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile0 = blockIdx.x * 4;                              // first mixture tile of this workgroup
    // ---- this workgroup's A fragments -> LDS (tiles beyond the model stay unread: their responsibilities are forced to 0)
    for (int i = tid; i < 4 * TILE_U4; i += EMS_WAVES * 64) {
        const int t = tile0 + i / TILE_U4;
        afr[i] = t < n_mix_tiles ? frags[(size_t)t * TILE_U4 + (i % TILE_U4)] : make_uint4(0, 0, 0, 0);
    }
    // rows of the slots beyond dim read as 0 (the DMA below never writes them)
    for (int i = tid; i < 8 * KS * EMM_XS; i += EMS_WAVES * 64) xt[i] = 0.f;
    const int tiles_per = (n_tiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_begin = (int)blockIdx.y * tiles_per, t_end = min(n_tiles, t_begin + tiles_per);
    // The tile's rows are one contiguous run of (frames in the tile) x dim floats.  Thread t carries elements t, t + 512, ... of the
    // NEXT tile in registers from the start of the current tile's phases (one workgroup per CU: a fetch between two tiles would
    // be waited for by everybody) and scatters them into the transposed tile xt[feature][frame] when the tile begins.
    constexpr int NPF = (EMM_FT * DP + EMS_WAVES * 64 - 1) / (EMS_WAVES * 64);
    int pf_off[NPF];                                               // LDS offset of element tid + 512 i, -1 beyond the tile
    {
        int f = tid / dim, d = tid - f * dim;
        const int qs = (EMS_WAVES * 64) / dim, rs = (EMS_WAVES * 64) - qs * dim;
#pragma unroll
        for (int i = 0; i < NPF; i++) {
            pf_off[i] = f < EMM_FT ? d * EMM_XS + f : -1;
            d += rs;
            f += qs;
            if (d >= dim) {
                d -= dim;
                f++;
            }
        }
    }
    float pf[NPF], pf_ll;
    auto prefetch_tile = [&](int tile) {
        const int64_t f0 = (int64_t)tile * EMM_FT;
        const int nfl = (int)min((int64_t)EMM_FT, n_frames - f0) * dim;       // floats of the tile that exist
        const float *src = X + f0 * dim;
#pragma unroll
        for (int i = 0; i < NPF; i++) {
            const int e = tid + i * (EMS_WAVES * 64);
            pf[i] = e < nfl ? src[e] : 0.f;                          // (frames beyond the data: zeros, and no responsibility)
        }
        const int64_t frame = f0 + 32 * (wave & 3) + (lane & 31);      // the frame this lane turns into a B fragment column
        pf_ll = frame < n_frames ? frame_ll[frame] : -3.0e38f;
    };
    // phase A' roles
    const int col = lane & 31, hh = lane >> 5;
    const int ft = wave & 3, mp = wave >> 2;
    float cs[KS][4], ss[KS][4];
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            cs[ks][i] = center[8 * ks + 4 * hh + i];
            ss[ks][i] = 1.0f;
        }
    const bool live0 = tile0 + 2 * mp < n_mix_tiles, live1 = tile0 + 2 * mp + 1 < n_mix_tiles;
    // phase B roles (as em_stats_mfma_kernel)
    const int j = lane & 15, fl = lane >> 4;
    constexpr int NMIX = NCB - 2 * NFULL;
    int boff[NFULL > 0 ? NFULL : 1], moff[NMIX];
    double al[NMIX], be[NMIX], ga[NMIX];
#pragma unroll
    for (int cb = 0; cb < NFULL; cb++) boff[cb] = (16 * cb + j) * EMM_XS + fl;
#pragma unroll
    for (int mb = 0; mb < NMIX; mb++) {
        const int q = 16 * mb + j;
        int d = 0;
        al[mb] = be[mb] = ga[mb] = 0.0;
        if (q < REM) { d = 16 * NFULL + q; al[mb] = 1.0; }
        else if (q < 2 * REM) { d = 16 * NFULL + q - REM; be[mb] = 1.0; }
        else if (q == 2 * REM) ga[mb] = 1.0;
        moff[mb] = d * EMM_XS + fl;
    }
    f64x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) acc[cb] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const float *g = gs + (size_t)wave * EMM_MB * EMM_GS;

    if (t_begin < t_end) prefetch_tile(t_begin);
    for (int tile = t_begin; tile < t_end; tile++) {
        __syncthreads();                       // every wave is done with the previous tile's rows and responsibilities
#pragma unroll
        for (int i = 0; i < NPF; i++)
            if (pf_off[i] >= 0) xt[pf_off[i]] = pf[i];
        // underflowed frames (and the lanes beyond the data) carry no responsibility (gmm.cc:482-498): their "log-likelihood"
        // is +1e30, so every 2^(. - lse2) below is 0 without a select
        const float lse2 = pf_ll >= EM_MINLOG ? pf_ll * LOG2E_F : 1.0e30f;
        __syncthreads();                       // this tile's rows are in place for every wave (and `afr` is filled)
        if (tile + 1 < t_end) prefetch_tile(tile + 1);
        // ---- phase A'
        {
            float xs[KS][4];
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int i = 0; i < 4; i++) xs[ks][i] = xt[(8 * ks + 4 * hh + i) * EMM_XS + 32 * ft + col];
            typename SC::frag breg[KS][P];
            float zmax = 0.0f;
            split_fragments_from_values<SC, KS>(xs, cs, ss, dim, hh, breg, zmax);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const uint4 *at = afr + (2 * mp + t) * TILE_U4 + lane;
                f32x16 c16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    typename SC::frag a[P];
#pragma unroll
                    for (int pi = 0; pi < P; pi++) a[pi] = __builtin_bit_cast(typename SC::frag, at[(ks * P + pi) * 64]);
#pragma unroll
                    for (int pr = 0; pr < SC::NPROD; pr++) c16 = SC::mfma(a[SC::AI[pr]], breg[ks][SC::BI[pr]], c16);
                }
                // accumulator r of lane (col, hh) is mixture row 8 (r / 4) + 4 hh + r % 4 of the tile, frame column col
                const float off = (t ? live1 : live0) ? lse2 : 1.0e30f;      // (a tile beyond the model: no responsibilities)
                float *grow = gs + (size_t)(32 * (2 * mp + t) + 4 * hh) * EMM_GS + 32 * ft + col;
#pragma unroll
                for (int r = 0; r < 16; r++) grow[(8 * (r >> 2) + (r & 3)) * EMM_GS] = __builtin_amdgcn_exp2f(c16[r] - off);
            }
        }
        __syncthreads();                       // the responsibilities of all 128 mixtures x 128 frames are in LDS
        // ---- phase B: EMM_FT / 4 frame groups x NCB column blocks on the fp64 matrix cores
#pragma unroll 4
        for (int fg = 0; fg < EMM_FT / 4; fg++) {
            const double av = (double)g[j * EMM_GS + 4 * fg + fl];             // A[mixture j][frame 4 fg + fl]
#pragma unroll
            for (int cb = 0; cb < NFULL; cb++) {
                const double v = (double)xt[boff[cb] + 4 * fg];
                acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, v, acc[cb], 0, 0, 0);
                acc[NFULL + cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, v * v, acc[NFULL + cb], 0, 0, 0);
            }
#pragma unroll
            for (int mb = 0; mb < NMIX; mb++) {
                const double v = (double)xt[moff[mb] + 4 * fg];
                const double y = __builtin_fma(v, __builtin_fma(be[mb], v, al[mb]), ga[mb]);
                acc[2 * NFULL + mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, y, acc[2 * NFULL + mb], 0, 0, 0);
            }
        }
    }
    // ---- this workgroup's sums -> its slab: D[mixture (l >> 4) + 4 r][column l & 15]
    double *slab = slabs + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * EMS_WG_MIX * (NCB * 16);
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            slab[(size_t)(wave * EMM_MB + fl + 4 * r) * (NCB * 16) + 16 * cb + j] = acc[cb][r];
}

// sums the frame chunks' slabs in chunk order and unpacks the column blocks: out[k][2 DP + 1] = sum g x_d | sum g x_d^2 | sum g
template <int DP>
__global__ __launch_bounds__(256)
void em_reduce64_kernel(const double *__restrict__ slabs, int n_chunks, int n_ranges, int k_pad /* the old layout's */,
                        double *__restrict__ out, int wg_mix /* mixtures per range: EMM_WG_MIX or EMS_WG_MIX */) {
    constexpr int REC = 2 * DP + 1, NFULL = DP / 16, REM = DP % 16, NC = emm_ncb(DP) * 16;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= k_pad * REC) return;
    const int k = e / REC, c = e - k * REC;
    // statistic c of the old layout -> column of the blocked one
    const int kind = c < DP ? 0 : c < 2 * DP ? 1 : 2, d = kind == 0 ? c : kind == 1 ? c - DP : 0;
    int col;
    if (kind == 2) col = 2 * NFULL * 16 + 2 * REM;
    else if (d < 16 * NFULL) col = (kind == 0 ? 0 : NFULL * 16) + d;
    else col = 2 * NFULL * 16 + (d - 16 * NFULL) + (kind == 0 ? 0 : REM);
    const int range = k / wg_mix, kk = k - range * wg_mix;
    double acc = 0.0;
    if (range < n_ranges)
        for (int ch = 0; ch < n_chunks; ch++)
            acc += slabs[(((size_t)ch * n_ranges + range) * wg_mix + kk) * NC + col];
    out[e] = acc;
}

__global__ __launch_bounds__(256)
void em_reduce_kernel(const float *__restrict__ slabs, int n_slabs, int n_elem,
                      double *__restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elem) return;
    double acc = 0.0;
    for (int s = 0; s < n_slabs; s++) acc += (double)slabs[(size_t)s * n_elem + e];
    out[e] = acc;
}

template <int DP>
static void launch_stats(const float *X, int64_t n, int dim, const float4 *params, const float *center, int n_records,
                         const float *mean_f32, const float *frame_ll, float *slabs, int n_tiles,
                         int grid) {
    const int gy = std::max(1, std::min(n_records, (4 * ctx().n_cu + grid - 1) / grid));
    hipLaunchKernelGGL((em_stats_kernel<DP>), dim3(grid, gy), dim3(256), 0, ctx().stream, X, n, dim,
                       params, center, n_records, mean_f32, frame_ll, slabs, n_tiles);
}

static void dispatch_stats(int DP, const float *X, int64_t n, int dim, const float4 *params, const float *center,
                           int n_records, const float *mean_f32, const float *frame_ll,
                           float *slabs, int n_tiles, int grid) {
#define SR_CASE(V) case V: launch_stats<V>(X, n, dim, params, center, n_records, mean_f32, frame_ll, slabs, n_tiles, grid); break;
    switch (DP) {
        SR_CASE(8) SR_CASE(13) SR_CASE(16) SR_CASE(24) SR_CASE(26) SR_CASE(32) SR_CASE(34)
        SR_CASE(39) SR_CASE(40) SR_CASE(48) SR_CASE(56) SR_CASE(64) SR_CASE(80) SR_CASE(96)
        default: fail("no EM kernel for padded dim %d", DP);
    }
#undef SR_CASE
}

template <int DP>
static void launch_stats_mfma(const float *X, int64_t n, int dim, const float4 *params, const float *center, int n_records,
                              const float *frame_ll, double *slabs, int n_tiles64, int n_ranges, int n_chunks, int k_pad, double *out) {
    static bool attr_set[MAX_DEVICES] = {};
    if (!attr_set[ctx().device]) {
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&em_stats_mfma_kernel<DP>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)emm_lds_bytes(DP)));
        attr_set[ctx().device] = true;
    }
    hipLaunchKernelGGL((em_stats_mfma_kernel<DP>), dim3(n_ranges, n_chunks), dim3(EMM_WAVES * 64), emm_lds_bytes(DP), ctx().stream, X,
                       n, dim, params, center, n_records, frame_ll, slabs, n_tiles64);
    const int n_elem = k_pad * (2 * DP + 1);
    hipLaunchKernelGGL((em_reduce64_kernel<DP>), dim3((unsigned)((n_elem + 255) / 256)), dim3(256), 0, ctx().stream, slabs, n_chunks,
                       n_ranges, k_pad, out, EMM_WG_MIX);
}

template <int DP, int KS>
static void launch_stats_split(const float *X, int64_t n, int dim, const uint4 *frags, int n_mix_tiles, const float *center,
                               const float *frame_ll, double *slabs, int n_tiles128, int n_ranges, int n_chunks, int k_pad, double *out) {
    static_assert(8 * KS >= DP, "the transposed tile has a row per slot feature");
    static bool attr_set[MAX_DEVICES] = {};
    if (!attr_set[ctx().device]) {
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&em_stats_split_kernel<DP, KS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, ems_lds_bytes(KS)));
        attr_set[ctx().device] = true;
    }
    hipLaunchKernelGGL((em_stats_split_kernel<DP, KS>), dim3(n_ranges, n_chunks), dim3(EMS_WAVES * 64), ems_lds_bytes(KS), ctx().stream, X,
                       n, dim, frags, n_mix_tiles, center, frame_ll, slabs, n_tiles128);
    const int n_elem = k_pad * (2 * DP + 1);
    hipLaunchKernelGGL((em_reduce64_kernel<DP>), dim3((unsigned)((n_elem + 255) / 256)), dim3(256), 0, ctx().stream, slabs, n_chunks,
                       n_ranges, k_pad, out, EMS_WG_MIX);
}

// instantiated for the (padded dim, contraction steps) pairs whose LDS fits a CU (DP = 40 with dim = 40 has KS = 6: 164 KB)
static bool stats_split_available(int DP, int dim) {
    const int ks = (dim + 8) / 8;
    return DP <= 40 && 8 * ks >= DP && ems_lds_bytes(ks) <= 160 * 1024;
}
static void dispatch_stats_split(int DP, int KS, const float *X, int64_t n, int dim, const uint4 *frags, int n_mix_tiles, const float *center,
                                 const float *frame_ll, double *slabs, int n_tiles128, int n_ranges, int n_chunks, int k_pad, double *out) {
#define SR_CASE(V, W) if (DP == V && KS == W) return launch_stats_split<V, W>(X, n, dim, frags, n_mix_tiles, center, frame_ll, slabs, n_tiles128, n_ranges, n_chunks, k_pad, out);
    SR_CASE(8, 1) SR_CASE(8, 2) SR_CASE(13, 2) SR_CASE(16, 2) SR_CASE(16, 3) SR_CASE(24, 3) SR_CASE(24, 4) SR_CASE(26, 4)
    SR_CASE(32, 4) SR_CASE(32, 5) SR_CASE(34, 5) SR_CASE(39, 5)         // (DP = 40 is dim = 40: six steps, 164 KB of LDS -- not available)
#undef SR_CASE
    fail("no split-responsibility EM kernel for padded dim %d with %d contraction steps", DP, KS);
}

// the fp64 matrix-core form is instantiated for padded dims <= 40 (its LDS -- 16 parameter records, a transposed 128-frame
// tile, the responsibilities: 75 KiB at DP = 39 -- lets two workgroups share a CU); wider rows keep em_stats_kernel
static bool stats_mfma_available(int DP) { return DP <= 40; }
static size_t stats_mfma_slab_doubles(int DP, int n_ranges, int n_chunks, int wg_mix = EMM_WG_MIX) {
    return (size_t)n_chunks * n_ranges * wg_mix * (size_t)(emm_ncb(DP) * 16);
}
static void dispatch_stats_mfma(int DP, const float *X, int64_t n, int dim, const float4 *params, const float *center, int n_records,
                                const float *frame_ll, double *slabs, int n_tiles64, int n_ranges, int n_chunks, int k_pad, double *out) {
#define SR_CASE(V) case V: launch_stats_mfma<V>(X, n, dim, params, center, n_records, frame_ll, slabs, n_tiles64, n_ranges, n_chunks, k_pad, out); break;
    switch (DP) {
        SR_CASE(8) SR_CASE(13) SR_CASE(16) SR_CASE(24) SR_CASE(26) SR_CASE(32) SR_CASE(34) SR_CASE(39) SR_CASE(40)
        default: fail("no fp64 matrix-core EM kernel for padded dim %d", DP);
    }
#undef SR_CASE
}

// initialisation: kmeans_init.hip (the reference's own draws, decision for decision)
void init_gmm_like_reference(GMM &g, const float *X, long n, int dim, const Parameter &param, long seed);
// em_small.hip
bool em_small_eligible(int K, int dim, long n, const Parameter &param);
bool train_em_small(GMM &gmm, const GMM *ubm, const float *dX, long n, int dim, const Parameter &param, double relevance, int *iterations);
// em_f64.hip
bool em_f64_eligible(int K, int dim, long n, const Parameter &param);
bool train_em_f64(GMM &gmm, const GMM *ubm, const float *dX, long n, int dim, const Parameter &param, double relevance, int *iterations);
void burn_reference_rand(int count);

// The model of one EM / MAP iteration as a set: the vector-ALU layout (the statistics kernels' records) and -- round 4 -- the
// split-bf16 one, so that the posteriors' denominators and the total log-likelihood of every second iteration run on the matrix
// cores at the same fp32 grade (K = 2048 x 39, 400 k frames: 2.7 ms on the vector engine per pass); score_device falls back to
// the vector engine by itself while the model is outside that layout's range (collapsed variances early in a fit).
static void pack_em_set(SRModelSet &set, const GMM &gmm) {
    if (gmm.dim <= MAX_MATRIX_DIM && score_options().engine == 0) {
        // (the two layouts are independent functions of the model: side by side on two host threads, 1.35 -> 0.9 ms at K = 2048 x 39;
        // a speaker-sized model -- 16 x 13: microseconds of packing -- is not worth the ~40 us a thread costs to start)
        if ((size_t)gmm.nr_mixtures * gmm.dim >= 8192) {
            auto split = std::async(std::launch::async, [&gmm] { return pack_models_split({&gmm}, SPLIT_BF16X3); });
            set.host = pack_models({&gmm});
            set.bx3 = split.get();
        } else {
            set.host = pack_models({&gmm});
            set.bx3 = pack_models_split({&gmm}, SPLIT_BF16X3);
        }
    } else {
        set.host = pack_models({&gmm});
        set.bx3 = PackedSplit();
    }
}

// A fit's packed sets are RECYCLED (train_em): a re-packed set keeps its device buffers, so its layouts go up into the
// allocations of the iteration before last -- upload_model_set + ensure_bx3_layout on a fresh SRModelSet were
// six hipMalloc, two waits and, when the set of the previous iteration died, six hipFree (each one a device synchronisation) per
// iteration: most of a 16-mixture speaker model's 0.16 ms per iteration.
// (What else a set caches on the device survives a re-pack unchanged: the group tables are keyed by their content, and the per-model
// record table of gmm_flush.hip is a function of the mixture count and the dimension, which a fit does not change.)
static void upload_em_set(SRModelSet &s) {
    ensure_device();
    s.d_params.upload(s.host.params.data(), s.host.params.size());
    s.d_center0.upload(s.host.center.data(), s.host.center.size());
    {   // (the chunk table describes the layout, not the parameters: the same from one iteration to the next)
        const size_t bytes = s.host.chunks.size() * sizeof(ChunkDesc);
        const char *img = reinterpret_cast<const char *>(s.host.chunks.data());
        if (!s.d_chunks.p || s.chunks_image.size() != bytes || std::memcmp(s.chunks_image.data(), img, bytes) != 0) {
            s.d_chunks.upload(s.host.chunks.data(), s.host.chunks.size());
            s.chunks_image.assign(img, img + bytes);
        }
    }
    // the matrix-core layout goes up when a kernel asks for it (ensure_bx3_layout), into the same buffers: a speaker-sized model
    // (16 mixtures: half of every 32-mixture tile would be padding) stays on the vector engine and never does
    s.bx3_stale = true;
    sync_stream();
    s.device = ctx().device;
}

struct EmWorkspace {
    DevBuf<float> slabs, mean_f32;
    DevBuf<double> stats, slabs64;
    PinnedBuf<double> h_stats;        // where an iteration's sums land on the host
};
static int &em_stats_engine_option() {
    // 0 = automatic (a speaker-sized fit whole in one launch, em_small.hip; short data through em_f64.hip; else fp64 matrix cores where instantiated, responsibilities
    // on the 16-bit ones where the model allows), 1 = the vector-ALU form always, 2 = fp64 matrix cores with the responsibilities on the
    // vector ALU (round 3's), 3 = automatic among the iteration-at-a-time engines (no whole-fit launch, no float64 iteration engine)
    static int v = 0;
    return v;
}
void set_em_stats_engine(int v) { em_stats_engine_option() = v; }
static std::atomic<int> g_last_stats_engine{0};
int last_em_stats_engine() { return g_last_stats_engine.load(); }
// The reference's trainers leave traces a drop-in user may be relying on: the parameter block on stdout at every train_model*
// call (pygmm.cc:31-41, :64, :88) and, after every second iteration, the model written to
// ./gmm-training-intermediate-dump.model with two lines around it (gmm.cc:622-630, unconditional: `if (true || ...)`).
// Off by default (a library that writes into the working directory unasked is a defect to most hosts);
// sr_set_option("reference_side_effects", 1) turns both on.
static int &reference_side_effects_option() {
    static int v = 0;
    return v;
}
void set_reference_side_effects(int v) { reference_side_effects_option() = v; }
int reference_side_effects() { return reference_side_effects_option(); }
static EmWorkspace &ews() { return per_device<EmWorkspace>(); }

int train_em(GMM &gmm, const GMM *ubm, const float *X, long n, int dim, const Parameter &param,
             long seed) {
    ensure_device();
    if (n <= 0) fail("X.size() == 0");                       // gmm.cc:582-586
    if (dim <= 0) fail("bad dimension %d", dim);
    if (param.verbosity >= 1 && !reference_side_effects_option())       // (with the reference's own block on, this summary would be extra)
        printf("nr_instance: %ld nr_dim: %d nr_mixture: %d min_covar: %f threshold: %f nr_iteration: %d "
               "init_with_kmeans: %d\n", n, dim, gmm.nr_mixtures, param.min_covar, param.threshold,
               param.nr_iteration, param.init_with_kmeans);
    if (ubm) {
        if (!ubm->trained()) fail("UBM has no parameters");
        if (ubm->dim != dim) fail("UBM dim %d != data dim %d", ubm->dim, dim);
        gmm.nr_mixtures = ubm->nr_mixtures;                  // gmm_replace_with, gmmubm.cc:29-38
        gmm.dim = ubm->dim;
        gmm.weights = ubm->weights;
        gmm.mean = ubm->mean;
        gmm.sigma = ubm->sigma;
        gmm.drop_single();
        if (seed < 0) burn_reference_rand(1 + gmm.nr_mixtures);   // the legacy symbol: keep libc's stream in step with the reference
    } else if (param.init_with_kmeans < 0 && gmm.trained() && gmm.dim == dim) {
        // extension: warm start from the handle's current parameters (no re-initialisation)
    } else {
        if (gmm.nr_mixtures <= 0) fail("GMM has no mixture count");
        if (n < 2) fail("need at least 2 frames to initialise the variances");
        init_gmm_like_reference(gmm, X, n, dim, param, seed);
    }
    const int K = gmm.nr_mixtures;
    const double relevance = 16.0;                           // gmm.hh:118-120
    const double min_sigma = std::sqrt(param.min_covar);

    // frames resident on the device for the whole fit
    SRBatch feat;
    feat.kind = SRBatch::FEATURES;
    feat.n_utt = 1;
    feat.dim = dim;
    feat.n_rows = n;
    feat.offsets = {0, (int64_t)n};
    feat.data.upload(X, (size_t)n * dim);
    feat.d_offsets.upload(feat.offsets.data(), feat.offsets.size());
    // a speaker-sized fit: every iteration, the stop rule included, in ONE launch (em_small.hip; last_em_stats_engine() = 4).  Not with
    // the reference's side effects on (a model file after every second iteration) nor with the per-phase trace (verbosity >= 2).
    if (em_stats_engine_option() == 0 && !reference_side_effects_option() && em_small_eligible(K, dim, n, param)) {
        int iterations = 0;
        if (train_em_small(gmm, ubm, feat.data.p, n, dim, param, relevance, &iterations)) {
            g_last_stats_engine.store(4);
            return iterations;
        }
    } else if (em_stats_engine_option() == 0 && !reference_side_effects_option() && em_f64_eligible(K, dim, n, param)) {
        // short data, a model of any size (a speaker's MAP enrolment from a large UBM): float64 iterations on the device, nothing
        // packed, nothing redone on the host (em_f64.hip; last_em_stats_engine() = 5)
        int iterations = 0;
        if (train_em_f64(gmm, ubm, feat.data.p, n, dim, param, relevance, &iterations)) {
            g_last_stats_engine.store(5);
            return iterations;
        }
    }
    sync_stream();

    const int n_tiles = (int)((n + 255) / 256);
    // (wide rows: a slab is K x (2 D + 1) floats per workgroup column -- one per CU, the records cut along gridDim.y instead)
    const int grid = std::min(n_tiles, ctx().n_cu * (dim > MAX_REG_DIM ? 1 : 4));
    auto &w = ews();

    double last_ll = -std::numeric_limits<double>::max();
    int it = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool trace = param.verbosity >= 2;                 // phase times of every iteration on stdout
    std::shared_ptr<SRModelSet> carried;
    // the fit's own packed sets, recycled (upload_em_set): at most two are alive at a time -- the one an E-step reads and the one the
    // total log-likelihood of every second iteration is taken under, which the next E-step then reads
    std::vector<std::shared_ptr<SRModelSet>> spare;
    auto take_set = [&]() {
        if (spare.empty()) return std::make_shared<SRModelSet>();
        std::shared_ptr<SRModelSet> p = std::move(spare.back());
        spare.pop_back();
        return p;
    };
    for (; it < param.nr_iteration; it++) {
        // ---- E-step ----
        const double t0 = now();
        // (the model as the previous iteration left it is already packed and resident when that iteration computed its total
        // log-likelihood: every second one, gmm.cc:622-623)
        std::shared_ptr<SRModelSet> set_owner = std::move(carried);
        carried.reset();
        bool fresh = !set_owner;
        bool own = true;                   // (the UBM handle's set is not this fit's to recycle)
        if (fresh && ubm && it == 0) {
            // MAP: the first E-step runs on the UBM's own parameters (gmm_replace_with, gmmubm.cc:29-38), the same for every
            // speaker enrolled from it -- its handle's packed set serves them all (K = 2048: 1.5 ms of packing per speaker)
            set_owner = single_model_set(ubm);
            fresh = false;
            own = false;
        }
        if (fresh) {
            set_owner = take_set();
            pack_em_set(*set_owner, gmm);
        }
        SRModelSet &set = *set_owner;
        const double t1 = now();
        if (fresh) upload_em_set(set);
        if (trace) sync_stream();
        const double t2 = now();
        const int DP = set.host.dp;
        const int n_records = (K + KB - 1) / KB;
        const int K_pad = n_records * KB;
        const int REC = 2 * DP + 1;
        const ScoreResult sres = score_device(set, feat, true, SCORE_PRECISE);
        if (trace) sync_stream();
        const double t3 = now();
        const size_t n_elem = (size_t)K_pad * REC;
        w.stats.ensure(n_elem);
        const int stats_engine = em_stats_engine_option();
        const bool use_mfma = stats_mfma_available(DP) && stats_engine != 1;
        // round 4: responsibilities on the 16-bit matrix cores when the model is inside the split-bf16 layout's range -- the
        // engine score_device has just taken for the denominators (em_stats_engine = 2 keeps them on the vector ALU)
        const bool use_split = use_mfma && (stats_engine == 0 || stats_engine == 3) && split_bf16_in_range(set) && stats_split_available(DP, dim);
        g_last_stats_engine.store(use_split ? 3 : use_mfma ? 2 : 1);
        if (use_split) {
            ensure_bx3_layout(set);
            const int n_mix_tiles = (int)set.bx3.chunks.size();
            const int n_ranges = (n_mix_tiles + 3) / 4;
            const int n_tiles128 = (int)((n + EMM_FT - 1) / EMM_FT);
            const int n_chunks = std::max(1, std::min(n_tiles128, (4 * ctx().n_cu + n_ranges - 1) / n_ranges));
            w.slabs64.ensure(stats_mfma_slab_doubles(DP, n_ranges, n_chunks, EMS_WG_MIX));
            ScopedKernelTimer t(T_ESTEP);
            dispatch_stats_split(DP, set.bx3.ks, feat.data.p, n, dim, reinterpret_cast<const uint4 *>(set.d_bx3_params.p), n_mix_tiles,
                                 set.d_bx3_center.p, sres.d_frame_ll, w.slabs64.p, n_tiles128, n_ranges, n_chunks, K_pad, w.stats.p);
        } else if (use_mfma) {
            // sums about the origin on the fp64 matrix cores (em_stats_mfma_kernel); re-centred below
            const int n_ranges = (n_records + EMM_WG_MIX / KB - 1) / (EMM_WG_MIX / KB);
            const int n_tiles64 = (int)((n + EMM_FT - 1) / EMM_FT);
            const int n_chunks = std::max(1, std::min(n_tiles64, (4 * ctx().n_cu + n_ranges - 1) / n_ranges));
            w.slabs64.ensure(stats_mfma_slab_doubles(DP, n_ranges, n_chunks));
            ScopedKernelTimer t(T_ESTEP);
            dispatch_stats_mfma(DP, feat.data.p, n, dim, reinterpret_cast<const float4 *>(set.d_params.p), set.d_center0.p, n_records,
                                sres.d_frame_ll, w.slabs64.p, n_tiles64, n_ranges, n_chunks, K_pad, w.stats.p);
        } else {
            std::vector<float> mean_f32((size_t)K_pad * DP, 0.f);
            for (int k = 0; k < K; k++)
                for (int d = 0; d < dim; d++) mean_f32[(size_t)k * DP + d] = (float)gmm.mean[(size_t)k * dim + d];
            w.mean_f32.upload(mean_f32.data(), mean_f32.size());
            w.slabs.ensure((size_t)grid * n_elem);
            SR_HIP(hipMemsetAsync(w.slabs.p, 0, (size_t)grid * n_elem * sizeof(float), ctx().stream));
            {
                ScopedKernelTimer t(T_ESTEP);
                if (DP > MAX_REG_DIM) {
                    const int gy = std::max(1, std::min((n_records + CB - 1) / CB, (4 * ctx().n_cu + grid - 1) / grid));
                    hipLaunchKernelGGL(em_stats_wide_kernel, dim3(grid, gy), dim3(256), 0, ctx().stream, feat.data.p, (int64_t)n, dim, DP,
                                       reinterpret_cast<const float4 *>(set.d_params.p), set.d_center0.p, n_records, w.mean_f32.p,
                                       sres.d_frame_ll, w.slabs.p, n_tiles);
                } else {
                    dispatch_stats(DP, feat.data.p, n, dim, reinterpret_cast<const float4 *>(set.d_params.p), set.d_center0.p,
                                   n_records, w.mean_f32.p, sres.d_frame_ll, w.slabs.p, n_tiles, grid);
                }
            }
            SR_HIP(hipGetLastError());
            hipLaunchKernelGGL(em_reduce_kernel, dim3((unsigned)((n_elem + 255) / 256)), dim3(256), 0,
                               ctx().stream, w.slabs.p, grid, (int)n_elem, w.stats.p);
        }
        SR_HIP(hipGetLastError());
        w.h_stats.ensure(n_elem);
        double *const stats = w.h_stats.p;
        w.stats.download(stats, n_elem);
        sync_stream();
        if (use_mfma) {
            // T1 = sum g x, T2 = sum g x^2, N = sum g  ->  the centred sums the M-step below works on, about the fp32 mean
            // the vector form centres on: sum g (x - m) = T1 - N m, sum g (x - m)^2 = T2 - 2 m T1 + N m^2 (float64)
            for (int k = 0; k < K; k++) {
                double *st = stats + (size_t)k * REC;
                const double N = st[2 * DP];
                for (int d = 0; d < dim; d++) {
                    const double m = (double)(float)gmm.mean[(size_t)k * dim + d];
                    const double t1 = st[d], t2 = st[DP + d];
                    st[d] = t1 - N * m;
                    st[DP + d] = t2 - 2.0 * m * t1 + N * m * m;
                }
            }
        }
        const double t4 = now();

        // Mixtures without support: a responsibility below fp32's range (~1e-38) is 0 on the device, while the
        // reference's float64 keeps it down to DBL_MIN -- its N_k is then tiny but not 0, and the mixture's mean
        // jumps to the responsibility-weighted mean of the frames (gmm.cc:396-412) instead of staying put.  Such
        // mixtures (rare: more mixtures than the data supports) get their three sums again on the host, in float64,
        // with the reference's rules (terms below DBL_MIN are 0; frames whose likelihood underflowed carry none).
        {
            std::vector<int> weak;
            for (int k = 0; k < K; k++)
                if (stats[(size_t)k * REC + 2 * DP] < 1e-25) weak.push_back(k);
            if (!weak.empty()) {
                std::vector<float> ll((size_t)n);
                SR_HIP(hipMemcpyAsync(ll.data(), sres.d_frame_ll, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx().stream));
                sync_stream();
                const double SQRT_2_PI = 2.5066282746310002, MINLOG = -708.396418532264;
                const float *ll_p = ll.data();
                auto redo = [&, ll_p](int k) {
                    double *st = stats + (size_t)k * REC;
                    for (int e = 0; e < REC; e++) st[e] = 0.0;
                    const double *mu = gmm.mean.data() + (size_t)k * dim, *sg = gmm.sigma.data() + (size_t)k * dim;
                    double c = gmm.weights[k] > 0 ? std::log(gmm.weights[k]) : -INFINITY;
                    for (int d = 0; d < dim; d++) c -= std::log(SQRT_2_PI * sg[d]);
                    for (long i = 0; i < n; i++) {
                        if (!(ll_p[i] >= (float)MINLOG)) continue;
                        const float *x = X + (size_t)i * dim;
                        double lp = c;
                        for (int d = 0; d < dim; d++) {
                            const double v = ((double)x[d] - mu[d]) / sg[d];
                            lp -= 0.5 * v * v;
                        }
                        if (!(lp >= MINLOG)) continue;
                        const double gam = std::exp(lp - (double)ll_p[i]);
                        for (int d = 0; d < dim; d++) {
                            const double dv = (double)x[d] - (double)(float)mu[d];       // centred on the fp32 mean the device used
                            st[d] += gam * dv;
                            st[DP + d] += gam * dv * dv;
                        }
                        st[2 * DP] += gam;
                    }
                };
                // A large UBM adapted on a short utterance leaves MANY mixtures without support (2048 mixtures, 3000 frames: this
                // loop was 4.6 of an iteration's 6.5 ms -- one division per (mixture, frame, dimension) on one core).  The mixtures
                // are independent -- each writes its own row of sums -- so they are dealt to a few host threads: same operations
                // in the same order per mixture, same bits for any number of threads.
                const size_t work = weak.size() * (size_t)n * (size_t)dim;
                const int n_thr = work < ((size_t)1 << 20) ? 1 : (int)std::max<size_t>(1, std::min<size_t>({weak.size(), (size_t)std::thread::hardware_concurrency(), (size_t)16}));
                if (n_thr <= 1) {
                    for (int k : weak) redo(k);
                } else {
                    std::atomic<size_t> next{0};
                    auto worker = [&]() {
                        for (size_t j; (j = next.fetch_add(1)) < weak.size();) redo(weak[j]);
                    };
                    std::vector<std::future<void>> helpers;
                    for (int t = 1; t < n_thr; t++) {
                        try {
                            helpers.push_back(std::async(std::launch::async, worker));
                        } catch (const std::system_error &) {      // (no thread to be had: the others take its share)
                            break;
                        }
                    }
                    worker();
                    for (auto &h : helpers) h.get();
                }
            }
        }

        // ---- M-step (float64, host; O(K*D)) ----
        std::vector<double> Nk(K);
        for (int k = 0; k < K; k++) {
            double v = stats[(size_t)k * REC + 2 * DP];
            if (v == 0) v = 1e-6;                            // min_n_k, gmm.cc:502-509
            Nk[k] = v;
        }
        if (!ubm) {                                          // update_weights, gmm.cc:388-394
            double wsum = 0;
            for (int k = 0; k < K; k++) {
                gmm.weights[k] = Nk[k] / (double)n;
                wsum += gmm.weights[k];
            }
            for (int k = 0; k < K; k++) gmm.weights[k] /= wsum;
        }
        for (int k = 0; k < K; k++) {
            for (int d = 0; d < dim; d++) {
                const double sd = stats[(size_t)k * REC + d];
                const double sdd = stats[(size_t)k * REC + DP + d];
                const double mu_old = gmm.mean[(size_t)k * dim + d];
                const double shift = sd / Nk[k];             // E_k[x] - mu_old
                if (ubm) {                                   // update_means, gmmubm.cc:53-74
                    const double alpha = Nk[k] / (Nk[k] + relevance);
                    gmm.mean[(size_t)k * dim + d] =
                        alpha * (mu_old + shift) + (1 - alpha) * ubm->mean[(size_t)k * dim + d];
                } else {                                     // gmm.cc:396-412, :415-437
                    gmm.mean[(size_t)k * dim + d] = mu_old + shift;
                    // sum g (x - mu_new)^2 = sdd - 2 shift sd + shift^2 N = sdd - N shift^2
                    double var = sdd / Nk[k] - shift * shift;
                    if (var < 0) var = 0;
                    gmm.sigma[(size_t)k * dim + d] = std::max(min_sigma, std::sqrt(var));
                }
            }
        }
        gmm.drop_single();
        if (own) spare.push_back(std::move(set_owner));       // (the E-step's kernels ended before the sums came back)
        if (trace)
            printf("iter %d: pack %.2f ms, upload %.2f ms, posteriors' denominators %.2f ms, statistics %.2f ms, M-step %.2f ms\n", it,
                   (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (now() - t4) * 1e3);

        if (it % 2 == 0) continue;                           // gmm.cc:622-623
        if (reference_side_effects_option()) {               // gmm.cc:624-630
            const char *dump_file = "gmm-training-intermediate-dump.model";
            printf("dumping model to %s ...\n", dump_file);
            const std::string text = gmm_format_text(gmm);
            if (FILE *f = fopen(dump_file, "w")) {
                fwrite(text.data(), 1, text.size(), f);
                fclose(f);
            }
            printf("model dumped to %s ...\n", dump_file);
        }
        // total log-likelihood under the updated model (gmm.cc:631-641), reference clamp on
        carried = take_set();
        SRModelSet &set2 = *carried;
        pack_em_set(set2, gmm);
        upload_em_set(set2);
        double ll = 0.0;
        score_batch_set(set2, feat, &ll, nullptr, nullptr, SR_CLAMP_COMPAT | SCORE_PRECISE);
        if (param.verbosity >= 1) printf("iter %d: ll %lf\n", it, ll);
        const double ll_diff = ll - last_ll;
        if (std::fabs(ll_diff) / std::fabs(ll) < param.threshold && ll_diff < param.threshold) {
            it++;
            break;                                           // gmm.cc:643-650
        }
        last_ll = ll;
    }
    return it;
}

}  // namespace sr
