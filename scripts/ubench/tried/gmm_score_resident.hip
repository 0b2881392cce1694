// gmm_score_resident.hip -- the split-fp16 scoring engine (gmm_score_split.hip, scheme f16x2; math of gmm.cc:176-202,
// :237-244, :533-569) for model sets SMALL ENOUGH TO LIVE IN LDS: BASELINE.json's north_star point (one 256-mixture x
// 39-dim model), a handful of small speaker models, the reference-offset pre-pass of the shared-sigma engine.
//
// gmm_score_split_kernel streams every mixture tile through LDS once per 128-frame workgroup: at K = 256 a wave gets
// 15 MFMAs (480 matrix-pipe cycles) of work between two workgroup barriers, each of which waits for an LDS-DMA stage
// with ~1 us of latency, and builds its frame operands (40 values split into fp16 parts: ~1000 vector cycles) for only
// 120 MFMAs.  Measured on the 256 x 39 point: 0.33 ms per 2 M frames, matrix pipe 30 % busy.
//
// Here the WHOLE parameter image (one tile = KS x 2 KiB; 80 KiB at K = 256, D = 39) is loaded into LDS once per
// workgroup -- 16 waves, one workgroup per CU, persistent -- and every wave then walks 32-frame tiles on its own:
// operands, all mixture tiles straight from LDS (ds_read_b128 per fragment, no barrier, no DMA in the loop), online
// log-sum-exp, close.  A wave never waits for another one after the prologue.
//
// Layout, arithmetic, underflow semantics, saturation flag and outputs are gmm_score_split_kernel<f16x2>'s (same
// PackedSplit image, same tile order): per-frame values are bit-identical to that kernel's; utterance sums are added
// per 32-frame tile here (per 128-frame tile there), in a fixed order either way.
#include "lse.hpp"
#include "score.hpp"
#include "wave_ops.hpp"

#include <algorithm>

namespace sr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int RES_WAVES = 16;

struct ResArgs {
    const float *X;
    const TileDesc *tiles;          // 32-frame tiles
    const uint4 *params;            // PackedSplit (f16x2) image: n_chunks tiles of KS * 2 * 64 fragments
    const ChunkDesc *chunks;
    const float *center, *scale;
    double *partial;                // [tile][model]
    float *frame_ll;
    int *oor_flag;
    int64_t n_frames;
    int dim, n_models, n_chunks, n_tiles, clamp;
    float band_hi;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

// bytes of one wave's frame slab: 32 rows of `dim` floats, rounded up to whole 256-byte LDS-DMA pieces
__host__ __device__ constexpr int res_slab_bytes(int dim) { return (32 * dim * 4 + 255) / 256 * 256; }

template <int KS>
__global__ __launch_bounds__(RES_WAVES * 64)
void gmm_score_res_kernel(const ResArgs a) {
    constexpr int P = 2;
    constexpr int TILE_U4 = KS * P * 64;
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;
    const int hh = lane >> 5;
    // LDS: [parameter image: n_chunks tiles][RES_WAVES frame slabs]
    const int slab_bytes = res_slab_bytes(a.dim);
    float *slab = reinterpret_cast<float *>(reinterpret_cast<char *>(lds + (size_t)a.n_chunks * TILE_U4) + (size_t)wave * slab_bytes);

    // ---- the parameter image -> LDS, once: pieces of 64 fragments (1 KiB) by LDS-DMA, a wave takes every 16th
    const int n_pieces = a.n_chunks * KS * P;
    for (int p = wave; p < n_pieces; p += RES_WAVES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.params + (size_t)p * 64 + lane),
                                         (__attribute__((address_space(3))) void *)(lds + (size_t)p * 64), 16, 0, 0);

    // A tile's rows are contiguous in X (frames of one utterance, row-major): they come in as coalesced 256-byte pieces by
    // LDS-DMA into this wave's slab -- 20 wave-instructions at D = 39 instead of 40 loads of one dword per lane 156 bytes
    // apart -- and the NEXT tile's are requested as soon as this tile's operands are built, a whole tile of matrix work ahead.
    const int slab_dwords = slab_bytes / 4;
    auto fetch_rows = [&](int tile_id) {
        const TileDesc t = a.tiles[tile_id];
        const float *src = a.X + t.start * a.dim;
        const int n = t.count * a.dim;                              // dwords of this tile
        for (int p0 = 0; p0 < slab_dwords; p0 += 64)                // (wave-uniform trip count)
            if (p0 + lane < n)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p0 + lane),
                                                 (__attribute__((address_space(3))) void *)(slab + p0), 4, 0, 0);
    };
    const int stride = gridDim.x * RES_WAVES;
    int tile_id = blockIdx.x * RES_WAVES + wave;
    if (tile_id < a.n_tiles) fetch_rows(tile_id);

    // centre and scale of the 8 KS contraction slots: wave-uniform, kept in scalar registers for the whole kernel
    float cen[8 * KS], scl[8 * KS];
#pragma unroll
    for (int d = 0; d < 8 * KS; d++) {
        const int dc = d < a.dim ? d : a.dim - 1;
        cen[d] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.center[dc])));
        scl[d] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.scale[dc])));
    }
    dma_publish_barrier();            // the image is in LDS for every wave (and this wave's first rows have landed)

    const float near_thr = lse_near_threshold(a.clamp);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (; tile_id < a.n_tiles; tile_id += stride) {
        const TileDesc tile = a.tiles[tile_id];
        const bool valid = col < tile.count;
        const int64_t row = tile.start + (valid ? col : 0);
        // ---- B fragments of this lane's frame: slot (ks, hh, j) = feature 8 ks + j, squared in the lower half-wave, itself
        //      in the upper one; the last upper slot carries the constant 1 (gmm_score_split.hip).  Two slots at a time:
        //      v_cvt_pk_f16_f32 rounds (to nearest even) and packs a pair of parts in one instruction.
        f16x8 breg[KS][P];
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's rows (requested a tile ago) are in its slab
            wave_sync();
            const float *mine = slab + (valid ? col : 0) * a.dim;
            float zmax = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                uint32_t w[P][4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    f32x2 v;
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int d = ks * 8 + j + e;
                        float xc = 0.0f;
                        if (d < a.dim) {                             // wave-uniform
                            xc = (mine[d] - cen[d]) * scl[d];
                            zmax = fmaxf(zmax, fabsf(xc));
                            xc = fminf(fmaxf(xc, -255.0f), 255.0f);  // x'^2 stays below fp16's 65504
                        }
                        float t = hh ? xc : xc * xc;
                        if (d == 8 * KS - 1) t = hh ? 1.0f : t;
                        v[e] = t;
                    }
                    const f16x2v h = __builtin_convertvector(v, f16x2v);                 // RNE, gradual underflow
                    const f16x2v l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2v);
                    w[0][j >> 1] = __builtin_bit_cast(uint32_t, h);
                    w[1][j >> 1] = __builtin_bit_cast(uint32_t, l);
                }
#pragma unroll
                for (int pi = 0; pi < P; pi++)
                    breg[ks][pi] = __builtin_bit_cast(f16x8, make_uint4(w[pi][0], w[pi][1], w[pi][2], w[pi][3]));
            }
            if (valid && zmax >= 255.0f) atomicOr(a.oor_flag, 1);   // saturated: the host re-scores on the fp32-grade engines
            wave_sync();                                             // every lane has read its row: the slab is free
            if (tile_id + stride < a.n_tiles) fetch_rows(tile_id + stride);
        }

        float m = NEG_BIG, ssum = 0.0f;
        for (int c = 0; c < a.n_chunks; c++) {
            const uint4 *at = lds + (size_t)c * TILE_U4 + lane;
            f32x16 acc;
            // A fragments two contraction steps ahead of the MFMAs that consume them; three part products per step in
            // the order of gmm_score_split.hip: (lo a, hi b), (hi a, lo b), (hi a, hi b)
            uint4 nx[2][P];
#pragma unroll
            for (int pi = 0; pi < P; pi++) nx[0][pi] = at[pi * 64];
            if (KS > 1) {
#pragma unroll
                for (int pi = 0; pi < P; pi++) nx[1][pi] = at[(P + pi) * 64];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                f16x8 av[P];
#pragma unroll
                for (int pi = 0; pi < P; pi++) av[pi] = __builtin_bit_cast(f16x8, nx[ks & 1][pi]);
                if (ks + 2 < KS) {
#pragma unroll
                    for (int pi = 0; pi < P; pi++) nx[ks & 1][pi] = at[((ks + 2) * P + pi) * 64];
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1], breg[ks][0], ks == 0 ? zero16 : acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0], breg[ks][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0], breg[ks][0], acc, 0, 0, 0);
            }
            lse_update16(acc, m, ssum, near_thr);
            const int s = a.chunks[c].model_done;                  // wave-uniform (scalar load)
            if (s >= 0) {
                const float ll = lse_close2(m, ssum, other_half(m), other_half(ssum), a.clamp);
                double mine = 0.0;
                bool hot = false;          // a frame in the band of the reference's partial-product flushes (lse.hpp)
                if (valid && hh == 0) {
                    mine = (double)ll;
                    if (a.frame_ll) a.frame_ll[(int64_t)s * a.n_frames + row] = ll;
                    hot = ll < a.band_hi;
                }
                mine = wave_sum_f64(mine);
                if (__builtin_amdgcn_ballot_w64(hot) != 0) mine = SR_FLUSH_POISON;
                if (lane == 0) a.partial[(int64_t)tile_id * a.n_models + s] = mine;
                m = NEG_BIG;
                ssum = 0.0f;
            }
        }
    }
}

// parameter image + one frame slab per wave
size_t resident_lds_bytes(int ks, int n_chunks, int dim) {
    return (size_t)n_chunks * ks * 2 * 64 * sizeof(uint4) + (size_t)RES_WAVES * res_slab_bytes(dim);
}

// a CU's LDS (gfx950: 160 KiB), all of which one workgroup may take
constexpr size_t RES_MAX_LDS = (size_t)160 << 10;

bool resident_fits(int ks, int n_chunks, int dim) {
    return ks >= 1 && ks <= 6 && n_chunks > 0 && resident_lds_bytes(ks, n_chunks, dim) <= RES_MAX_LDS;
}

template <int KS>
static void launch_res(const ResArgs &a, size_t lds_bytes) {
    static bool attr_set[MAX_DEVICES] = {};
    const int dev = ctx().device;
    if (!attr_set[dev]) {
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gmm_score_res_kernel<KS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_MAX_LDS));
        attr_set[dev] = true;
    }
    const int grid = std::max(1, std::min(ctx().n_cu, (a.n_tiles + RES_WAVES - 1) / RES_WAVES));
    hipLaunchKernelGGL((gmm_score_res_kernel<KS>), dim3((unsigned)grid), dim3(RES_WAVES * 64), lds_bytes, ctx().stream, a);
}

// `l`: the launch description of the split engine (tiles = the batch's 32-frame tile table, params = the f16x2 image)
void launch_score_resident(const MfmaLaunch &l, int KS, int n_chunks) {
    ResArgs a;
    a.X = l.X;
    a.tiles = l.tiles;
    a.params = reinterpret_cast<const uint4 *>(l.params);
    a.chunks = l.chunks;
    a.center = l.center;
    a.scale = l.scale;
    a.partial = l.partial;
    a.frame_ll = l.frame_ll;
    a.oor_flag = l.oor_flag;
    a.n_frames = l.n_frames;
    a.dim = l.dim;
    a.n_models = l.n_models;
    a.n_chunks = n_chunks;
    a.n_tiles = l.n_tiles;
    a.clamp = l.clamp;
    a.band_hi = l.band_hi;
    const size_t bytes = resident_lds_bytes(KS, n_chunks, l.dim);
    switch (KS) {
        case 1: launch_res<1>(a, bytes); break;
        case 2: launch_res<2>(a, bytes); break;
        case 3: launch_res<3>(a, bytes); break;
        case 4: launch_res<4>(a, bytes); break;
        case 5: launch_res<5>(a, bytes); break;
        case 6: launch_res<6>(a, bytes); break;
        default: fail("no resident scoring kernel for %d contraction steps", KS);
    }
}

}  // namespace sr
