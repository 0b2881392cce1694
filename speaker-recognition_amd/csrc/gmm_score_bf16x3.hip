// gmm_score_bf16x3.hip -- third engine for the scoring math of gmm.cc:176-202, :237-244, :533-569:
// the expanded quadratic form of gmm_score_mfma.hip
//   log2 density_k(x) = sum_d ( A2_kd x'_d^2 + A1_kd x'_d ) + C_k ,   x' = x - centre
// evaluated on the bf16 matrix cores at fp32 accuracy.  Every fp32 operand is split into three
// bf16 parts (hi + mid + lo = all 24 significand bits, round-to-nearest-even at each step, so the
// split is exact), and a product a*b is taken as the six part products whose weight is >= 2^-24:
//   a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)
// Each part product is exact in fp32 (8 x 8 significand bits) and v_mfma_f32_32x32x16_bf16
// accumulates in fp32, so the result carries the rounding of an fp32 accumulation plus the three
// dropped products (< 3 * 2^-24 relative per term) -- measured against the float64 oracle it is
// as close as the fp32 FMA chain of the other two engines (tests/test_gpu_gmm.py).
//
// Why: gfx950 has no tf32/xf32 and its fp32 MFMA runs at the vector rate (157 TFLOP/s); the bf16
// MFMA is 16x that, so six bf16 products cost 6/16 of one fp32 MFMA pass over the same tile.
//
// Mapping (same roles as the fp32 matrix-core kernel): MFMA rows = 32 mixtures, A fragments
// streamed through LDS by LDS-DMA (one 32-mixture tile = KS*3 fragments of 1 KiB per chunk,
// double-buffered); MFMA columns = 32 frames, B fragments (three bf16 parts of the frame's
// (x'^2, x', 1) vector) resident in VGPRs for the whole kernel.  A wave owns FT column tiles; the
// accumulator layout keeps a frame's 16 mixture rows in one lane, so the online log-sum-exp is
// lane-local and the two half-waves merge once per model.
#include "score.hpp"
#include "wave_ops.hpp"

#include <algorithm>

namespace sr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr float BX_LN2_F = 0.69314718055994530942f;
constexpr float BX_MINLOG_F = -708.396418532264f;
constexpr float BX_LN_1E_15_F = -34.538776394910684f;

__host__ __device__ constexpr int bx3_waves_per_eu(int ks, int ft) {
    const int regs = ft * (ks * 12 + 16) + 24 + 36;
    return regs <= 96 ? 5 : regs <= 128 ? 4 : regs <= 168 ? 3 : regs <= 256 ? 2 : 1;
}

__device__ __forceinline__ uint32_t bx_rne(float v) {   // fp32 bit pattern of v rounded to bf16 (RNE)
    const uint32_t u = __float_as_uint(v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

template <int KS, int FT>
__global__ __launch_bounds__(256, bx3_waves_per_eu(KS, FT))
void gmm_score_bf16x3_kernel(const float *__restrict__ X, const TileDesc *__restrict__ tiles,
                             const uint4 *__restrict__ params, const ChunkDesc *__restrict__ chunks,
                             const int *__restrict__ group_chunk_begin,
                             const float *__restrict__ center, double *__restrict__ partial,
                             float *__restrict__ frame_ll, int64_t n_frames, int dim, int n_models,
                             int clamp, int n_groups, int n_tiles) {
    constexpr int TILE_U4 = KS * 3 * 64;       // 16-byte fragments-per-lane of one 32-mixture tile
    constexpr int PF = (TILE_U4 + 255) / 256;
    __shared__ uint4 lds_a[TILE_U4];
    __shared__ uint4 lds_b[TILE_U4];
    __shared__ double close_slot[2][4];        // the four waves' sums of a closed model, two generations

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;                 // frame column inside a 32-frame tile
    const int hh = lane >> 5;                  // which 8 of the 16 contraction indices of a step
    const int tile_lo = blockIdx.x & 7;        // XCD-aware order, as gmm_score_kernel
    const int q = blockIdx.x >> 3;
    const int g = q % n_groups;
    const int tile_id = (q / n_groups) * 8 + tile_lo;
    if (tile_id >= n_tiles) return;
    const TileDesc tile = tiles[tile_id];
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

    // ---- resident B fragments.  Contraction slot (ks, hh, j) is feature d = 8 ks + j: its square in
    //      the lower half-wave (hh = 0), the value itself in the upper one (hh = 1); the very last
    //      upper slot carries the constant 1 that picks up C_k (8 KS > dim, so it is free).
    //      breg[ft][ks][part] = the three bf16 parts of this lane's 8 slots of step ks. ----
    bf16x8 breg[FT][KS][3];
    bool valid[FT];
    int64_t row[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ft++) {
        const int local = (wave * FT + ft) * 32 + col;
        valid[ft] = local < tile.count;
        row[ft] = tile.start + (valid[ft] ? local : 0);
        const float *src = X + row[ft] * dim;
        float xs[8 * KS];
#pragma unroll
        for (int d = 0; d < 8 * KS; d++) xs[d] = src[d < dim ? d : dim - 1];
        // keep the loads unconditional and batched: without this the compiler sinks each one into
        // its own `d < dim` branch with a full wait
#pragma unroll
        for (int d = 0; d < 8 * KS; d++) asm volatile("" : "+v"(xs[d]));
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            uint32_t w[3][4];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int d = ks * 8 + j;
                const float xc = xs[d] - center[d < dim ? d : dim - 1];
                float v = hh ? xc : xc * xc;
                v = d < dim ? v : 0.0f;
                if (d == 8 * KS - 1) v = hh ? 1.0f : v;
                const uint32_t p0 = bx_rne(v);
                const float r1 = v - __uint_as_float(p0);
                const uint32_t p1 = bx_rne(r1);
                const float r2 = r1 - __uint_as_float(p1);
                const uint32_t p2 = bx_rne(r2);
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
            for (int p = 0; p < 3; p++) {
                const uint4 u = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
                breg[ft][ks][p] = __builtin_bit_cast(bf16x8, u);
            }
        }
    }

    float m[FT], ssum[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ft++) {
        m[ft] = NEG_BIG;
        ssum[ft] = 0.0f;
    }
    __syncthreads();

    // A model's four wave sums meet in LDS and leave as ONE double per (tile, model): the store
    // happens after the chunk's closing barrier, at the top of the next chunk (or after the loop).
    int pending_model = -1, pending_gen = 0, gen = 0;
    auto flush_pending = [&]() {
        if (pending_model >= 0 && tid == 0) {
            const double *p = close_slot[pending_gen];
            partial[(int64_t)tile_id * n_models + pending_model] = ((p[0] + p[1]) + p[2]) + p[3];
        }
        pending_model = -1;
    };

    auto do_chunk = [&](const uint4 *cur, uint4 *other, int c) {
        flush_pending();
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
        uint4 n0 = at[0], n1 = at[64], n2 = at[128];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, n0);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, n1);
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, n2);
            if (ks + 1 < KS) {
                n0 = at[((ks + 1) * 3 + 0) * 64];
                n1 = at[((ks + 1) * 3 + 1) * 64];
                n2 = at[((ks + 1) * 3 + 2) * 64];
            }
            // small products first, and consecutive MFMAs share one operand
            const bf16x8 aa[6] = {a2, a1, a1, a0, a0, a0};
            const int bi[6] = {0, 0, 1, 1, 2, 0};
#pragma unroll
            for (int pr = 0; pr < 6; pr++)
#pragma unroll
                for (int ft = 0; ft < FT; ft++)
                    acc[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[pr], breg[ft][ks][bi[pr]],
                                                                      (ks == 0 && pr == 0) ? zero16 : acc[ft], 0, 0, 0);
        }
        // online log2-sum-exp over this lane's 16 mixture rows of each frame column
#pragma unroll
        for (int ft = 0; ft < FT; ft++) {
            float mx = acc[ft][0];
#pragma unroll
            for (int r = 1; r < 16; r++) mx = fmaxf(mx, acc[ft][r]);
            const float mn = fmaxf(m[ft], mx);
            float e = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) e += __builtin_amdgcn_exp2f(acc[ft][r] - mn);
            ssum[ft] = fmaf(ssum[ft], __builtin_amdgcn_exp2f(m[ft] - mn), e);
            m[ft] = mn;
        }

        if (model_done >= 0) {
            const int s = model_done;
            double mine = 0.0;
#pragma unroll
            for (int ft = 0; ft < FT; ft++) {
                // merge the two half-waves (the other 16 mixture rows of the same frame)
                const float om = other_half(m[ft]);
                const float os = other_half(ssum[ft]);
                const float mn = fmaxf(m[ft], om);
                const float tot = ssum[ft] * __builtin_amdgcn_exp2f(m[ft] - mn) +
                                  os * __builtin_amdgcn_exp2f(om - mn);
                float ll = BX_LN2_F * (mn + log2f(tot));
                if (clamp && ll < BX_MINLOG_F) ll = BX_LN_1E_15_F;
                if (valid[ft] && hh == 0) {
                    mine += (double)ll;
                    if (frame_ll) frame_ll[(int64_t)s * n_frames + row[ft]] = ll;
                }
                m[ft] = NEG_BIG;
                ssum[ft] = 0.0f;
            }
            mine = wave_sum_f64(mine);
            if (lane == 0) close_slot[gen][wave] = mine;
            pending_model = s;
            pending_gen = gen;
            gen ^= 1;
        }
        __syncthreads();
    };

    for (int c = chunk_begin; c < chunk_end; c += 2) {
        do_chunk(lds_a, lds_b, c);
        if (c + 1 < chunk_end) do_chunk(lds_b, lds_a, c + 1);
    }
    flush_pending();
}

template <int KS, int FT>
static void launch_bx3(const MfmaLaunch &a) {
    dim3 grid((unsigned)((int64_t)a.n_groups * ((a.n_tiles + 7) / 8) * 8));
    hipLaunchKernelGGL((gmm_score_bf16x3_kernel<KS, FT>), grid, dim3(256), 0, ctx().stream, a.X, a.tiles,
                       reinterpret_cast<const uint4 *>(a.params), a.chunks, a.group_chunk_begin, a.center,
                       a.partial, a.frame_ll, a.n_frames, a.dim, a.n_models, a.clamp, a.n_groups, a.n_tiles);
}

template <int KS>
static void dispatch_bx3_ft(const MfmaLaunch &a, int FT) {
    if (FT == 1) return launch_bx3<KS, 1>(a);
    if constexpr (KS <= 6) {
        if (FT == 2) return launch_bx3<KS, 2>(a);
    }
    fail("bf16x3 engine: %d column tiles per wave not instantiated for %d contraction steps", FT, KS);
}

int bx3_max_ft(int ks) { return ks <= 6 ? 2 : 1; }

void launch_score_bf16x3(const MfmaLaunch &a, int KS, int FT) {
    switch (KS) {
        case 1: dispatch_bx3_ft<1>(a, FT); break;
        case 2: dispatch_bx3_ft<2>(a, FT); break;
        case 3: dispatch_bx3_ft<3>(a, FT); break;
        case 4: dispatch_bx3_ft<4>(a, FT); break;
        case 5: dispatch_bx3_ft<5>(a, FT); break;
        case 6: dispatch_bx3_ft<6>(a, FT); break;
        case 7: dispatch_bx3_ft<7>(a, FT); break;
        case 8: dispatch_bx3_ft<8>(a, FT); break;
        case 9: dispatch_bx3_ft<9>(a, FT); break;
        default: fail("no bf16x3 scoring kernel for %d contraction steps", KS);
    }
}

}  // namespace sr
