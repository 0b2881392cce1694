// gmm_score_splitp.hip -- the generic split engines (gmm_score_split.hip: any set of independent models; math of
// gmm.cc:176-202, :237-244, :533-569) in the shape large batches take since round 4: ONE wide workgroup per CU, a 32-frame
// tile per wave, and the log-sum-exp of a chunk software-pipelined under the matrix instructions of the next one.
//
// Why (profiles/r04_splitp.txt).  The 4-wave kernel streams every 32-mixture chunk (KS * PARTS KiB) from L2 into LDS once per
// 128 frames: 16 GB of LDS-DMA per configs[1] pass (1 M frames x 100 models x 64 mixtures) -- at the ~6.4 TB/s the chip's
// LDS-DMA sustains that alone is 2.5 ms of the 2.76 ms the pass took, whatever the schedule inside the wave does.  One copy
// of the stream per CU shared by 12 or 16 waves is 5.3 / 4 GB.  Round 3 had tried that shape and lost (3.40 against 2.83 ms):
// with 3-4 waves per SIMD instead of 5 nothing hid a chunk's ~60-instruction online log-sum-exp any more, because inside ONE
// wave the 15 MFMAs and the epilogue ran one after the other.  Here they do not: while chunk c's MFMAs run into one
// accumulator, the wave's vector ALU works off chunk c-1's 16 values per lane from the other one -- 8 v_max3, 16 x (sub, exp,
// add), the rescale -- a few operations per MFMA, each slot fenced by sched_barrier so that the order survives the
// scheduler (an MFMA holds the matrix pipe for 32 cycles; the same wave's independent vector instructions issue in its
// shadow: scripts/ubench/mfma_lse_pinned.hip).  Builtins, not asm: the compiler then places the MFMA -> VALU and the
// transcendental hazards itself and counts lgkmcnt exactly; this file is built without SLP vectorisation (v_pk_add_f32 does
// not hide beside MFMAs).
//
// Stream: chunks of one group of models, G per stage, a ring of three stages in three separately named LDS arrays (a
// compiler-visible ds_read that may alias an LDS-DMA target makes hipcc drain vmcnt in front of it).  The barrier in front
// of stage s publishes stage s + 1 (every wave has waited for its own pieces, issued a stage earlier) and frees the slot of
// stage s - 1 for the LDS-DMA of stage s + 2; fragments are read one contraction step ahead, across chunk and stage borders.
//
// Per-model close: the two half-waves' states merge (lse.hpp, the reference's underflow semantics), the frame's value goes to
// a per-wave slab in LDS, and every 16 models the wave adds the slab up -- 4 lanes per model, 8 frames each, float64, fixed
// order -- and writes 16 partials: ~30 instructions per model instead of the ~70 of a DPP wave reduction per model.
// A (32-frame tile, model) with a frame in the band of the reference's partial-product flushes leaves +inf (lse.hpp).
#include "lse.hpp"
#include "score.hpp"
#include "split_prologue.hpp"
#include "split_schemes.hpp"
#include "wave_ops.hpp"

#include <algorithm>
#include <type_traits>

namespace sr {

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Workgroup shapes: 16 or 12 waves = one workgroup per CU (one copy of the stream for all of them); 8 waves = two per CU: twice the
// stream, but one workgroup's frame prologue (~500 vector instructions per wave, no MFMA) runs under the other's chains -- what short
// streams want (one 256-mixture model: 8 chunks per prologue).
// chunks per LDS stage: an even count (the two accumulators alternate statically), three stages within the workgroup's share of LDS
__host__ __device__ constexpr int splitp_stage_chunks(int ks, int parts, int waves) { return waves > 8 && ks * parts <= 10 ? 4 : 2; }
__host__ __device__ constexpr int splitp_lds_bytes(int ks, int parts, int waves) {
    return 3 * splitp_stage_chunks(ks, parts, waves) * ks * parts * 1024 + waves * 16 * 33 * 4;
}
__host__ __device__ constexpr bool splitp_fits(int ks, int parts, int waves) {
    // 4 waves per SIMD (16 waves, or two workgroups of 8) leave 128 registers: resident frame fragments of up to 10 x 4 beside the
    // two accumulators (ks * parts = 12: 12 bytes of scratch, 14: 64, 16: 116 -- build/gmm_score_splitp.resources); 12 waves have 168
    if (waves != 12 && ks * parts > 10) return false;
    return splitp_lds_bytes(ks, parts, waves) <= (160 * 1024 - 512) / (waves > 8 ? 1 : 2);
}

// (The parts-off measurement builds of round 4 -- the kernel with its log-sum-exp update, model close, LDS-DMA, fragment reads or
// MFMAs compiled out, profiles/r04_splitp.txt -- were macro hooks in this file until round 6; `git show 951632a:` has them.)
constexpr int SLAB_M = 16;          // models per slab flush
constexpr int SLAB_STRIDE = 33;     // floats per slab row (32 frames + 1: the 4-lanes-per-model read is conflict-free)
constexpr int EPI_OPS = 59;         // operations of one chunk's log-sum-exp update (see epi_op)

// operations of the epilogue issued in slots < u of an NM-slot chain: none in slots 0 and 1 (the accumulator they read was
// written by the MFMA right in front of this chain)
__host__ __device__ constexpr int epi_before(int nm, int u) {
    return nm <= 2 ? (u >= nm ? EPI_OPS : 0) : u <= 2 ? 0 : u >= nm ? EPI_OPS : (EPI_OPS * (u - 2) + (nm - 3)) / (nm - 2);
}

// scalars of a launch (the pointers are separate __restrict__ kernel parameters: inside a by-value struct they lose the
// qualifier, and the uniform loads of centre / scale turned into vector-memory loads with a full wait each)
struct SplitpArgs {
    int64_t n_frames;
    int dim, n_models, clamp, n_groups, n_tiles;
    int chunks_per_model;       // 32-mixture chunks of every model of the set
    float band_hi;
};

// state of the online log2-sum-exp of the model in progress + the temporaries of the update in flight
struct EpiState {
    float m, ssum;              // running maximum / sum relative to it (lse.hpp)
    float m_in, ssum_in;        // their values before the update in flight (its slow path starts over from them)
    float t[5], u0, u1, mn, r, x, ex[2], e;
};

// Operation OP of the update (m, ssum) <- (m, ssum) (+) the 16 values p: exactly lse_update16's fast path, value by value
// and in its order of additions, cut into single instructions.
template <int OP>
__device__ __forceinline__ void epi_op(EpiState &s, const f32x16 &p) {
    if constexpr (OP < 5) {
        s.t[OP] = fmaxf(fmaxf(p[3 * OP], p[3 * OP + 1]), p[3 * OP + 2]);
    } else if constexpr (OP == 5) {
        s.u0 = fmaxf(fmaxf(s.t[0], s.t[1]), s.t[2]);
    } else if constexpr (OP == 6) {
        s.u1 = fmaxf(fmaxf(s.t[3], s.t[4]), p[15]);
    } else if constexpr (OP == 7) {
        s.m_in = s.m;
        s.ssum_in = s.ssum;
        s.mn = fmaxf(fmaxf(s.u0, s.u1), s.m);
    } else if constexpr (OP == 8) {
        s.x = s.m - s.mn;
    } else if constexpr (OP == 9) {
        s.r = __builtin_amdgcn_exp2f(s.x);
    } else if constexpr (OP == 10) {
        s.x = p[0] - s.mn;
    } else if constexpr (OP == 11) {
        s.ex[0] = __builtin_amdgcn_exp2f(s.x);
    } else if constexpr (OP < 57) {
        constexpr int i = (OP - 12) / 3 + 1, k = (OP - 12) % 3;          // value i = 1..15: sub, exp, then the add of value i - 1
        if constexpr (k == 0) s.x = p[i] - s.mn;
        else if constexpr (k == 1) s.ex[i & 1] = __builtin_amdgcn_exp2f(s.x);
        else if constexpr (i == 1) s.e = 0.0f + s.ex[0];
        else s.e += s.ex[(i - 1) & 1];
    } else if constexpr (OP == 57) {
        s.e += s.ex[1];
    } else {
        s.ssum = fmaf(s.ssum, s.r, s.e);
        s.m = s.mn;
    }
}

template <typename SC, int KS, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES > 8 ? WAVES / 4 : 4)
void gmm_score_splitp_kernel(const float *__restrict__ X, const TileDesc *__restrict__ tiles, const uint4 *__restrict__ params,
                             const ChunkDesc *__restrict__ chunks, const int *__restrict__ group_chunk_begin,
                             const float *__restrict__ center, const float *__restrict__ scale, double *__restrict__ partial,
                             float *__restrict__ frame_ll, int *__restrict__ oor_flag, const SplitpArgs a) {
    constexpr int P = SC::PARTS, NPROD = SC::NPROD, NM = KS * NPROD;
    constexpr int TILE_U4 = KS * P * 64;                 // 16-byte fragments-per-lane of one 32-mixture chunk
    constexpr int G = splitp_stage_chunks(KS, P, WAVES);
    constexpr int STAGE_U4 = G * TILE_U4;
    constexpr int N_PIECES = STAGE_U4 / 64;              // 1 KiB wave-instructions per stage
    constexpr int PIECES_PER_CHUNK = TILE_U4 / 64;
    static_assert(G % 2 == 0, "the accumulators alternate statically within a stage");
    typedef typename SC::frag frag;
    __shared__ uint4 ring0[STAGE_U4];
    __shared__ uint4 ring1[STAGE_U4];
    __shared__ uint4 ring2[STAGE_U4];
    __shared__ float slab_all[WAVES * SLAB_M * SLAB_STRIDE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;
    const int hh = lane >> 5;
    const int wg_lo = blockIdx.x & 7;                    // XCD-aware order, as gmm_score_kernel
    const int q = blockIdx.x >> 3;
    const int g = q % a.n_groups;
    const int tile0 = ((q / a.n_groups) * 8 + wg_lo) * WAVES;
    if (tile0 >= a.n_tiles) return;
    const int chunk_begin = group_chunk_begin[g];
    const int chunk_end = group_chunk_begin[g + 1];
    const int n_chunks = chunk_end - chunk_begin;
    const int n_stages = (n_chunks + G - 1) / G;
    float *const slab = slab_all + wave * (SLAB_M * SLAB_STRIDE);

    auto ring = [&](auto RB) -> uint4 * {
        constexpr int rb = decltype(RB)::value % 3;
        if constexpr (rb == 0) return ring0;
        else if constexpr (rb == 1) return ring1;
        else return ring2;
    };
    // stage s of this group's stream -> ring slot RB; wave w issues pieces w, w + WAVES, ... (pieces of chunks past the
    // group's end are not fetched: the parameter buffer ends there)
    auto stage_load = [&](auto RB, int s) {
        const int c0 = chunk_begin + s * G;
        const uint4 *src = params + (size_t)c0 * TILE_U4;
        uint4 *dst = ring(RB);
#pragma unroll
        for (int i = 0; i < (N_PIECES + WAVES - 1) / WAVES; i++) {
            const int piece = i * WAVES + wave;
            if (piece < N_PIECES && c0 + piece / PIECES_PER_CHUNK < chunk_end)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 64), 16, 0, 0);
        }
    };
    stage_load(std::integral_constant<int, 0>{}, 0);
    if (n_stages > 1) stage_load(std::integral_constant<int, 1>{}, 1);

    // ---- resident B fragments of this lane's frame (as gmm_score_split_kernel: slot (ks, hh, j) is feature d = 8 ks + j,
    //      its square in the lower half-wave, the value itself in the upper one; the last upper slot carries the 1) ----
    const int tile_id = tile0 + wave;
    const bool has = tile_id < a.n_tiles;
    const TileDesc tile = tiles[has ? tile_id : a.n_tiles - 1];
    const bool valid = has && col < tile.count;
    const int64_t tile_start = tile.start;             // (wave-uniform; a lane's row = tile_start + col where it is needed)
    frag breg[KS][P];
    {
        float zmax = 0.0f;
        split_frame_fragments<SC, KS>(X + (tile_start + (valid ? col : 0)) * a.dim, a.dim, hh, center, scale, breg, zmax);
        if constexpr (SC::SCALED) {
            if (zmax >= 255.0f) atomicOr(oor_flag, 1);       // saturated: the host re-scores on the fp32-grade engines
        }
    }

    const float near_thr = lse_near_threshold(a.clamp);
    EpiState st;
    st.m = NEG_BIG;
    st.ssum = 0.0f;
    st.m_in = NEG_BIG;
    st.ssum_in = 0.0f;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        acc[0][r] = 0.0f;
        acc[1][r] = -__builtin_inff();      // "the chunk before the first": its update leaves (m, ssum) as they are
    }
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // ---- closing a model: merge the half-waves, park the frame's value in the slab; every SLAB_M models add the slab up ----
    int model_next = (int)(((int64_t)g * a.n_models) / a.n_groups);      // (the host's model-group boundaries, score_device)
    int slab_first = model_next, slab_n = 0;
    auto slab_flush = [&]() {
        wave_sync();
        const int j = lane >> 2, qq = lane & 3;            // 4 lanes per model, 8 frames each
        // The additions are those of wave_sum_f64 over a wave whose lanes 0..31 hold the frames (wave_ops.hpp: pairs, pairs of
        // pairs, ... within each row of 16, then the rows in order), so that this kernel and gmm_score_split_kernel leave the
        // same double for a (32-frame tile, model): which of them a batch's size selects does not show in an utterance's sum.
        float v[8];
        bool hot = false;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            v[i] = slab[j * SLAB_STRIDE + qq * 8 + i];
            hot |= v[i] < a.band_hi;
        }
        double sum = (((double)v[1] + (double)v[0]) + ((double)v[3] + (double)v[2])) + (((double)v[5] + (double)v[4]) + ((double)v[7] + (double)v[6]));
        // the four lanes of a model (DPP quad broadcasts): rows of 16 frames, then the two rows
        union { double d; int i[2]; } s, b;
        s.d = sum;
        const unsigned long long hot_mask = __builtin_amdgcn_ballot_w64(hot);
        double x[4];
#define SR_QUAD_BCAST(K)                                                              \
        b.i[0] = __builtin_amdgcn_update_dpp(0, s.i[0], (K) * 0x55, 0xf, 0xf, true);       \
        b.i[1] = __builtin_amdgcn_update_dpp(0, s.i[1], (K) * 0x55, 0xf, 0xf, true);       \
        x[K] = b.d;
        SR_QUAD_BCAST(0) SR_QUAD_BCAST(1) SR_QUAD_BCAST(2) SR_QUAD_BCAST(3)
#undef SR_QUAD_BCAST
        double tot = ((0.0 + (x[1] + x[0])) + (x[3] + x[2]));
        if ((hot_mask >> (lane & ~3)) & 0xfull) tot = SR_FLUSH_POISON;      // a frame of this model's tile in the band (lse.hpp)
        if (has && qq == 0 && j < slab_n) partial[(int64_t)tile_id * a.n_models + slab_first + j] = tot;
        wave_sync();
        slab_first += slab_n;
        slab_n = 0;
    };
    auto close_model = [&]() {
        const float ll = lse_close2(st.m, st.ssum, other_half(st.m), other_half(st.ssum), a.clamp);
        if (hh == 0) {
            if (valid && frame_ll) frame_ll[(int64_t)model_next * a.n_frames + tile_start + col] = ll;
            slab[slab_n * SLAB_STRIDE + col] = valid ? ll : 0.0f;
        }
        st.m = NEG_BIG;
        st.ssum = 0.0f;
        model_next++;
        if (++slab_n == SLAB_M) slab_flush();
    };
    // what follows a chunk's update: the reference's sub-DBL_MIN terms (lse.hpp; never taken on real data), the model's close
    auto after_update = [&](const f32x16 &p, bool closes) {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(st.mn < near_thr) != 0, 0)) {
            st.m = st.m_in;
            st.ssum = st.ssum_in;
            lse_update16(p, st.m, st.ssum, near_thr);
        }
        if (closes) close_model();
    };

    // A fragments, one contraction step ahead: F[(chunk parity * KS + ks) & 1][part]
    uint4 F[2][P];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_stages > 2) stage_load(std::integral_constant<int, 2>{}, 2);
#pragma unroll
    for (int pi = 0; pi < P; pi++) F[0][pi] = ring0[pi * 64 + lane];

    // Does chunk i of the group close its model?  Every model of the set has the same number of chunks (the launcher checks: the
    // reference's speaker sets are trained at one order, gmmset.py:24-31), and a group starts at a model: a countdown.  (A per-chunk
    // flag fetched from the chunk table -- as the 4-wave kernel does -- was a load whose result the loop carries: hipcc waited
    // for it, and with it for the LDS-DMA just issued, at the end of the block that issued it.)
    int chunks_left = a.chunks_per_model;
    bool prev_closes = false;

    auto do_chunk = [&](auto RB, auto CI) {
        constexpr int ci = decltype(CI)::value;
        constexpr int par = ci & 1;
        f32x16 &cur = acc[par];
        const f32x16 &prev = acc[par ^ 1];
        const bool closes = --chunks_left == 0;
        if (closes) chunks_left = a.chunks_per_model;
        const uint4 *here = ring(RB) + ci * TILE_U4 + lane;
        // the chunk after this one: the next of this stage, or the first of the next stage (published by this stage's barrier)
        const uint4 *next = ci + 1 < G ? here + TILE_U4 : ring(std::integral_constant<int, decltype(RB)::value + 1>{}) + lane;
        static_for<0, NM>([&](auto U) {
            constexpr int u = decltype(U)::value;
            constexpr int ks = u / NPROD, pr = u % NPROD;
            constexpr int f = (ci * KS + ks) & 1;
            if constexpr (pr == 0) {
#pragma unroll
                for (int pi = 0; pi < P; pi++) F[f ^ 1][pi] = ks + 1 < KS ? here[((ks + 1) * P + pi) * 64] : next[pi * 64];
            }
            cur = SC::mfma(__builtin_bit_cast(frag, F[f][SC::AI[pr]]), breg[ks][SC::BI[pr]], u == 0 ? zero16 : cur);
            static_for<epi_before(NM, u), epi_before(NM, u + 1)>([&](auto OP) { epi_op<decltype(OP)::value>(st, prev); });
            // (the update's results are only USED behind the chain -- by a branch, at that: without a use here the optimiser
            // sinks the whole update out of the slots, whatever the scheduling fences say)
            if constexpr (epi_before(NM, u) != epi_before(NM, u + 1))
                asm volatile("" : "+v"(st.mn), "+v"(st.r), "+v"(st.x), "+v"(st.ex[0]), "+v"(st.ex[1]), "+v"(st.e), "+v"(st.ssum), "+v"(st.m));
            __builtin_amdgcn_sched_barrier(0);
        });
        after_update(prev, prev_closes);
        prev_closes = closes;
    };
    auto do_stage = [&](auto RB, int s) {
        if (s > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage s + 1 (issued a stage ago)
            __syncthreads();
            if (s + 2 < n_stages) stage_load(std::integral_constant<int, decltype(RB)::value + 2>{}, s + 2);
        }
        static_for<0, G>([&](auto CI) {
            const int i = s * G + decltype(CI)::value;
            if (i < n_chunks) do_chunk(RB, CI);
        });
    };
    for (int s = 0; s < n_stages; s += 3) {
        do_stage(std::integral_constant<int, 0>{}, s);
        if (s + 1 < n_stages) do_stage(std::integral_constant<int, 1>{}, s + 1);
        if (s + 2 < n_stages) do_stage(std::integral_constant<int, 2>{}, s + 2);
    }
    // the last chunk's update has no chain to ride on
    if (n_chunks > 0) {
        if ((n_chunks - 1) & 1) {
            static_for<0, EPI_OPS>([&](auto OP) { epi_op<decltype(OP)::value>(st, acc[1]); });
            after_update(acc[1], prev_closes);
        } else {
            static_for<0, EPI_OPS>([&](auto OP) { epi_op<decltype(OP)::value>(st, acc[0]); });
            after_update(acc[0], prev_closes);
        }
    }
    if (slab_n > 0) slab_flush();
}

template <typename SC, int KS, int WAVES>
void launch_splitp(const MfmaLaunch &l, int chunks_per_model) {
    SplitpArgs a;
    a.n_frames = l.n_frames;
    a.dim = l.dim;
    a.n_models = l.n_models;
    a.clamp = l.clamp;
    a.n_groups = l.n_groups;
    a.n_tiles = l.n_tiles;
    a.chunks_per_model = chunks_per_model;
    a.band_hi = l.band_hi;
    const int n_wg = (l.n_tiles + WAVES - 1) / WAVES;
    dim3 grid((unsigned)((int64_t)l.n_groups * ((n_wg + 7) / 8) * 8));
    hipLaunchKernelGGL((gmm_score_splitp_kernel<SC, KS, WAVES>), grid, dim3(WAVES * 64), 0, ctx().stream, l.X, l.tiles,
                       reinterpret_cast<const uint4 *>(l.params), l.chunks, l.group_chunk_begin, l.center, l.scale, l.partial,
                       l.frame_ll, l.oor_flag, a);
}

template <typename SC, int KS>
bool dispatch_splitp_waves(const MfmaLaunch &l, int waves, int chunks_per_model) {
    if constexpr (splitp_fits(KS, SC::PARTS, 16)) {
        if (waves == 16) {
            launch_splitp<SC, KS, 16>(l, chunks_per_model);
            return true;
        }
    }
    if constexpr (splitp_fits(KS, SC::PARTS, 12)) {
        if (waves == 12) {
            launch_splitp<SC, KS, 12>(l, chunks_per_model);
            return true;
        }
    }
    if constexpr (splitp_fits(KS, SC::PARTS, 8)) {
        if (waves == 8) {
            launch_splitp<SC, KS, 8>(l, chunks_per_model);
            return true;
        }
    }
    return false;
}

}  // namespace

// workgroups of that shape a CU holds
int splitp_resident_per_cu(int waves) { return waves > 8 ? 1 : 2; }

// 32-frame tiles a workgroup of the wide shape takes (= its waves)
int splitp_waves(int scheme, int ks, int want) {
    // (instantiated for the two-part fp16 scheme: what the dispatcher takes for every well-conditioned set; the bf16x3 fallback
    // keeps the 4-wave kernel -- 28 more variants of this file cost a minute and a half of build time)
    const int parts = 2;
    if (scheme != SPLIT_F16X2) return 0;
    if (ks < 2 || ks > 8) return 0;
    for (int w : {want, 16, 12, 8})
        if ((w == 16 || w == 12 || w == 8) && splitp_fits(ks, parts, w)) return w;
    return 0;
}

// `l.tiles` = the batch's 32-frame tiles.  false: no such variant (the caller takes gmm_score_split_kernel).
// `chunks_per_model`: the 32-mixture chunks of EVERY model of the set (sets of models of different orders take the 4-wave kernel).
bool launch_score_splitp(const MfmaLaunch &l, int scheme, int KS, int waves, int chunks_per_model) {
#define SR_SPLITP_CASE(K) \
    case K:               \
        return dispatch_splitp_waves<f16x2, K>(l, waves, chunks_per_model);
    if (scheme != SPLIT_F16X2 || chunks_per_model <= 0) return false;
    switch (KS) {
        SR_SPLITP_CASE(2) SR_SPLITP_CASE(3) SR_SPLITP_CASE(4) SR_SPLITP_CASE(5) SR_SPLITP_CASE(6) SR_SPLITP_CASE(7) SR_SPLITP_CASE(8)
        default: return false;
    }
#undef SR_SPLITP_CASE
}

}  // namespace sr
