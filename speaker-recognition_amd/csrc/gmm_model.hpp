// gmm_model.hpp -- host-side model objects behind the C ABI handles, the reference's text
// model format, and the packing of mixture parameters into the layout the scoring kernel
// streams through LDS.
#pragma once

#include "common.hpp"

#include <memory>
#include <string>
#include <vector>

struct SRModelSet;

// The handle type of the C ABI (`GMM *`).  Reference: class GMM, src/gmm/src/gmm.hh:130-173;
// per-Gaussian mean/sigma (sigma = STANDARD DEVIATIONS) as gmm.hh:24-46.
namespace sr {
uint64_t next_gmm_uid();       // gmm_model.cpp
}

struct GMM {
    int nr_mixtures = 0;
    int covariance_type = 1;  // COVTYPE_DIAGONAL, gmm.hh:18-22
    int dim = 0;
    std::vector<double> weights;  // [K]
    std::vector<double> mean;     // [K*D]
    std::vector<double> sigma;    // [K*D]
    // lazily packed one-model sets, one per device (threads on different GPUs may score the same handle concurrently:
    // each holds its device's lock only); invalidated by training.  (mutable: a cache -- a UBM handed to the MAP trainer as
    // `const` lends its packed set to every speaker's first E-step, em.hip)
    mutable std::shared_ptr<SRModelSet> single[sr::MAX_DEVICES];
    // identity of the PARAMETERS for caches that hold several models (sr_score_models_f32): `uid` is unique per object for the life
    // of the process (an address can come back), `generation` moves whenever the parameters change (every writer calls drop_single)
    uint64_t uid = sr::next_gmm_uid();
    uint64_t generation = 0;
    void drop_single() {
        for (auto &s : single) s.reset();
        generation++;
    }
    bool trained() const { return dim > 0 && (int)weights.size() == nr_mixtures; }
};

namespace sr {

// ---- text format: GMM::load gmm.cc:664-682 / Gaussian::load :125-150; GMM::dump :655-662 /
// Gaussian::dump :101-123 (default ostream precision = "%g", 6 significant digits) ----
void gmm_parse_text(const std::string &text, GMM &out);
std::string gmm_format_text(const GMM &g);

// ---- packed parameters ----
// A record holds KB=4 mixtures for all (padded) dims: for d in [0,DP): float4 pair
//   {s0,m0,s1,m1}, {s2,m2,s3,m3}      with s = sqrt(log2(e)/2)/sigma, m = -(mean - center)*s
// followed by one float4 {c0,c1,c2,c3}, c = log2(e) * (ln w - sum_d ln(sqrt(2 pi) sigma_d)),
// so that  log2-density_k(x) = c_k - sum_d (x'_d*s_kd + m_kd)^2,  x' = x - center   (2 FMAs per (d,k)).
// Padded mixtures carry c = -1e30 (contribute 2^-inf = 0); padded dims carry s = m = 0.
// Record size = (2*DP+1) float4.  A chunk = up to CB consecutive records of ONE model and is
// what a workgroup copies into LDS at a time.
constexpr int KB = 4;
constexpr int CB = 8;
constexpr float NEG_BIG = -1.0e30f;

struct ChunkDesc {
    uint32_t offset_f4;  // float4 index of the chunk's first record in the params buffer
    int32_t n_records;   // 1..CB
    int32_t model_done;  // model index when this chunk is the last of its model, else -1
    int32_t pad;
};

constexpr int MAX_REG_DIM = 96;     // largest feature dimension a lane keeps whole in registers (gmm_score_kernel<DP,...>); beyond it the
                                    // D-sliced kernels are the faster ones (scripts/debug/wide_ab.py, round 6: D = 100 3.20 -> 1.97 ms,
                                    // D = 128 2.63 -> 1.84 ms on 200 k frames x 16 x 64; D = 96 1.56 against 1.95: stays)
constexpr int WIDE_DC = 64;         // wider rows go through the kernels slice by slice of this many dimensions
constexpr int MAX_DIM = 1 << 16;    // sanity bound only (a corrupt model file); the reference has none (gmm.cc:40-51)
constexpr int MAX_MATRIX_DIM = 64;  // largest one the matrix-core engines are packed for
int pick_padded_dim(int dim);  // <= MAX_REG_DIM: smallest instantiated kernel dim >= dim; above: whole slices of WIDE_DC; throws if > MAX_DIM

struct PackedModels {
    int n_models = 0;
    int dim = 0;  // actual feature dim
    int dp = 0;   // padded dim the kernels are instantiated for
    std::vector<float> params;      // float4-granular
    std::vector<float> center;      // [dp] the set's centre (mean of all mixture means): the kernels work on x - center,
                                    // so that a feature space far from the origin costs no digits in x*s + m
    std::vector<ChunkDesc> chunks;  // all models, in model order
    std::vector<int> model_chunk_begin;  // [S+1]
    // Width (nats) of the band above ln DBL_MIN in which the reference's flushes of PARTIAL products can change a
    // frame's log-likelihood (lse.hpp): max_k sum_d max(0, -ln sigma_kd) + ln K + 17.5 over the set's models.
    double flush_band = 0.0;
};
PackedModels pack_models(const std::vector<const GMM *> &models);

// ---- the expanded quadratic form the matrix-core layouts below share ----
//   log2 density_k(x) = sum_d ( A2_kd x'_d^2 + A1_kd x'_d ) + C_k,   x' = x - center
//   A2 = -log2e/(2 sigma^2),  A1 = log2e mu'/sigma^2,  C = log2e (ln w - sum ln(sqrt(2pi) sigma) - sum mu'^2/(2 sigma^2))
// `amp` = max_k sum_d (mu'_d / sigma_d)^2 measures the cancellation the expanded form has to survive; the dispatcher keeps
// the 2-FMA vector kernel (first layout) when it is large.  (The fp32 image of this form, for v_mfma_f32_32x32x2_f32, was the
// round-1 engine 2: never selected once the split engines existed, removed in round 5.)
constexpr int MT = 32;          // mixtures per MFMA tile

// ---- third layout: the same expanded form with every coefficient split into three bf16 parts
// (hi + mid + lo carries all 24 significand bits of the fp32 value), for the bf16 matrix cores:
//   a*b ~= a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)      (the three dropped products are < 2^-24 |a b|)
// six v_mfma_f32_32x32x16_bf16 products with fp32 accumulation, 16x the fp32 MFMA rate each.
// Contraction: KS = ceil((D+1)/8) steps of 16 slots; slot (ks, hh, j) belongs to feature d = 8 ks + j
// and holds A2_kd (hh = 0, against x'_d^2) or A1_kd (hh = 1, against x'_d); the last upper slot
// (d = 8 KS - 1 >= D) holds C_k against the constant 1.  A mixture tile = 32 mixtures; its image is
// [ks][part][lane][8 bf16]: lane l supplies mixture l & 31, hh = l >> 5, j = 0..7 (one 16-byte LDS
// read per lane and part).
// The same layout serves the two-part fp16 scheme (gmm_score_split.hip: [ks][part][lane][8 x fp16],
// two parts, three part products).  fp16 has a 5-bit exponent, so that scheme works on
// x' = (x - center) * scale with a per-dimension power-of-two `scale` ~ 1 / (geometric mean of the
// dimension's sigmas) folded into the coefficients (exact), pads dead mixtures with C = -60000
// instead of -1e30, and is only offered when every dimension's sigma range is moderate (`sigma_ratio`).
constexpr int SPLIT_BF16X3 = 0, SPLIT_F16X2 = 1;
constexpr int SPLIT_F16X1 = 2;      // launch-time only: the fp16 image, HIGH parts only (one part product: a few nats of error -- an offset, not a result)
constexpr float F16_NEG_BIG = -60000.0f;
struct PackedSplit {
    int scheme = SPLIT_BF16X3, parts = 3;
    int ks = 0;
    std::vector<uint16_t> params;    // 16-byte granular (8 x 16 bit)
    std::vector<ChunkDesc> chunks;   // one mixture tile per chunk; offset in 16-byte units
    std::vector<int> model_chunk_begin;
    std::vector<float> center;       // [8 ks]: dim entries, zero-padded (the kernels index slots, split_prologue.hpp)
    std::vector<float> scale;        // [8 ks]: dim powers of two (all 1 for bf16x3), zero-padded
    double amp = 0.0;
    double pad_waste = 0.0;
    double sigma_ratio = 1.0;        // max over dims of (max sigma / min sigma) across all mixtures
    double coef_max = 0.0;           // largest |coefficient| after scaling (fp16 range check)
};
typedef PackedSplit PackedBf16x3;
PackedSplit pack_models_split(const std::vector<const GMM *> &models, int scheme);

// ---- fourth layout: speaker sets that share sigma and weights (MAP adaptation moves the means only,
// gmmubm.cc:40-81 -- a UBM and every speaker adapted from it).  The quadratic half of the
// contraction, Q_k(x) = sum_d A2_kd x'_d^2, is then the same for every model at a given mixture:
// it is evaluated once per (mixture tile, block of SHARED_SB models) and fed as the C operand of
// each model's linear half L_sk(x) = sum_d A1_skd x'_d + C_sk.  Split-bf16 arithmetic as above.
//   quadratic tile: KQ = ceil(D/16) steps, slot c = 16 ks + 8 hh + j <-> feature c (A2, against x'^2)
//   linear tile:    KL = ceil((D+1)/16) steps, slot c < D <-> A1 (against x'), slot 16 KL - 1 <-> C_k (against 1)
// Stream order (what a workgroup walks): per block of SHARED_SB models, per mixture tile t:
//   [Q_t][L_{s0,t}] ... [L_{s14,t}]  -- 1 + SHARED_SB images, an even count, so the two LDS buffers
// alternate statically.  The last block is padded with phantom models (C = -1e30, never stored).
constexpr int SHARED_SB = 15;
struct SharedBlock {
    uint32_t offset_u4;   // first 16-byte fragment of the block's stream
    int32_t first_model;
    int32_t n_models;     // live models in this block (<= SHARED_SB)
    int32_t pad;
};
struct PackedBx3Shared {
    int kq = 0, kl = 0, n_tiles = 0;   // mixture tiles per model
    std::vector<uint16_t> params;
    std::vector<SharedBlock> blocks;
    std::vector<float> center;
    double amp = 0.0, pad_waste = 0.0;
};
// ---- fifth layout: the shared-sigma form on two fp16 parts (gmm_score_h2_shared.hip).  The three
// part products of a half are laid end to end as one contraction:
//   quadratic: [lo(A2_d) x hi(z_d^2) | hi(A2_d) x lo(z_d^2) | hi(A2_d) x hi(z_d^2)]          3D slots
//   linear:    [lo(A1_d, C) x hi(z_d, 1) | hi(A1_d) x lo(z_d) | hi(A1_d, C) x hi(z_d, 1)]     3D+2 slots
// (z = (x - center) * scale, per-dimension power-of-two scale as PackedSplit), each padded to a
// multiple of 16 once.  `q_desc` / `l_desc` tell the kernel what the frame side of slot c is:
// d | op << 8 with op 0 = zero, 1 = high part, 2 = low part, 3 = the constant 1.
// An image is [ks][lane][8 x fp16] with lane l = mixture l & 31, slots 16 ks + 8 (l >> 5) + j, padded
// to max(KQF, KLF) KiB; stream order as PackedBx3Shared.  `ref` is the set's first model alone in
// the generic two-part layout: its per-frame log-likelihood is the kernel's log-sum-exp offset.
struct PackedH2Shared {
    int kqf = 0, klf = 0, n_tiles = 0;
    std::vector<uint16_t> params;
    std::vector<SharedBlock> blocks;
    std::vector<float> center, scale;
    std::vector<uint16_t> q_desc, l_desc;
    double amp = 0.0, pad_waste = 0.0, sigma_ratio = 1.0, coef_max = 0.0;
    PackedSplit ref;
};
PackedH2Shared pack_models_h2_shared(const std::vector<const GMM *> &models);
// Work items of the pipelined shared-sigma kernel over a table of tiles with `counts[t]` frames each (<= frames_per_tile): item =
// {tile, tile or -1, tile or -1, tile or -1}.  Every FULL tile is an item of its own, in order; with `pack_tails` the ragged ones are
// packed greedily, in order, up to four and up to frames_per_tile columns to an item, and all packed items come LAST (their waves
// close several tiles per model block: together at the end of the grid they stay in step with each other, scattered they push their
// workgroups out of phase with the others of their XCD and the parameter stream out of its L2).  Host-only: tests/host/host_checks.cpp.
struct WorkItem { int t[4]; };
std::vector<WorkItem> pack_tail_tiles(const std::vector<int> &counts, int frames_per_tile, bool pack_tails);
// Sweep starts of the MFCC kernels' mel gather (mfcc.hip: four lanes per band, one ds_read_b128 each per step, bands {0,3,5,6},
// {1,2,4,7}, {8,11,13,14}, {9,10,12,15} of a pass of 16 served together over 16 slots of 16 bytes): start[b] <= col0[b], a multiple
// of 4, >= 0, moved down only as far as the band's padded run still fits pass_len[b / 16]; per group the choice with the fewest extra
// LDS cycles, then the least padding.  Host-only.
void mel_sweep_starts(const int *col0, const int *cnt, int n_bands, const int pass_len[4], int *start);
int mel_sweep_extra_cycles(const int *start, const int *cnt, int n_bands, int pass);      // extra LDS cycles per read instruction of that pass
bool models_share_sigma_and_weights(const std::vector<const GMM *> &models);
PackedBx3Shared pack_models_bx3_shared(const std::vector<const GMM *> &models);
void split_bf16x3(float v, uint16_t out[3]);   // round-to-nearest-even hi/mid/lo parts
void split_f16x2(float v, uint16_t out[2]);    // round-to-nearest-even hi/lo fp16 parts (gradual underflow)
uint16_t f32_to_f16_rne(float v);
float f16_to_f32(uint16_t h);

}  // namespace sr

// Device-resident speaker set (C ABI handle `SRModelSet *`).
struct SRModelSet {
    sr::PackedModels host;
    sr::DevBuf<float> d_params, d_center0;     // vector layout and its centre
    sr::DevBuf<sr::ChunkDesc> d_chunks;
    sr::PackedSplit bx3;             // split-bf16 layout for the bf16 matrix-core engine
    sr::DevBuf<uint16_t> d_bx3_params;
    sr::DevBuf<float> d_bx3_center;
    sr::DevBuf<sr::ChunkDesc> d_bx3_chunks;
    std::vector<char> chunks_image;  // (em.hip) the bytes d_chunks holds: a re-packed model of the same shape has the same chunk table
    bool bx3_stale = false;          // a re-packed set (em.hip: a fit recycles its sets): ensure_bx3_layout fills the standing buffers again
    sr::PackedSplit h2;              // two-part fp16 layout (three part products)
    sr::DevBuf<uint16_t> d_h2_params;
    sr::DevBuf<float> d_h2_center, d_h2_scale;
    sr::DevBuf<sr::ChunkDesc> d_h2_chunks;
    sr::PackedBx3Shared shared;      // shared-sigma layout (empty unless the set qualifies)
    sr::DevBuf<uint16_t> d_shared_params;
    sr::DevBuf<sr::SharedBlock> d_shared_blocks;
    sr::DevBuf<float> d_shared_center;
    sr::PackedH2Shared h2s;          // shared-sigma layout on two fp16 parts (empty unless the set qualifies)
    sr::DevBuf<uint16_t> d_h2s_params, d_h2s_qdesc, d_h2s_ldesc, d_h2s_ref_params;
    sr::DevBuf<sr::SharedBlock> d_h2s_blocks;
    sr::DevBuf<float> d_h2s_center, d_h2s_scale, d_h2s_ref_center, d_h2s_ref_scale;
    sr::DevBuf<sr::ChunkDesc> d_h2s_ref_chunks;
    sr::DevBuf<int> d_h2s_ref_gcb;
    // Hybrid form of an ill-conditioned set (score.hpp): the mixtures whose expanded form would cancel in fp32 (tight and
    // far from the centre) as one sub-set on the direct-form vector engine, the rest as another on the matrix cores;
    // the two per-frame log-likelihoods are merged by a log-add-exp.  Empty unless the set needed it.
    // Model-group tables (chunk / block index per group) of recent scoring calls and their device copies: the number of groups
    // follows the batch's size, and a caller that alternates between sizes (the pieces of sr_multi_predict_pcm) must not re-upload
    // -- and synchronise the stream -- on every call.  A handful of entries, oldest replaced.
    struct GroupTable {
        std::vector<int> host;
        sr::DevBuf<int> dev;
    };
    std::vector<std::unique_ptr<GroupTable>> group_tables;
    size_t group_table_next = 0;
    sr::DevBuf<int> d_flush_models;  // per model {first record, records} of the vector layout (gmm_flush.hip; built on first use)
    std::unique_ptr<SRModelSet> hy_good, hy_bad;
    int hy_bad_mixtures = 0;         // mixtures of the set's largest model that went to the vector engine
    int device = -1;
};
