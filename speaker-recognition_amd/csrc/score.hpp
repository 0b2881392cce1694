// score.hpp -- host entry points of the scoring path shared between translation units.
#pragma once

#include "batch.hpp"
#include "gmm_model.hpp"

namespace sr {

struct ScoreOptions {
    int frames_per_lane = 0;   // 0 = auto; 1, 2 or 4 frames resident per lane
    int model_groups = 0;      // 0 = auto; workgroups per frame tile along the model axis
    int packed = 0;            // -1 = scalar v_fma_f32; 0 (auto) / 1 = v_pk_fma_f32, two frames per VGPR pair
    int engine = 0;            // 0 = auto; 1 = vector-ALU 2-FMA kernel; (2 = the fp32 matrix-core kernel of round 1, removed);
                               // 3 = split-bf16 (3 parts, 6 products) matrix-core kernel;
                               // 4 = split-bf16, shared-sigma form (sets whose models share sigma and weights)
                               // 5 = split-fp16 (2 parts, 3 products) matrix-core kernel;
                               // 6 = split-fp16, shared-sigma form
    int mfma_ft = 0;           // 32-frame column tiles per wave in the 4-wave generic split kernels (0 = one)
    int h2s_shape = 0;         // workgroup shape of the split-fp16 shared-sigma engine: 0 = automatic; 1 = 4 waves (three
                               // workgroups per CU); 2 = 12 waves (one per CU, one copy of the stream in LDS); 3 = 12 waves with the
                               // image loop software-pipelined inside each wave (gmm_score_h2p_kernel)
    int split_shape = 0;       // workgroup shape of the generic split-fp16 engine: 0 = automatic; 1 = 4 waves (gmm_score_split_kernel);
                               // 16 / 12 / 8 = gmm_score_splitp_kernel with that many waves (one 32-frame tile each, log-sum-exp pipelined
                               // under the next chunk's MFMAs; 16 and 12: one workgroup per CU, 8: two)
    int h2s_pack_tails = 1;    // 0: the pipelined shared-sigma kernel takes one tile per wave even when it is a ragged tail (A/B, tests)
    int h2s_force_exc = 0;     // testing: send every workgroup of the split-fp16 shared-sigma engine through its exception pass
    int flush_list_cap = 0;    // testing: capacity of the list of (tile, model) pairs in the partial-product band (0 = automatic);
                               // a pass that notes more re-runs with a list of the counted length
};

// The matrix-core engines are used when the expanded form is well conditioned in fp32 and the
// 32-mixture tiles are not mostly padding; otherwise the 2-FMA vector kernel (direct form).
constexpr double MFMA_MAX_AMP = 2000.0;      // max_k sum_d (mu'_d/sigma_d)^2
constexpr double MFMA_MAX_PAD_WASTE = 0.25;
// The two-part fp16 engines carry 22 significand bits per operand (error ~4x an fp32 FMA chain's per
// term, scripts/emulate_split.py) and fp16's 5-bit exponent: offered when the cancellation is
// moderate, every dimension's sigmas stay within a factor the gradual-underflow error analysis
// covers (HISTORY.md 2.1), and the scaled coefficients fit fp16.
constexpr double F16_MAX_AMP = 1000.0;
// Hybrid form: a set the expanded form is ill conditioned for (amp above the limits) because of FEW of its mixtures
// -- collapsed components at the sigma floor, outlier catchers -- is cut in two: those mixtures (at most
// HYBRID_MAX_BAD_FRACTION of them) run on the direct-form vector engine, the rest on the matrix cores.
constexpr double HYBRID_MAX_BAD_FRACTION = 0.25;
constexpr double F16_MAX_SIGMA_RATIO = 256.0;
constexpr double F16_MAX_COEF = 30000.0;
// internal scoring flag (beside SR_CLAMP_COMPAT): keep to the fp32-grade engines (EM, serving stream)
constexpr int SCORE_PRECISE = 0x200;
// internal: leave the partial-product band alone (the two halves of a hybrid set: their merge looks at the merged value)
constexpr int SCORE_NO_FLUSH = 0x400;
// Small result sets of callers that fetch them right away (fetch_results): gmm_finalize_kernel's last workgroup writes sums, argmax
// and the pass's two counters into page-locked host memory and releases a sequence number the host polls for -- no device-to-host
// copies, no stream synchronisation -- and clears the pass's counters for the next pass (no memset in front of it).
constexpr int SCORE_HOST_DELIVER = 0x800;
constexpr size_t HOST_DELIVER_MAX_BYTES = (size_t)64 << 10;
constexpr size_t HOST_DELIVER_MAX_UTTS = 256;      // every utterance's workgroup takes a ticket from ONE counter: 2000 utterances x 1 model
                                                   // (16 KiB of results) lost 80 us per pass to that queue, more than the copies cost
inline bool host_deliverable(size_t n_utt, size_t n_models) {
    return n_utt > 0 && n_utt <= HOST_DELIVER_MAX_UTTS && n_utt * n_models * sizeof(double) + n_utt * sizeof(int) <= HOST_DELIVER_MAX_BYTES;
}
// what the last workgroup leaves in host memory: this header, then double sums[U][S], then int argmax[U]
struct DeliverHeader {
    int oor, n_flush;
    unsigned seq;
    int pad;
};

struct MfmaLaunch {
    const float *X;
    const TileDesc *tiles;
    const float4 *params;
    const ChunkDesc *chunks;
    const int *group_chunk_begin;
    const float *center;
    const float *scale = nullptr;   // fp16 scheme: per-dimension power-of-two scale of x - center
    double *partial;
    float *frame_ll;
    int *oor_flag = nullptr;        // fp16 scheme: set when a frame saturated (|scaled x'| >= 255)
    int64_t n_frames;
    int dim, n_models, clamp, n_groups, n_tiles;
    float band_hi = -__builtin_inff();   // below it a frame goes to the partial-product path (lse.hpp); -inf: never
};
struct SharedLaunch {
    const float *X;
    const TileDesc *tiles;
    const uint16_t *params;
    const SharedBlock *blocks;
    const int *group_block_begin;
    const float *center;
    double *partial;
    float *frame_ll;
    int64_t n_frames;
    int dim, n_models, n_mix_tiles, clamp, n_groups, n_tiles;
    float band_hi = -__builtin_inff();
};
void launch_score_bx3_shared(const SharedLaunch &a, int KQ, int KL);
constexpr int H2P_ROUND_ITEMS = 96;     // the pipelined kernel's work table is padded to whole rounds of 8 workgroups x 12 waves
struct H2sLaunch {
    const float *X;
    const TileDesc *tiles;
    const uint16_t *params;
    const SharedBlock *blocks;
    const int *group_block_begin;
    const float *center, *scale;
    const uint16_t *q_desc, *l_desc;
    const float *ref_ll;
    double *partial;
    float *frame_ll;
    int *oor_flag;
    int n_work = 0;     // pipelined shape: `tiles` is TileTable::d_tiles_work -- the tiles, then this many work items (+ padding)
    int *exc_list;      // int2 [n_blocks][n_tiles] {tile, listed columns}, then int [n_blocks][n_tiles + 1] (the exception pass's plan)
    int *exc_count;     // [n_blocks] entries, then [n_blocks] items
    int n_blocks;
    int64_t n_frames;
    int dim, n_models, n_mix_tiles, clamp, n_groups, n_tiles;
    float log2_k;
    int force_exc;
    int shape = 0;          // 0: 4-wave workgroups; 1: 12-wave workgroups (`tiles` = 32-frame tiles); 2: 12 waves, pipelined (gmm_score_h2p_kernel)
    float band_hi = -__builtin_inff();
};
int launch_score_h2_shared(const H2sLaunch &a, int KQF, int KLF);
int h2s_resident_per_cu(int kqf, int klf, int shape);   // workgroups the kernel variant keeps resident per CU
int h2s_tiles_per_wg(int shape);                        // 32-frame tiles a workgroup of that shape takes
bool h2s_msplit_direct(int kqf, int klf);               // shape 3 runs as gmm_score_h2m_kernel (images straight into registers) for these chain lengths
bool h2s_pipelined_available(int kqf, int klf);         // shape 2 (12 waves, image loop software-pipelined inside each wave) exists for these chain lengths
// Minimum set size for the shared-sigma engine (blocks of SHARED_SB models; smaller sets would be
// mostly phantom models).
constexpr int SHARED_MIN_MODELS = 12;
constexpr int H2S_WIDE_SHAPE = 1;       // the one-workgroup-per-CU shape
constexpr int H2S_MSPLIT_SHAPE = 3;     // four waves on ONE tile, the block's models split between them: the smallest batches (round 4)
constexpr int H2S_PIPELINED_SHAPE = 2;  // the same with the image loop pipelined inside each wave: what the dispatcher takes for large batches
void launch_score_split(const MfmaLaunch &a, int scheme, int KS, int FT);   // a.params = the split image
int split_max_ft(int ks);
// gmm_score_splitp.hip: the same engines as ONE wide workgroup per CU (12 or 16 waves, a 32-frame tile each, the chunk's log-sum-exp
// pipelined under the next chunk's MFMAs); `a.tiles` = 32-frame tiles.  splitp_waves: waves per workgroup of the variant that exists
// for this layout (`want` = 0, 12 or 16), 0 = none.
int splitp_waves(int scheme, int ks, int want);
int splitp_resident_per_cu(int waves);
bool launch_score_splitp(const MfmaLaunch &a, int scheme, int KS, int waves, int chunks_per_model);
ScoreOptions &score_options();
const char *last_score_kernel();   // name of the kernel variant the last scoring call launched

// Device-resident results of the last scoring call (valid until the next one).
struct ScoreResult {
    const double *d_sums = nullptr;    // [U][S]
    const int *d_argmax = nullptr;     // [U]
    const float *d_frame_ll = nullptr; // [S][n_frames] when requested
    const int *d_oor = nullptr;        // fp16 engines: nonzero when a frame saturated -> results must be redone
    // reference clamp on: (tile, model) pairs with a frame whose value the reference's flushes of partial products may
    // change (lse.hpp), left out of `d_sums` by gmm_finalize_kernel -- `*d_flush_count` of them (it counts past
    // `flush_cap`), to be resolved by flush_resolve before the results are used (fetch_results does)
    const int *d_flush_count = nullptr;
    const int2 *d_flush_list = nullptr;
    int flush_cap = 0;
    const TileTable *tiles = nullptr;  // the tile table the pass ran on
    // SCORE_HOST_DELIVER honoured: where the results arrive (page-locked host memory) and the sequence number that says they have
    const volatile DeliverHeader *h_deliver = nullptr;
    unsigned deliver_seq = 0;
};

// Scores every utterance of `feat` against every model of `set`; leaves results on the device.
// `frame_ll_dst`: device buffer [S][n_frames] the per-frame values go to instead of the workspace's own.
ScoreResult score_device(SRModelSet &set, SRBatch &feat, bool want_frame_ll, int flags, float *frame_ll_dst = nullptr);
// Same, then copies what the caller asked for to host memory.
void score_batch_set(SRModelSet &set, SRBatch &feat, double *sums_out, int *argmax_out,
                     float *frame_ll_out, int flags);
// Results of the last scoring call -> host memory, through pinned staging buffers.
// Returns false when the fp16 engine reported saturated frames (nothing was copied out: score again
// with SCORE_PRECISE).
bool fetch_results(SRModelSet &set, SRBatch &feat, int flags, const ScoreResult &r, double *sums_out,
                   int *argmax_out, float *frame_ll_out);
// gmm_flush.hip: the frames of the noted (tile, model) pairs again with the reference's own linear-domain arithmetic;
// adds the tiles' sums to `d_sums`, redoes the argmax of the utterances touched, overwrites the per-frame values
void flush_resolve(SRModelSet &set, SRBatch &feat, const TileTable &tt, const int2 *d_list, int count, double *d_sums,
                   int *d_argmax, float *d_frame_ll);
// the same for results that already sit in host memory (sums[U][S], argmax[U] of the batch `feat`): patched on the host, one wait
void flush_resolve_host(SRModelSet &set, SRBatch &feat, const TileTable &tt, const int2 *d_list, int count, double *h_sums,
                        int *h_argmax);
int &flush_order_option();     // 2 = partial products as the reference DSO's compiler forms them (default), 1 = source order
void flush_stats(long *calls, long *pairs, long *frames);
// PCM batch -> MFCC -> CMVN/deltas -> all models -> sums + argmax on the host (abi.cpp; pipelined over
// chunks of utterances for large batches).
}  // namespace sr
struct SRMfcc;
namespace sr {
void predict_pcm(SRMfcc *m, SRModelSet *set, SRBatch *pcm, int nd, double *sums_out, int *argmax_out, int flags);
// Packs + uploads a model set on the current device.
void upload_model_set(SRModelSet &s);
// a GMM handle's own one-model set on the current device, packed and uploaded once (abi.cpp; invalidated by GMM::drop_single)
std::shared_ptr<SRModelSet> single_model_set(const GMM *g);
// the split-bf16 layout of a set that carries one (s.bx3), on the device: lazily, as every matrix-core layout (em.hip reads it too)
void ensure_bx3_layout(SRModelSet &s);
bool split_bf16_in_range(const SRModelSet &s);       // what score_device asks before it takes that engine
// Packs the layouts a set needs (all of them for small sets; for large ones the vector layout plus
// the one the dispatcher will pick, or the one forced by score_engine at creation time).
void pack_model_set(SRModelSet &s, const std::vector<const GMM *> &models);

}  // namespace sr
