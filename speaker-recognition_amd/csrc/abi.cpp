// abi.cpp -- extern "C" surface of lib/pygmm.so (declared in include/pygmm_hip.h).
// Part 1 reproduces the reference's ten symbols (src/gmm/src/pygmm.hh:28-41) on top of the
// HIP path; part 2 is the contiguous / batched interface.  Nothing throws across this file.
#include "../../include/pygmm_hip.h"

#include "batch.hpp"
#include "common.hpp"
#include "fork_proxy.hpp"
#include "gmm_model.hpp"
#include "mfcc.hpp"
#include "score.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace sr {
void burn_reference_rand(int count);     // kmeans_init.hip
void reference_rand_sample(int *out, int count);
int train_em(GMM &gmm, const GMM *ubm, const float *X, long n, int dim, const Parameter &param,
             long seed);
void set_em_stats_engine(int v);
void set_em_small_test_absent(int v);   // em_small.hip
int last_em_stats_engine();
void set_reference_side_effects(int v);
void set_kmeans_assign_engine(int v);
}
std::atomic<int> &stream_debug_capture_delay_ms();      // stream.cpp
std::atomic<int> &multi_merge_option();
namespace sr {
void kmeans_fast_stats(long *passes, long *rechecked);
int reference_side_effects();
}  // namespace sr

using namespace sr;

// ---- guards: park the message, never unwind into C ----
#define SR_TRY try {                                                     \
    std::lock_guard<std::recursive_mutex> _api_lock(api_mutex());
#define SR_CATCH(ret)                          \
    }                                          \
    catch (const std::exception &e) {          \
        set_error("%s", e.what());             \
        return ret;                            \
    }                                          \
    catch (...) {                              \
        set_error("unknown C++ exception");    \
        return ret;                            \
    }
#define SR_CATCH_VOID                                          \
    }                                                          \
    catch (const std::exception &e) {                          \
        set_error("%s", e.what());                             \
        fprintf(stderr, "pygmm.so: %s\n", e.what());           \
        return;                                                \
    }                                                          \
    catch (...) {                                              \
        set_error("unknown C++ exception");                    \
        return;                                                \
    }

namespace sr {
// The handle's one-model set on the current device: packed (every layout a small set carries) and uploaded once, until training
// changes the parameters.  The caller holds this device's lock.
std::shared_ptr<SRModelSet> single_model_set(const GMM *g) {
    if (!g) fail("null GMM handle");
    if (!g->trained()) fail("GMM has no parameters yet (train or load it first)");
    auto &slot = g->single[current_device()];          // one per device
    if (!slot || slot->device != ctx().device) {
        auto s = std::make_shared<SRModelSet>();
        pack_model_set(*s, {g});
        upload_model_set(*s);
        slot = s;
    }
    return slot;
}
}  // namespace sr
static SRModelSet &single_set(GMM *g) { return *sr::single_model_set(g); }

static std::unique_ptr<SRBatch> feature_batch(const float *X, int64_t n, int dim,
                                              const int64_t *offsets, int n_utt) {
    ensure_device();
    if (n < 0 || dim <= 0 || n_utt < 0) fail("bad batch shape");
    auto b = std::make_unique<SRBatch>();
    b->bind_device();
    b->kind = SRBatch::FEATURES;
    b->n_utt = n_utt;
    b->dim = dim;
    b->n_rows = n;
    b->offsets.assign(offsets, offsets + n_utt + 1);
    if (b->offsets.front() != 0 || b->offsets.back() != n) fail("offsets must run from 0 to n");
    for (int u = 0; u < n_utt; u++)
        if (b->offsets[u + 1] < b->offsets[u]) fail("offsets must be non-decreasing");
    b->data.upload(X, (size_t)n * dim);
    b->d_offsets.upload(b->offsets.data(), b->offsets.size());
    sync_stream();
    return b;
}

// double** rows -> contiguous fp32 (the reference deep-copies too: pygmm.cc:16-23)
static std::vector<float> rows_to_f32(double **X, long n, int dim) {
    if (n > 0 && !X) fail("null X_in");
    std::vector<float> out((size_t)n * dim);
    for (long i = 0; i < n; i++) {
        const double *r = X[i];
        for (int j = 0; j < dim; j++) out[(size_t)i * dim + j] = (float)r[j];
    }
    return out;
}

// The legacy ABI scores ONE model per call, and its callers loop over the speakers with the same
// utterance (gmmset.py:59-64, :95-99: S calls of score_all(x)): re-uploading x S times would make
// the drop-in path pay S H2D copies and S tile-table builds.  The last uploaded matrix stays on the
// device, keyed by shape and CONTENTS: a call's matrix is compared with the host copy kept beside the device one (memcmp: 5 us
// for 1000 x 39 frames on a hit, a few bytes on a miss).  (Through round 5 a 64-bit FNV-1a hash went first: a dependent
// multiply per 8 bytes, 20 us of a 60 us call, for a comparison that decides by itself.)  Matrices above LEGACY_CACHE_MAX_BYTES
// are not kept (neither copy outlives the call).
constexpr size_t LEGACY_CACHE_MAX_BYTES = (size_t)256 << 20;
struct LegacyFeatureCache {
    long n = -1;
    int dim = -1;
    std::vector<float> host;
    std::unique_ptr<SRBatch> batch;
};

namespace sr {
void score_one_local(GMM *g, const float *X, long n, int dim, float *ll_out, double *sum_out, int flags);
void score_models_local(GMM *const *models, int n_models, const float *X, long n, int dim, double *sums_out, int flags);
}
static std::unique_ptr<SRBatch> feature_batch(const float *X, int64_t n, int dim, const int64_t *offsets, int n_utt);
// One model against a contiguous fp32 matrix: here, or -- in a process that lost its GPU runtime to fork() -- in its helper.
static void score_one(GMM *g, const float *X, long n, int dim, float *ll_out, double *sum_out, int flags) {
    if (gpu_runtime_lost()) return fork_proxy_score(g, X, n, dim, ll_out, sum_out, flags);
    score_one_local(g, X, n, dim, ll_out, sum_out, flags);
}
static int train_one(GMM &gmm, const GMM *ubm, const float *X, long n, int dim, const Parameter &param, long seed) {
    if (gpu_runtime_lost()) return fork_proxy_train(gmm, ubm, X, n, dim, param, seed);
    return train_em(gmm, ubm, X, n, dim, param, seed);
}

void sr::score_one_local(GMM *g, const float *X, long n, int dim, float *ll_out, double *sum_out, int flags) {
    SRModelSet &set = single_set(g);
    if (dim != g->dim) fail("nr_dim %d does not match the model's dim %d", dim, g->dim);
    auto &cache = per_device<LegacyFeatureCache>();
    const size_t bytes = (size_t)n * dim * sizeof(float);
    const bool hit = cache.batch && cache.n == n && cache.dim == dim &&
                     cache.host.size() * sizeof(float) == bytes && std::memcmp(cache.host.data(), X, bytes) == 0;
    if (!hit) {
        const int64_t off[2] = {0, n};
        cache.batch.reset();                               // (frees the previous matrix before the new one is allocated)
        cache.batch = feature_batch(X, n, dim, off, 1);
        cache.n = n;
        cache.dim = dim;
        if (bytes <= LEGACY_CACHE_MAX_BYTES) cache.host.assign(X, X + (size_t)n * dim);
        else std::vector<float>().swap(cache.host);
    }
    double sum = 0.0;
    score_batch_set(set, *cache.batch, &sum, nullptr, ll_out, flags);
    if (sum_out) *sum_out = sum;
    if (bytes > LEGACY_CACHE_MAX_BYTES) {                  // too large to pin in HBM between calls
        cache.batch.reset();
        cache.n = -1;
    }
}

// Several models on ONE utterance in one fused pass (sr_score_models_f32): what gmmset.py:95-99's loop of score_all calls computes.
// The packed set of the last model list stays on the device, keyed by every model's identity and parameter generation.
namespace {
struct ModelsSetCache {
    std::vector<std::pair<uint64_t, uint64_t>> key;      // (uid, generation) per model
    std::unique_ptr<SRModelSet> set;
    std::unique_ptr<SRBatch> batch;                      // refilled per call (device buffers only grow)
};
}  // namespace
void sr::score_models_local(GMM *const *models, int n_models, const float *X, long n, int dim, double *sums_out, int flags) {
    if (!models || n_models <= 0) fail("empty model list");
    if (!sums_out) fail("null sums_out");
    if (n < 0 || dim <= 0) fail("bad frame matrix shape");
    if (n > 0 && !X) fail("null X");
    std::vector<std::pair<uint64_t, uint64_t>> key((size_t)n_models);
    for (int i = 0; i < n_models; i++) {
        const GMM *g = models[i];
        if (!g) fail("null GMM handle in model list");
        if (!g->trained()) fail("GMM has no parameters yet (train or load it first)");
        if (g->dim != dim) fail("nr_dim %d does not match model %d's dim %d", dim, i, g->dim);
        key[(size_t)i] = {g->uid, g->generation};
    }
    ensure_device();
    auto &cache = per_device<ModelsSetCache>();
    if (!cache.set || cache.key != key || cache.set->device != ctx().device) {
        cache.set.reset();
        auto s = std::make_unique<SRModelSet>();
        pack_model_set(*s, std::vector<const GMM *>(models, models + n_models));
        upload_model_set(*s);
        cache.set = std::move(s);
        cache.key = key;
    }
    const int64_t off[2] = {0, n};
    if (!cache.batch) {
        cache.batch = feature_batch(X, n, dim, off, 1);
    } else if (sr_batch_reset_features(cache.batch.get(), X, n, dim, off, 1) != 0) {
        fail("%s", last_error().c_str());
    }
    score_batch_set(*cache.set, *cache.batch, sums_out, nullptr, nullptr, flags);
}

extern "C" {

// ======================= Part 1: legacy symbols =======================

GMM *new_gmm(int nr_mixture, int covariance_type) {
    SR_TRY
    if (covariance_type != 1) fail("only diagonal matrix supported.");  // gmm.cc:211-215
    if (nr_mixture <= 0) fail("nr_mixture must be positive");
    GMM *g = new GMM();
    g->nr_mixtures = nr_mixture;
    g->covariance_type = covariance_type;
    return g;
    SR_CATCH(nullptr)
}

GMM *load(const char *model_file) {
    SR_TRY
    if (!model_file) fail("null model_file");
    std::ifstream fin(model_file, std::ios::binary);
    if (!fin) fail("cannot open model file '%s'", model_file);
    std::stringstream ss;
    ss << fin.rdbuf();
    auto g = std::make_unique<GMM>();
    gmm_parse_text(ss.str(), *g);
    burn_reference_rand(g->nr_mixtures);    // GMM::load seeds one Random per Gaussian from libc rand() (gmm.cc:671-676, gmm.hh:44)
    return g.release();
    SR_CATCH(nullptr)
}

void dump(GMM *gmm, const char *model_file) {
    SR_TRY
    if (!gmm || !model_file) fail("null argument to dump");
    std::ofstream fout(model_file, std::ios::binary);
    if (!fout) fail("cannot write model file '%s'", model_file);
    fout << gmm_format_text(*gmm);
    SR_CATCH_VOID
}

// pygmm.cc:31-41
static void print_param_block(const struct Parameter *param) {
    printf("nr_instance   :   %d\n", param->nr_instance);
    printf("nr_dim        :   %d\n", param->nr_dim);
    printf("nr_mixture    :   %d\n", param->nr_mixture);
    printf("min_covar     :   %f\n", param->min_covar);
    printf("threshold     :   %f\n", param->threshold);
    printf("nr_iteration  :   %d\n", param->nr_iteration);
    printf("init_with_kmeans: %d\n", param->init_with_kmeans);
    printf("concurrency   :   %d\n", param->concurrency);
    printf("verbosity     :   %d\n", param->verbosity);
}

void train_model(GMM *gmm, double **X_in, struct Parameter *param) {
    SR_TRY
    if (!gmm || !param) fail("null argument to train_model");
    if (reference_side_effects()) print_param_block(param);
    std::vector<float> X = rows_to_f32(X_in, param->nr_instance, param->nr_dim);
    if (train_one(*gmm, nullptr, X.data(), param->nr_instance, param->nr_dim, *param, -1) < 0)
        fail("%s", last_error().c_str());
    SR_CATCH_VOID
}

void train_model_from_ubm(GMM *gmm, GMM *ubm, double **X_in, struct Parameter *param) {
    SR_TRY
    if (!gmm || !ubm || !param) fail("null argument to train_model_from_ubm");
    if (reference_side_effects()) print_param_block(param);
    std::vector<float> X = rows_to_f32(X_in, param->nr_instance, param->nr_dim);
    if (train_one(*gmm, ubm, X.data(), param->nr_instance, param->nr_dim, *param, -1) < 0)
        fail("%s", last_error().c_str());
    SR_CATCH_VOID
}

double score_all(GMM *gmm, double **X_in, int nr_instance, int nr_dim, int /*concurrency*/) {
    SR_TRY
    std::vector<float> X = rows_to_f32(X_in, nr_instance, nr_dim);
    double sum = 0.0;
    score_one(gmm, X.data(), nr_instance, nr_dim, nullptr, &sum, SR_CLAMP_COMPAT);
    return sum;
    SR_CATCH(std::numeric_limits<double>::quiet_NaN())
}

void score_batch(GMM *gmm, double **X_in, double *prob_out, int nr_instance, int nr_dim,
                 int /*concurrency*/) {
    SR_TRY
    if (nr_instance > 0 && !prob_out) fail("null prob_out");
    std::vector<float> X = rows_to_f32(X_in, nr_instance, nr_dim);
    std::vector<float> ll((size_t)nr_instance);
    score_one(gmm, X.data(), nr_instance, nr_dim, ll.data(), nullptr, SR_CLAMP_COMPAT);
    for (int i = 0; i < nr_instance; i++) prob_out[i] = ll[i];
    SR_CATCH_VOID
}

double score_instance(GMM *gmm, double *x_in, int nr_dim) {
    SR_TRY
    if (!x_in) fail("null x_in");
    double *rows[1] = {x_in};
    std::vector<float> X = rows_to_f32(rows, 1, nr_dim);
    double sum = 0.0;
    score_one(gmm, X.data(), 1, nr_dim, nullptr, &sum, SR_CLAMP_COMPAT);
    return sum;
    SR_CATCH(std::numeric_limits<double>::quiet_NaN())
}

int get_dim(GMM *gmm) { return gmm ? gmm->dim : 0; }
int get_nr_mixtures(GMM *gmm) { return gmm ? gmm->nr_mixtures : 0; }

// ======================= Part 2: extensions =======================

const char *sr_last_error(void) { return last_error().c_str(); }

int sr_gpu_runtime_lost(void) { return gpu_runtime_lost() ? 1 : 0; }

int sr_device_count(void) { return visible_devices(); }

// No api lock here: these only move the calling thread between devices.
int sr_set_device(int device) {
    try {
        set_default_device(device);
        return 0;
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return -1;
    }
}

int sr_set_thread_device(int device) {
    try {
        set_thread_device(device);
        return 0;
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return -1;
    }
}

int sr_get_device(void) { return current_device(); }

int sr_device_synchronize(void) {
    SR_TRY
    ensure_device();
    SR_HIP(hipDeviceSynchronize());
    return 0;
    SR_CATCH(-1)
}

int sr_device_name(char *buf, int buflen) {
    SR_TRY
    ensure_device();
    hipDeviceProp_t prop;
    SR_HIP(hipGetDeviceProperties(&prop, ctx().device));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
    SR_CATCH(-1)
}

int sr_device_numa_node(int device) {
    try {
        return device_numa_node(device);
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return -1;
    }
}

int sr_bind_thread_near_device(int device) {
    try {
        return bind_thread_near_device(device);
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return -1;
    }
}

void sr_free_gmm(GMM *gmm) { delete gmm; }

GMM *sr_gmm_from_arrays(int K, int D, const double *weights, const double *mean, const double *sigma) {
    SR_TRY
    if (K <= 0 || D <= 0 || !weights || !mean || !sigma) fail("bad arguments to sr_gmm_from_arrays");
    auto g = std::make_unique<GMM>();
    g->nr_mixtures = K;
    g->dim = D;
    g->weights.assign(weights, weights + K);
    g->mean.assign(mean, mean + (size_t)K * D);
    g->sigma.assign(sigma, sigma + (size_t)K * D);
    for (double s : g->sigma)
        if (!(s > 0)) fail("sigma must be positive");
    return g.release();
    SR_CATCH(nullptr)
}

int sr_gmm_get_params(GMM *g, double *weights, double *mean, double *sigma) {
    SR_TRY
    if (!g || !g->trained()) fail("GMM has no parameters");
    if (weights) std::memcpy(weights, g->weights.data(), sizeof(double) * g->weights.size());
    if (mean) std::memcpy(mean, g->mean.data(), sizeof(double) * g->mean.size());
    if (sigma) std::memcpy(sigma, g->sigma.data(), sizeof(double) * g->sigma.size());
    return 0;
    SR_CATCH(-1)
}

int sr_gmm_dumps(GMM *g, char *buf, long buflen, long *needed) {
    SR_TRY
    if (!g) fail("null GMM handle");
    const std::string s = gmm_format_text(*g);
    if (needed) *needed = (long)s.size() + 1;
    if (buf && buflen > (long)s.size()) {
        std::memcpy(buf, s.c_str(), s.size() + 1);
        return 0;
    }
    return buf ? -2 : 0;
    SR_CATCH(-1)
}

GMM *sr_gmm_loads(const char *text) {
    SR_TRY
    if (!text) fail("null text");
    auto g = std::make_unique<GMM>();
    gmm_parse_text(text, *g);
    return g.release();
    SR_CATCH(nullptr)
}

int sr_score_frames_f32(GMM *gmm, const float *X, long n, int dim, float *ll_out, double *sum_out,
                        int flags) {
    SR_TRY
    if (n > 0 && !X) fail("null X");
    score_one(gmm, X, n, dim, ll_out, sum_out, flags);
    return 0;
    SR_CATCH(-1)
}

SRModelSet *sr_modelset_create(GMM *const *models, int n_models) {
    SR_TRY
    if (!models || n_models <= 0) fail("empty model list");
    std::vector<const GMM *> v(models, models + n_models);
    for (auto *m : v)
        if (!m) fail("null GMM handle in model list");
    auto s = std::make_unique<SRModelSet>();
    pack_model_set(*s, v);
    upload_model_set(*s);
    return s.release();
    SR_CATCH(nullptr)
}

void sr_modelset_free(SRModelSet *set) { delete set; }

// Conditioning of a packed set, as the engine dispatcher sees it (score.hpp): out[0] = amp (max_k
// sum_d (mu'_kd / sigma_kd)^2 of the expanded form), out[1] = padding waste of the 32-mixture tiles,
// out[2] = largest per-dimension sigma ratio, out[3] = largest scaled coefficient (fp16 schemes),
// out[4] = 1 when the set shares sigma and weights, out[5] = models, out[6] = device.
int sr_modelset_info(SRModelSet *set, double *out8) {
    SR_TRY
    if (!set || !out8) fail("null argument");
    for (int i = 0; i < 8; i++) out8[i] = 0.0;
    const bool shared = !set->h2s.params.empty() || !set->shared.params.empty();
    if (!set->h2s.params.empty()) {
        out8[0] = set->h2s.amp; out8[1] = set->h2s.pad_waste; out8[2] = set->h2s.sigma_ratio; out8[3] = set->h2s.coef_max;
    } else if (!set->h2.params.empty()) {
        out8[0] = set->h2.amp; out8[1] = set->h2.pad_waste; out8[2] = set->h2.sigma_ratio; out8[3] = set->h2.coef_max;
    } else if (!set->shared.params.empty()) {
        out8[0] = set->shared.amp; out8[1] = set->shared.pad_waste;
    } else if (!set->bx3.params.empty()) {
        out8[0] = set->bx3.amp; out8[1] = set->bx3.pad_waste; out8[2] = set->bx3.sigma_ratio;
    }
    out8[4] = shared ? 1.0 : 0.0;
    out8[5] = set->host.n_models;
    out8[6] = set->device;
    out8[7] = set->hy_good ? set->hy_bad_mixtures : 0;      // hybrid form: mixtures (of the largest model) on the vector engine
    return 0;
    SR_CATCH(-1)
}
int sr_modelset_size(SRModelSet *set) { return set ? set->host.n_models : 0; }
int sr_modelset_dim(SRModelSet *set) { return set ? set->host.dim : 0; }

static SRBatch *pcm_batch(const void *pcm, bool is_f32, const int64_t *sample_offsets, int n_utt) {
    ensure_device();
    if (n_utt < 0 || !sample_offsets) fail("bad PCM batch arguments");
    auto b = std::make_unique<SRBatch>();
    b->bind_device();
    b->kind = is_f32 ? SRBatch::PCMF32 : SRBatch::PCM16;
    b->n_utt = n_utt;
    b->offsets.assign(sample_offsets, sample_offsets + n_utt + 1);
    if (b->offsets.front() != 0) fail("sample_offsets[0] must be 0");
    for (int u = 0; u < n_utt; u++)
        if (b->offsets[u + 1] < b->offsets[u]) fail("sample_offsets must be non-decreasing");
    b->n_rows = b->offsets.back();
    if (b->n_rows > 0 && !pcm) fail("null PCM pointer");
    if (is_f32)
        b->data.upload(static_cast<const float *>(pcm), (size_t)b->n_rows);
    else
        b->pcm16.upload(static_cast<const int16_t *>(pcm), (size_t)b->n_rows);
    b->d_offsets.upload(b->offsets.data(), b->offsets.size());
    sync_stream();
    return b.release();
}

SRBatch *sr_batch_from_pcm(const int16_t *pcm, const int64_t *sample_offsets, int n_utt) {
    SR_TRY
    return pcm_batch(pcm, false, sample_offsets, n_utt);
    SR_CATCH(nullptr)
}

SRBatch *sr_batch_from_pcm_f32(const float *pcm, const int64_t *sample_offsets, int n_utt) {
    SR_TRY
    return pcm_batch(pcm, true, sample_offsets, n_utt);
    SR_CATCH(nullptr)
}

SRBatch *sr_batch_from_features(const float *X, int64_t n_frames, int dim,
                                const int64_t *frame_offsets, int n_utt) {
    SR_TRY
    if (!frame_offsets) fail("null frame_offsets");
    if (n_frames > 0 && !X) fail("null X");
    return feature_batch(X, n_frames, dim, frame_offsets, n_utt).release();
    SR_CATCH(nullptr)
}

int sr_batch_reset_features(SRBatch *b, const float *X, int64_t n_frames, int dim, const int64_t *frame_offsets, int n_utt) {
    SR_TRY
    if (!b || !frame_offsets) fail("null argument");
    if (b->kind != SRBatch::FEATURES) fail("sr_batch_reset_features needs a feature batch");
    if (n_frames < 0 || dim <= 0 || n_utt < 0) fail("bad batch shape");
    if (frame_offsets[0] != 0 || frame_offsets[n_utt] != n_frames) fail("offsets must run from 0 to n");
    for (int u = 0; u < n_utt; u++)
        if (frame_offsets[u + 1] < frame_offsets[u]) fail("offsets must be non-decreasing");
    if (n_frames > 0 && !X) fail("null X");
    b->bind_device();
    const bool same_layout = b->n_utt == n_utt && b->offsets.size() == (size_t)n_utt + 1 && std::equal(b->offsets.begin(), b->offsets.end(), frame_offsets);
    b->n_utt = n_utt;
    b->dim = dim;
    b->n_rows = n_frames;
    // (device buffers only ever grow.)  A serving-sized refill goes through the batch's page-locked copies and is left in flight: the
    // caller's matrix is its own again when the call returns, whatever is queued next on the stream runs behind the transfer
    const size_t n_val = (size_t)n_frames * dim;
    const bool staged = n_val * sizeof(float) <= ((size_t)4 << 20) && ((size_t)n_utt + 1) * sizeof(int64_t) <= STAGED_TABLE_MAX_BYTES;
    if (!same_layout) {
        b->offsets.assign(frame_offsets, frame_offsets + n_utt + 1);
        b->invalidate_tiles();
        if (staged) b->stage_offsets.send(b->d_offsets, b->offsets.data(), b->offsets.size());
        else b->d_offsets.upload(b->offsets.data(), b->offsets.size());
    }
    if (staged) {
        b->stage_rows.send(b->data, X, n_val);
        return 0;
    }
    b->data.upload(X, n_val);
    sync_stream();
    return 0;
    SR_CATCH(-1)
}

// A serving decision's worth of PCM (<= 4 MB): through the batch's page-locked copy, transfer left in flight (batch.hpp) -- the stream
// synchronisation sr_batch_update_pcm used to end with was a sixth of a single-utterance decision (round 6).  false: too large, the
// caller uploads and waits.
static bool stage_small_pcm(SRBatch *b, const int16_t *pcm, int64_t n_samples) {
    if ((size_t)n_samples * sizeof(int16_t) > ((size_t)4 << 20) || n_samples <= 0) return false;
    if (!b->stage_done.e) SR_HIP(hipEventCreateWithFlags(&b->stage_done.e, hipEventDisableTiming));
    else if (hipEventQuery(b->stage_done.e) != hipSuccess) SR_HIP(hipEventSynchronize(b->stage_done.e));
    if ((size_t)n_samples > b->h_stage.n) b->h_stage.ensure((size_t)n_samples + (size_t)n_samples / 4);   // (headroom: as StagedUpload)
    if ((size_t)n_samples > b->pcm16.n) b->pcm16.ensure((size_t)n_samples + (size_t)n_samples / 4);
    g_devbuf_epoch++;                       // (contents changed: a captured graph that depends on them is re-captured, as upload())
    // more than 1 MB: in pieces of 256 K samples, a piece's DMA under the host's copy of the next one (64 utterances x 3 s:
    // 0.936 -> 0.905 ms per call; a second piece costs a small batch its 5 us)
    const int64_t PIECE = n_samples <= ((int64_t)512 << 10) ? n_samples : ((int64_t)256 << 10);
    for (int64_t at = 0; at < n_samples; at += PIECE) {
        const size_t n = (size_t)std::min<int64_t>(PIECE, n_samples - at);
        std::memcpy(b->h_stage.p + at, pcm + at, n * sizeof(int16_t));
        SR_HIP(hipMemcpyAsync(b->pcm16.p + at, b->h_stage.p + at, n * sizeof(int16_t), hipMemcpyHostToDevice, ctx().stream));
    }
    SR_HIP(hipEventRecord(b->stage_done.e, ctx().stream));
    return true;
}

int sr_batch_update_pcm(SRBatch *b, const int16_t *pcm, int64_t n_samples) {
    SR_TRY
    if (!b || !pcm) fail("null argument");
    if (b->kind != SRBatch::PCM16) fail("sr_batch_update_pcm needs an int16 PCM batch");
    if (n_samples != b->n_rows) fail("sample count %lld does not match the batch (%lld)", (long long)n_samples, (long long)b->n_rows);
    b->bind_device();
    if (stage_small_pcm(b, pcm, n_samples)) return 0;
    b->pcm16.upload(pcm, (size_t)n_samples);
    sync_stream();
    return 0;
    SR_CATCH(-1)
}

int sr_batch_reset_pcm(SRBatch *b, const int16_t *pcm, const int64_t *sample_offsets, int n_utt) {
    SR_TRY
    if (!b || !sample_offsets) fail("null argument");
    if (b->kind != SRBatch::PCM16) fail("sr_batch_reset_pcm needs an int16 PCM batch");
    if (n_utt < 0 || sample_offsets[0] != 0) fail("bad PCM batch arguments");
    for (int u = 0; u < n_utt; u++)
        if (sample_offsets[u + 1] < sample_offsets[u]) fail("sample_offsets must be non-decreasing");
    const int64_t n = sample_offsets[n_utt];
    if (n > 0 && !pcm) fail("null PCM pointer");
    b->bind_device();
    b->n_utt = n_utt;
    b->offsets.assign(sample_offsets, sample_offsets + n_utt + 1);
    b->n_rows = n;
    b->invalidate_tiles();
    // (device buffers only ever grow.)  A decision's worth of samples and their offsets: page-locked copies, left in flight
    if (((size_t)n_utt + 1) * sizeof(int64_t) <= STAGED_TABLE_MAX_BYTES && stage_small_pcm(b, pcm, n)) {
        b->stage_offsets.send(b->d_offsets, b->offsets.data(), b->offsets.size());
        return 0;
    }
    b->pcm16.upload(pcm, (size_t)n);
    b->d_offsets.upload(b->offsets.data(), b->offsets.size());
    sync_stream();
    return 0;
    SR_CATCH(-1)
}

void sr_batch_free(SRBatch *b) { delete b; }
int sr_batch_num_utterances(SRBatch *b) { return b ? b->n_utt : 0; }
int64_t sr_batch_num_rows(SRBatch *b) { return b ? b->n_rows : 0; }
int sr_batch_dim(SRBatch *b) { return (b && b->kind == SRBatch::FEATURES) ? b->dim : 0; }

int sr_batch_offsets(SRBatch *b, int64_t *offsets_out) {
    SR_TRY
    if (!b || !offsets_out) fail("null argument");
    std::memcpy(offsets_out, b->offsets.data(), sizeof(int64_t) * b->offsets.size());
    return 0;
    SR_CATCH(-1)
}

int sr_batch_download(SRBatch *b, float *out) {
    SR_TRY
    if (!b || !out) fail("null argument");
    if (b->kind != SRBatch::FEATURES) fail("only feature batches can be downloaded");
    b->bind_device();
    b->data.download(out, (size_t)b->n_rows * b->dim);
    sync_stream();
    return 0;
    SR_CATCH(-1)
}

int sr_score_batch_set(SRModelSet *set, SRBatch *features, double *sums_out, int *argmax_out,
                       float *frame_ll_out, int flags) {
    SR_TRY
    if (!set || !features) fail("null argument");
    score_batch_set(*set, *features, sums_out, argmax_out, frame_ll_out, flags);
    return 0;
    SR_CATCH(-1)
}

SRMfcc *sr_mfcc_create(double fs, double win_length_ms, double win_shift_ms, int fft_size,
                       int n_filters, int n_ceps, double pre_emphasis) {
    SR_TRY
    return new SRMfcc(fs, win_length_ms, win_shift_ms, fft_size, n_filters, n_ceps, pre_emphasis);
    SR_CATCH(nullptr)
}

int sr_mfcc_set_lpc(SRMfcc *m, int n_lpc) {
    SR_TRY
    if (!m) fail("null extractor");
    if (n_lpc != 0 && n_lpc != 10 && n_lpc != 12 && n_lpc != 15 && n_lpc != 16 && n_lpc != 20)
        fail("LPC order %d is not instantiated (10, 12, 15, 16, 20; 0 = off)", n_lpc);
    m->n_lpc = n_lpc;
    return 0;
    SR_CATCH(-1)
}

void sr_mfcc_free(SRMfcc *m) { delete m; }
int sr_mfcc_frame_len(SRMfcc *m) { return m ? m->frame_len : 0; }
int sr_mfcc_frame_shift(SRMfcc *m) { return m ? m->frame_shift : 0; }
int64_t sr_mfcc_num_frames(SRMfcc *m, int64_t n_samples) { return m ? mfcc_num_frames(*m, n_samples) : 0; }

int sr_mfcc_tables(SRMfcc *m, double *window, double *melbank, double *dct) {
    SR_TRY
    if (!m) fail("null extractor");
    if (window) std::memcpy(window, m->window.data(), sizeof(double) * m->window.size());
    if (melbank) std::memcpy(melbank, m->melbank.data(), sizeof(double) * m->melbank.size());
    if (dct) std::memcpy(dct, m->dct.data(), sizeof(double) * m->dct.size());
    return 0;
    SR_CATCH(-1)
}

SRBatch *sr_mfcc_extract_batch(SRMfcc *m, SRBatch *pcm, int nd, int cmvn) {
    SR_TRY
    if (!m || !pcm) fail("null argument");
    if (!cmvn && nd != 0) fail("raw cepstra (cmvn=0) come without deltas");
    auto out = std::make_unique<SRBatch>();
    mfcc_extract_batch(*m, *pcm, nd, cmvn, *out);
    return out.release();
    SR_CATCH(nullptr)
}

// The fused serving step: PCM batch -> MFCC -> CMVN / deltas -> all models -> sums + argmax on the host, one pass.
// (Rounds 1-3 carried an option that cut the batch into chunks and ran the feature kernels of chunk i + 1 on a second stream under the
// scoring of chunk i.  It measured slower on every workload -- configs[1] 5.12 ms in one pass against 5.17 / 5.43 / 5.87 ms with 2 / 4 /
// 8 chunks, configs[2] 337 against 350 ms and later 267 against 538: next to the power-capped scoring kernel the feature launches stretch
// from 20 to 271 ms and the scoring chunks pay their tails, profiles/r02_overlap.txt -- and its second stream did not wait for the
// uploads of sr_multi_predict_pcm's pieces.  Removed in round 4.)
static void predict_unpipelined(SRMfcc *m, SRModelSet *set, SRBatch *pcm, int nd, double *sums_out, int *argmax_out,
                                int flags) {
    SRBatch *feat_ws = &per_device<SRBatch>();   // reused across steps: the serving loop allocates nothing
    mfcc_extract_batch(*m, *pcm, nd, 1, *feat_ws);
    // (small result sets land in host memory by themselves: SCORE_HOST_DELIVER, score.hpp)
    const int deliver = (sums_out && argmax_out && host_deliverable((size_t)pcm->n_utt, (size_t)set->host.n_models)) ? SCORE_HOST_DELIVER : 0;
    const ScoreResult r = score_device(*set, *feat_ws, false, flags | deliver);
    if (!fetch_results(*set, *feat_ws, flags, r, sums_out, argmax_out, nullptr)) {
        // a frame left the fp16 engine's range: score the batch again on the fp32-grade engines
        const ScoreResult r2 = score_device(*set, *feat_ws, false, flags | SCORE_PRECISE);
        fetch_results(*set, *feat_ws, flags | SCORE_PRECISE, r2, sums_out, argmax_out, nullptr);
    }
}

}  // extern "C"

namespace sr {
void predict_pcm(SRMfcc *m, SRModelSet *set, SRBatch *pcm, int nd, double *sums_out, int *argmax_out, int flags) {
    if (!m || !set || !pcm) fail("null argument");
    ensure_device();
    predict_unpipelined(m, set, pcm, nd, sums_out, argmax_out, flags);
}
}  // namespace sr

extern "C" {

int sr_predict_pcm_batch(SRMfcc *m, SRModelSet *set, SRBatch *pcm, int nd, double *sums_out,
                         int *argmax_out, int flags) {
    SR_TRY
    predict_pcm(m, set, pcm, nd, sums_out, argmax_out, flags);
    return 0;
    SR_CATCH(-1)
}

int sr_train_f32(GMM *gmm, GMM *ubm_or_null, const float *X, long n, int dim,
                 const struct Parameter *param, long seed) {
    SR_TRY
    if (!gmm || !X || !param) fail("null argument to sr_train_f32");
    return train_one(*gmm, ubm_or_null, X, n, dim, *param, seed);
    SR_CATCH(-1)
}

int sr_score_models_f32(GMM *const *models, int n_models, const float *X, long n_frames, int dim, double *sums_out, int flags) {
    SR_TRY
    if (gpu_runtime_lost()) fork_proxy_score_models(models, n_models, X, n_frames, dim, sums_out, flags);
    else score_models_local(models, n_models, X, n_frames, dim, sums_out, flags);
    return 0;
    SR_CATCH(-1)
}

int sr_hbm_copy_gbps(size_t bytes, int iters, double *gbps_out) {
    SR_TRY
    if (!gbps_out || bytes == 0 || iters <= 0) fail("bad arguments to sr_hbm_copy_gbps");
    ensure_device();
    DevBuf<char> a(bytes), b(bytes);
    SR_HIP(hipMemsetAsync(a.p, 1, bytes, ctx().stream));
    hipEvent_t e0, e1;
    SR_HIP(hipEventCreate(&e0));
    SR_HIP(hipEventCreate(&e1));
    SR_HIP(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, ctx().stream));   // warm
    SR_HIP(hipEventRecord(e0, ctx().stream));
    for (int i = 0; i < iters; i++)
        SR_HIP(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, ctx().stream));
    SR_HIP(hipEventRecord(e1, ctx().stream));
    SR_HIP(hipStreamSynchronize(ctx().stream));
    float ms = 0.f;
    SR_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gbps_out = 2.0 * (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
    return 0;
    SR_CATCH(-1)
}

int sr_profile_enable(int on) {
    SR_TRY
    if (on) {
        ensure_device();
        profile_prewarm();
    }
    ctx().profiling = on != 0;
    return 0;
    SR_CATCH(-1)
}

int sr_profile_reset(void) {
    SR_TRY
    profile_reset();
    return 0;
    SR_CATCH(-1)
}

int sr_profile_get(int kind, double *total_ms, long *launches) {
    SR_TRY
    profile_get(kind, total_ms, launches);
    return 0;
    SR_CATCH(-1)
}

int64_t sr_ltsd_num_windows(int64_t n_samples, int winsize) {
    SR_TRY
    return ltsd_num_windows(n_samples, winsize);
    SR_CATCH(-1)
}

int sr_ltsd_noise_spectrum(SRBatch *noise, int winsize, float *avg_amp_out) {
    SR_TRY
    if (!noise || !avg_amp_out) fail("null argument");
    ltsd_noise_spectrum(*noise, winsize, avg_amp_out);
    return 0;
    SR_CATCH(-1)
}

int sr_ltsd_compute(SRBatch *pcm, int winsize, int order, const float *noise_amp, float *ltsd_out,
                    int64_t *win_offsets_out) {
    SR_TRY
    if (!pcm || !noise_amp || !ltsd_out) fail("null argument");
    ltsd_compute(*pcm, winsize, order, noise_amp, ltsd_out, win_offsets_out);
    return 0;
    SR_CATCH(-1)
}

int sr_set_option(const char *key, long value) {
    SR_TRY
    if (!key) fail("null key");
    const std::string k(key);
    if (k == "score_frames_per_lane") {
        if (value != 0 && value != 1 && value != 2 && value != 4) fail("frames per lane must be 0/1/2/4");
        score_options().frames_per_lane = (int)value;
    } else if (k == "score_model_groups") {
        score_options().model_groups = (int)value;
    } else if (k == "score_packed") {
        score_options().packed = (int)value;
    } else if (k == "score_engine") {
        if (value < 0 || value > 6 || value == 2)
            fail("score_engine must be 0 (auto), 1 (vector ALU), 3 (split-bf16 matrix cores), 4 (split-bf16, shared-sigma form), "
                 "5 (split-fp16 matrix cores) or 6 (split-fp16, shared-sigma form); 2 was the fp32 matrix-core engine, removed in round 5 "
                 "(never selected: 1.45-1.8x slower than split-bf16 at the same accuracy)");
        score_options().engine = (int)value;
    } else if (k == "score_h2s_shape") {
        if (value < 0 || value > 4)
            fail("score_h2s_shape must be 0 (automatic), 1 (4-wave workgroups), 2 (12-wave workgroups), 3 (12 waves, image loop pipelined inside the wave) "
                 "or 4 (4 waves on one tile, the block's models split between them)");
        score_options().h2s_shape = (int)value;
    } else if (k == "flush_list_cap") {
        if (value < 0) fail("flush_list_cap must be >= 0");
        score_options().flush_list_cap = (int)value;
    } else if (k == "score_h2s_pack_tails") {
        if (value != 0 && value != 1) fail("score_h2s_pack_tails must be 0 or 1");
        score_options().h2s_pack_tails = (int)value;
    } else if (k == "score_h2s_force_exc") {
        score_options().h2s_force_exc = value != 0;
    } else if (k == "score_split_shape") {
        if (value != 0 && value != 1 && value != 8 && value != 12 && value != 16)
            fail("score_split_shape must be 0 (automatic), 1 (4-wave workgroups) or 8 / 12 / 16 (waves of the wide, pipelined form)");
        score_options().split_shape = (int)value;
    } else if (k == "score_mfma_ft") {
        if (value < 0 || value > 4) fail("score_mfma_ft must be 0..4");
        score_options().mfma_ft = (int)value;
    } else if (k == "flush_order") {
        if (value != 1 && value != 2)
            fail("flush_order must be 2 (partial products as the reference DSO's compiler forms them: even / odd dimensions) or "
                 "1 (the source's order, gmm.cc:192-195)");
        flush_order_option() = (int)value;
    } else if (k == "kmeans_assign_engine") {
        if (value != 0 && value != 1) fail("kmeans_assign_engine must be 0 (fast full search, exact pass for what it cannot decide) or 1 (exact pass only)");
        set_kmeans_assign_engine((int)value);
    } else if (k == "reference_side_effects") {
        if (value != 0 && value != 1) fail("reference_side_effects must be 0 or 1");
        set_reference_side_effects((int)value);
    } else if (k == "em_stats_engine") {
        if (value < 0 || value > 3)
            fail("em_stats_engine must be 0 (automatic), 1 (vector ALU), 2 (fp64 matrix cores, responsibilities on the vector ALU) or "
                 "3 (automatic, an iteration per launch: no whole-fit kernel)");
        set_em_stats_engine((int)value);
    } else if (k == "multi_merge_same_device") {
        multi_merge_option().store(value != 0);
    } else if (k == "multi_numa_bind") {
        if (value != 0 && value != 1) fail("multi_numa_bind must be 0 or 1");
        numa_bind_option().store((int)value);
    } else if (k == "debug_capture_delay_ms") {
        if (value < 0 || value > 1000) fail("debug_capture_delay_ms must be 0 .. 1000");
        stream_debug_capture_delay_ms().store((int)value);         // test hook (tests/test_gpu_pipeline.py)
    } else if (k == "debug_em_small_absent_workgroup") {
        if (value != 0 && value != 1) fail("debug_em_small_absent_workgroup must be 0 or 1");
        set_em_small_test_absent((int)value);                       // test hook (tests/test_gpu_em_small.py): a workgroup of the whole-fit kernel stays away from a barrier
    } else if (k == "debug_helper_max_models") {
        if (value < 0) fail("debug_helper_max_models must be >= 0 (0: the default)");
        fork_proxy_set_max_models(value);                           // test hook (tests/test_gpu_fork.py): applies in the helper, where it is forwarded
    } else if (k == "mfcc_generic") {
        mfcc_set_force_generic(value != 0);
    } else if (k == "mfcc_precision") {
        if (value != 0 && value != 2)
            fail("mfcc_precision must be 2 (float64 spectrum, ln and DCT for every frame: the reference's arithmetic, MFCC.py:59-70) or "
                 "0 (fp32 throughout: ~1.5x faster, features up to 2e-2 off on voices whose mel bands lie > 60 dB apart)");
        mfcc_set_precision((int)value);
    } else {
        fail("unknown option '%s'", key);
    }
    fork_proxy_note_option(key, value);
    return 0;
    SR_CATCH(-1)
}

const char *sr_last_score_kernel(void) { return last_score_kernel(); }
int sr_last_em_stats_engine(void) { return sr::last_em_stats_engine(); }

void sr_flush_stats(long *calls, long *pairs, long *frames) { flush_stats(calls, pairs, frames); }

void sr_kmeans_fast_stats(long *passes, long *rechecked) { kmeans_fast_stats(passes, rechecked); }

int sr_mfma_peak_probe(double ms_target, double *tflops, double *mhz) {
    SR_TRY
    if (!(ms_target > 0.0) || ms_target > 2000.0) fail("ms_target must be in (0, 2000]");
    mfma_peak_probe(ms_target, tflops, mhz);
    return 0;
    SR_CATCH(-1)
}

int sr_mfma_streamed_probe(double ms_target, double *tflops, double *mhz) {
    SR_TRY
    if (!(ms_target > 0.0) || ms_target > 2000.0) fail("ms_target must be in (0, 2000]");
    mfma_peak_probe(ms_target, tflops, mhz, 1);
    return 0;
    SR_CATCH(-1)
}

int sr_reference_rand_sample(int *out, int count) {
    SR_TRY
    if (!out || count < 0) fail("bad arguments");
    reference_rand_sample(out, count);
    return 0;
    SR_CATCH(-1)
}

}  // extern "C"
