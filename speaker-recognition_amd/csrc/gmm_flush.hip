// gmm_flush.hip -- the reference's flush-to-zero decisions on PARTIAL products.
//
// Gaussian::probability_of_fast_exp (src/gmm/src/gmm.cc:176-202) multiplies D per-dimension factors
//   p_i = remez5(-d_i^2 / (2 s_i^2)) / (sqrt(2 pi) s_i)          (fastexp.cc:99-212 for the exp)
// in the LINEAR domain, and the reference DSO runs with FTZ/DAZ (its -ffast-math start-up object): an
// intermediate below DBL_MIN is exactly 0 and stays 0, even when later factors > 1 (s < 0.399: the rule for
// delta features) would have lifted the mixture's full product back above DBL_MIN; a dimension whose exponent
// reaches fastexp.cc's floor (:104-105,128-131) gives e_i = 0 outright.  The scoring engines work in the log
// domain on full products (lse.hpp) and cannot see any of that -- but they can tell which frames it could
// matter for, and poison the partial sum of such a frame's (tile, model); gmm_finalize_kernel leaves those
// out of the utterance sums and lists them, and they come here: every frame of the tile against that model,
// one wave per frame, a lane per mixture, the reference's arithmetic restated operation by operation in
// float64 with explicit flushes; the tile's sum is then added to the utterance's.
//
// WHICH intermediates exist is the compiler's choice under -ffast-math.  order 2 (default) is what g++ 11
// -O3 -ffast-math -msse2 -- the reference's flags, oracle/Makefile -- emits (read off the disassembly,
// pinned by tests/golden/make_flush_golden.py on the DSO): b_i = (x-m)(m-x) 0.5 / (s s);
// p_i = (e_i * 0.3989422804014327) / s_i; one running product over the even and one over the odd dimensions,
// even * odd, then the last dimension of an odd D, then * w_k (gmm.cc:241).  order 1 is the source's own
// order (one running product over the dimensions, p_i = e_i / (sqrt(2 pi) s_i)), for a DSO built by a compiler
// that does not reassociate; sr_set_option("flush_order", 1).
//
// Inputs are what the device already holds: the fp32 features and the vector engine's parameter records
// (s = sqrt(log2e/2)/sigma, m = -(mu - centre) s, c = log2e (ln w - sum ln(sqrt(2 pi) sigma)); gmm_model.hpp),
// widened to float64.  That moves a decision quantity by ~1e-4 nats (the goldens keep 0.03 away).
#include "lse.hpp"
#include "score.hpp"
#include "wave_ops.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace sr {

namespace {

struct FlushModel {
    uint32_t offset_f4;   // first record of the model in the vector-engine parameter buffer
    int32_t n_records;    // of KB = 4 mixtures
};

__device__ __forceinline__ double ftz(double v) { return fabs(v) < DBL_MIN ? 0.0 : v; }

// remez5_0_log2_sse, fastexp.cc:99-212, one value, FTZ on the last product
__device__ __forceinline__ double remez5_ftz(double x) {
    const double maxlog = 7.09782712893383996843e2, minlog = -7.08396418532264106224e2;
    x = fmin(x, maxlog);
    x = fmax(x, minlog);
    double a = x * 1.4426950408889634073599;
    if (a < 0) a -= 1.0;
    const int k = (int)a;                              // truncation, as _mm_cvttpd_epi32
    const double p = (double)k;
    x -= p * 6.93145751953125E-1;
    x -= p * 1.42860682030941723212E-6;
    a = x * 1.185268231308989403584147407056378360798378534739e-2 + 3.87412011356070379615759057344100690905653320886699e-2;
    a = a * x + 0.16775408658617866431779970932853611481292418818223;
    a = a * x + 0.49981934577169208735732248650232562589934399402426;
    a = a * x + 1.00001092396453942157124178508842412412025643386873;
    a = a * x + 0.99999989311082729779536722205742989232069120354073;
    const unsigned long long bits = (unsigned long long)(unsigned)(k + 1023) << 52;
    return ftz(a * __longlong_as_double((long long)bits));
}

// One wave per frame of a noted (tile, model) pair; lane l takes mixtures l, l + 64, ...
// grid = (pairs of this batch, ceil(frames per tile / 4)), 4 waves per workgroup; exact_out[pair][frame in tile].
// The frame's centred row sits in LDS as float64 (dynamic: 4 waves x dim doubles); rows too wide for that (XG) are
// re-formed from global memory at every use -- the same values, the reference has no limit on dim (gmm.cc:40-51).
constexpr int FLUSH_LDS_MAX_DIM = 1024;
template <int ORDER, bool XG>
__global__ __launch_bounds__(256)
void gmm_flush_exact_kernel(const float *__restrict__ X, const float *__restrict__ center,
                            const float *__restrict__ params, const FlushModel *__restrict__ models,
                            const TileDesc *__restrict__ tiles, const int2 *__restrict__ pairs, int dim, int dp,
                            int frames_per_tile, int64_t n_frames, float band_hi, float *__restrict__ exact_out,
                            float *__restrict__ frame_ll) {
    extern __shared__ double xs_all[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double *xs = xs_all + (size_t)wave * (XG ? 0 : dim);
    const double SQRT_HALF_LOG2E = 0.84932180028801907;        // sqrt(log2(e) / 2)
    const double LN2 = 0.69314718055994530942;
    const double SQRT_2_PI = 2.5066282746310002;               // gmm.cc:22
    const double INV_SQRT_2_PI = 0.3989422804014327;           // the DSO's folded constant
    const size_t rec_f = ((size_t)2 * dp + 1) * 4;
    const int2 pr = pairs[blockIdx.x];
    const TileDesc tile = tiles[pr.x];
    const int j = blockIdx.y * 4 + wave;                        // frame inside the tile
    if (j >= tile.count) return;                                // (whole wave; waves only meet through their own xs slab)
    const int64_t row = tile.start + j;
    if (!XG) {
        for (int d = lane; d < dim; d += 64) xs[d] = (double)X[row * dim + d] - (double)center[d];
        wave_sync();
    }
    auto xv = [&](int d) -> double { return XG ? (double)X[row * dim + d] - (double)center[d] : xs[d]; };
    const FlushModel fm = models[pr.y];
    // ---- phase 1: the frame's value in the log domain (float64, full-product underflow rule of lse.hpp: what the
    //      engines compute, 2 FMAs per mixture and dimension).  Most frames of a noted tile are ordinary ones that
    //      merely share the tile with the frame that was in the band: their value stands.
    const double MINLOG2 = -708.396418532264106224 * 1.4426950408889634073599;
    double vmax = -INFINITY;
    for (int k = lane; k < fm.n_records * KB; k += 64) {
        const float *rec = params + ((size_t)fm.offset_f4 * 4 + (size_t)(k / KB) * rec_f);
        const int jj = k % KB;
        const float c = rec[(size_t)2 * dp * 4 + jj];
        if (!(c > NEG_BIG)) continue;
        double q = 0.0;
        for (int d = 0; d < dim; d++) {
            const double t = xv(d) * (double)rec[d * 8 + jj * 2] + (double)rec[d * 8 + jj * 2 + 1];
            q += t * t;
        }
        vmax = fmax(vmax, (double)c - q);
    }
    for (int o = 32; o > 0; o >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, o));
    double s1 = 0.0;
    for (int k = lane; k < fm.n_records * KB; k += 64) {
        const float *rec = params + ((size_t)fm.offset_f4 * 4 + (size_t)(k / KB) * rec_f);
        const int jj = k % KB;
        const float c = rec[(size_t)2 * dp * 4 + jj];
        if (!(c > NEG_BIG)) continue;
        double q = 0.0;
        for (int d = 0; d < dim; d++) {
            const double t = xv(d) * (double)rec[d * 8 + jj * 2] + (double)rec[d * 8 + jj * 2 + 1];
            q += t * t;
        }
        const double v = (double)c - q;
        if (v >= MINLOG2) s1 += exp2(v - vmax);
    }
    s1 = wave_sum_f64(s1);
    const double ll1 = vmax < MINLOG2 ? (double)LSE_LN_1E_15 : LN2 * (vmax + log2(s1));
    if (!(ll1 < (double)band_hi)) {                      // (also a NaN frame: it stays NaN)
        if (lane == 0) {
            const float ll = (float)ll1;
            exact_out[(int64_t)blockIdx.x * frames_per_tile + j] = ll;
            if (frame_ll) frame_ll[(int64_t)pr.y * n_frames + row] = ll;
        }
        return;
    }
    // ---- phase 2: the band.  The reference's own arithmetic.
    double sum = 0.0;
    for (int k = lane; k < fm.n_records * KB; k += 64) {
        const float *rec = params + ((size_t)fm.offset_f4 * 4 + (size_t)(k / KB) * rec_f);
        const int jj = k % KB;
        const float c = rec[(size_t)2 * dp * 4 + jj];
        if (!(c > NEG_BIG)) continue;                         // tile padding, weight 0: adds exactly 0 (gmm.cc:241)
        double log2w = (double)c;                              // + sum log2(sqrt(2 pi) sigma) = log2 w
        double lanes[2] = {1.0, 1.0};
        double tail = 1.0;
        const int paired = ORDER == 2 ? (dim & ~1) : dim;
        for (int d = 0; d < dim; d++) {
            const double s = (double)rec[d * 8 + jj * 2];
            const double m = (double)rec[d * 8 + jj * 2 + 1];
            const double sig = SQRT_HALF_LOG2E / s;
            const double t = xv(d) * s + m;
            const double b = ftz(-(t * t) * LN2);              // -d^2 / (2 sigma^2)
            const double ex = remez5_ftz(b);
            log2w += log2(SQRT_2_PI * sig);
            double p;
            if (ORDER == 2) p = ftz(ftz(ex * INV_SQRT_2_PI) / sig);
            else p = ftz(ex / (SQRT_2_PI * sig));
            if (d < paired) {
                const int which = ORDER == 2 ? (d & 1) : 0;
                lanes[which] = ftz(lanes[which] * p);
            } else {
                tail = p;
            }
        }
        double prob = ORDER == 2 ? ftz(lanes[0] * lanes[1]) : lanes[0];
        if (ORDER == 2 && (dim & 1)) prob = ftz(prob * tail);
        const double w = ftz(exp2(log2w));
        sum += ftz(w * prob);
    }
    sum = wave_sum_f64(sum);
    // safe_log, gmm.cc:34-38 (a NaN frame stays NaN: `sum > 0` and `sum <= 0` are both false for it)
    const float ll = sum > 0.0 ? (float)log(sum) : (sum <= 0.0 ? LSE_LN_1E_15 : (float)sum);
    if (lane == 0) {
        exact_out[(int64_t)blockIdx.x * frames_per_tile + j] = ll;
        if (frame_ll) frame_ll[(int64_t)pr.y * n_frames + row] = ll;
    }
}

// The tile's sum for gmm_finalize_kernel's utterance sum: float64, fixed order (a wave per pair).
__global__ __launch_bounds__(64)
void gmm_flush_tile_sum_kernel(const float *__restrict__ exact, const TileDesc *__restrict__ tiles,
                               const int2 *__restrict__ pairs, int frames_per_tile, double *__restrict__ tile_sum) {
    const int count = tiles[pairs[blockIdx.x].x].count;
    double mine = 0.0;
    for (int j = threadIdx.x; j < count; j += 64) mine += (double)exact[(int64_t)blockIdx.x * frames_per_tile + j];
    mine = wave_sum_f64(mine);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = mine;
}

struct FlushPatch {
    int utt, model;
    double delta;
};

__global__ void gmm_flush_patch_kernel(const FlushPatch *__restrict__ patches, int n, int n_models, double *__restrict__ sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sums[(int64_t)patches[i].utt * n_models + patches[i].model] += patches[i].delta;   // one patch per (utt, model)
}

// argmax of the patched utterances again: first maximum wins (gmmset.py:62-64), as gmm_finalize_kernel
__global__ __launch_bounds__(256)
void gmm_flush_argmax_kernel(const int *__restrict__ utts, int n_models, const double *__restrict__ sums, int *__restrict__ argmax) {
    const int u = utts[blockIdx.x];
    double best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int s = threadIdx.x; s < n_models; s += 256) {
        const double v = sums[(int64_t)u * n_models + s];
        if (v > best) {
            best = v;
            best_i = s;
        }
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
    if (threadIdx.x == 0) argmax[u] = si[0] != 0x7fffffff ? si[0] : -1;
}

struct FlushWorkspace {
    DevBuf<int2> sorted;
    DevBuf<float> exact;
    DevBuf<double> tile_sum;
    DevBuf<FlushPatch> patches;
    DevBuf<int> utts;
};

struct FlushHostStaging {          // page-locked: the small copies back do not queue behind a bulk upload's staging
    PinnedBuf<int2> list;
    PinnedBuf<double> tile_sum;
};

struct FlushStats {
    std::atomic<long> calls{0}, pairs{0}, frames{0};
};
FlushStats g_flush_stats;

}  // namespace

int &flush_order_option() {
    static int v = 2;
    return v;
}

void flush_stats(long *calls, long *pairs, long *frames) {
    if (calls) *calls = g_flush_stats.calls.load();
    if (pairs) *pairs = g_flush_stats.pairs.load();
    if (frames) *frames = g_flush_stats.frames.load();
}

// Re-evaluates every frame of the `count` noted (tile, model) pairs (device list `d_list`, any order; tiles of `tt`)
// with the reference's arithmetic and completes the device-resident results: sums[u][s] += the tiles' sums (added on
// the host in (tile, model) order: deterministic), the argmax of the utterances touched, the per-frame values.
// The device part + one host wait: the exact kernels run on the list AS THE DEVICE LEFT IT (the order in which gmm_finalize_kernel's
// atomics appended the pairs decides nothing: a pair's sum is its own), list and tile sums come back together, and the host sorts
// the pairs by (tile, model) before it adds them up -- the deterministic order of additions, one synchronisation instead of two.
// -> patches (utterance, model, sum to add), sorted, and the utterances they touch.
static void flush_evaluate(SRModelSet &set, SRBatch &feat, const TileTable &tt, const int2 *d_list, int count, float *d_frame_ll,
                           std::vector<FlushPatch> &patches, std::vector<int> &utts) {
    auto &fw = per_device<FlushWorkspace>();
    const int S = set.host.n_models;
    // the per-model record table of the vector layout (always resident: upload_model_set)
    if (!set.d_flush_models.p) {
        std::vector<FlushModel> fm(S);
        for (int s = 0; s < S; s++) {
            const int cb = set.host.model_chunk_begin[s], ce = set.host.model_chunk_begin[s + 1];
            fm[s].offset_f4 = set.host.chunks[cb].offset_f4;
            int n = 0;
            for (int c = cb; c < ce; c++) n += set.host.chunks[c].n_records;
            fm[s].n_records = n;
        }
        static_assert(sizeof(FlushModel) == 2 * sizeof(int), "two ints per model");
        set.d_flush_models.upload(reinterpret_cast<const int *>(fm.data()), (size_t)2 * S);
        sync_stream();
    }
    const size_t n_pairs = (size_t)count;
    fw.tile_sum.ensure(n_pairs);
    const int fpt = tt.frames_per_tile;
    // batches of pairs: the per-frame scratch stays below 64 MiB
    const size_t per_batch = std::max<size_t>(1, ((size_t)64 << 20) / ((size_t)fpt * sizeof(float)));
    fw.exact.ensure(std::min(per_batch, n_pairs) * (size_t)fpt);
    for (size_t base = 0; base < n_pairs; base += per_batch) {
        const size_t n = std::min(per_batch, n_pairs - base);
        const dim3 grid((unsigned)n, (unsigned)((fpt + 3) / 4));
#define SR_FLUSH_LAUNCH(ORDER, XG)                                                                                        \
        hipLaunchKernelGGL((gmm_flush_exact_kernel<ORDER, XG>), grid, dim3(256), (XG) ? 0 : (size_t)4 * feat.dim * sizeof(double),   \
                           ctx().stream, feat.data.p, set.d_center0.p,                                                     \
                           set.d_params.p, reinterpret_cast<const FlushModel *>(set.d_flush_models.p), tt.d_tiles.p,       \
                           d_list + base, feat.dim, set.host.dp, fpt, feat.n_rows,                                         \
                           (float)(-708.396418532264 + set.host.flush_band), fw.exact.p, d_frame_ll)
        if (feat.dim > FLUSH_LDS_MAX_DIM) {
            if (flush_order_option() == 1) SR_FLUSH_LAUNCH(1, true); else SR_FLUSH_LAUNCH(2, true);
        } else {
            if (flush_order_option() == 1) SR_FLUSH_LAUNCH(1, false); else SR_FLUSH_LAUNCH(2, false);
        }
#undef SR_FLUSH_LAUNCH
        hipLaunchKernelGGL(gmm_flush_tile_sum_kernel, dim3((unsigned)n), dim3(64), 0, ctx().stream, fw.exact.p, tt.d_tiles.p,
                           d_list + base, fpt, fw.tile_sum.p + base);
        SR_HIP(hipGetLastError());
    }
    auto &hs = per_device<FlushHostStaging>();
    hs.list.ensure(n_pairs);
    hs.tile_sum.ensure(n_pairs);
    SR_HIP(hipMemcpyAsync(hs.list.p, d_list, n_pairs * sizeof(int2), hipMemcpyDeviceToHost, ctx().stream));
    SR_HIP(hipMemcpyAsync(hs.tile_sum.p, fw.tile_sum.p, n_pairs * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
    sync_stream();
    // (utterance, model) patches: pairs in (tile, model) order -- tiles are in utterance order --, one patch per (utterance, model)
    long frames = 0;
    std::vector<std::pair<int64_t, double>> acc;     // key = (tile * S + model), then (utt * S + model)
    acc.reserve(n_pairs);
    for (size_t i = 0; i < n_pairs; i++) acc.emplace_back((int64_t)hs.list.p[i].x * S + hs.list.p[i].y, hs.tile_sum.p[i]);
    std::sort(acc.begin(), acc.end(), [](const auto &a, const auto &b) { return a.first < b.first; });   // (a pair occurs once: no ties)
    for (auto &e : acc) {
        const TileDesc &td = tt.h_tiles[(size_t)(e.first / S)];
        frames += td.count;
        e.first = (int64_t)td.utt * S + (e.first % S);
    }
    std::stable_sort(acc.begin(), acc.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    patches.clear();
    utts.clear();
    for (size_t i = 0; i < acc.size();) {
        size_t j = i;
        double d = 0.0;
        for (; j < acc.size() && acc[j].first == acc[i].first; j++) d += acc[j].second;
        patches.push_back(FlushPatch{(int)(acc[i].first / S), (int)(acc[i].first % S), d});
        if (utts.empty() || utts.back() != patches.back().utt) utts.push_back(patches.back().utt);
        i = j;
    }
    g_flush_stats.calls++;
    g_flush_stats.pairs += (long)n_pairs;
    g_flush_stats.frames += frames;
}

void flush_resolve(SRModelSet &set, SRBatch &feat, const TileTable &tt, const int2 *d_list, int count, double *d_sums,
                   int *d_argmax, float *d_frame_ll) {
    if (count <= 0) return;
    auto &fw = per_device<FlushWorkspace>();
    const int S = set.host.n_models;
    std::vector<FlushPatch> patches;
    std::vector<int> utts;
    flush_evaluate(set, feat, tt, d_list, count, d_frame_ll, patches, utts);
    fw.patches.upload(patches.data(), patches.size());
    fw.utts.upload(utts.data(), utts.size());
    hipLaunchKernelGGL(gmm_flush_patch_kernel, dim3((unsigned)((patches.size() + 255) / 256)), dim3(256), 0, ctx().stream,
                       fw.patches.p, (int)patches.size(), S, d_sums);
    hipLaunchKernelGGL(gmm_flush_argmax_kernel, dim3((unsigned)utts.size()), dim3(256), 0, ctx().stream, fw.utts.p, S, d_sums,
                       d_argmax);
    SR_HIP(hipGetLastError());
    sync_stream();       // the uploads above read host vectors that die with this frame
}

// The same, completing HOST copies of the results (sr_multi_predict_pcm's pieces: their sums and argmax are already in
// page-locked host memory when the count is known): nothing goes back to the device, one host wait in all.
void flush_resolve_host(SRModelSet &set, SRBatch &feat, const TileTable &tt, const int2 *d_list, int count, double *h_sums,
                        int *h_argmax) {
    if (count <= 0) return;
    const int S = set.host.n_models;
    std::vector<FlushPatch> patches;
    std::vector<int> utts;
    flush_evaluate(set, feat, tt, d_list, count, nullptr, patches, utts);
    for (const FlushPatch &p : patches) h_sums[(size_t)p.utt * S + p.model] += p.delta;
    for (int u : utts) {                     // first maximum wins (gmm_flush_argmax_kernel, gmmset.py:62-64)
        double best = -INFINITY;
        int best_i = -1;
        for (int s = 0; s < S; s++) {
            const double v = h_sums[(size_t)u * S + s];
            if (v > best) {
                best = v;
                best_i = s;
            }
        }
        h_argmax[u] = best_i;
    }
}

}  // namespace sr
