// kmeans_init.hip -- the reference's initialisation of a GMM before EM, decision for decision:
// GMMTrainerBaseline::init_gaussians (src/gmm/src/gmm.cc:306-361): data variance, then either K random
// frames (init_with_kmeans = 0, the reference's default) or k-means|| (KMeansIISolver::cluster,
// kmeansII.cc:82-171: oversampling rounds -> weighted k-means++ on the candidates, kmeans++.cc:161-212 ->
// weighted Lloyd, kmeans.cc:249-342 -> Lloyd on the full data, kmeans.cc:150-246).
//
// What makes the reference reproducible is its random numbers: every draw comes from libc rand() or
// from a std::default_random_engine seeded by rand() (random.hh:17-56), and rand() starts from its default
// seed in a fresh process.  The same draws are made here, in the same order, from the same generators
// (libstdc++'s engine and distribution; glibc's rand() -- restated below, because the process-wide one
// cannot be used: the HIP runtime draws from it while it initialises, scripts/debug/randcheck.py), so a
// model trained from scratch through the legacy train_model symbol comes out as the reference's does in a
// process where it is the only user of rand() (tests/golden/make_init_golden.py).  A caller who passes
// a seed (sr_train_f32) gets the same algorithm on a stream of its own.
//
// Arithmetic: the reference works in float64 and sums in a fixed order (per worker block of
// ceil(n / concurrency) points, then over the blocks); sums that feed a decision are formed in that
// order here as well.  The O(n K D) part -- nearest centre and squared distance of every point -- runs on
// the device in float64, dimension by dimension with separate multiply and add as the reference's SSE2
// build does (no FMA); the per-cluster sums of a Lloyd step are added on the host in the reference's
// order (O(n D) per step, one host thread per worker block).
#include "score.hpp"

#include "../../include/pygmm_hip.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <random>
#include <thread>

namespace sr {

namespace {

// random.hh:17-56
struct RefRandom {
    std::default_random_engine generator;
    std::uniform_real_distribution<double> real_distribution;     // [0, 1)
    explicit RefRandom(long long seed) { generator.seed(seed); }
    double rand_real() { return real_distribution(generator); }
    int rand_int(int max_val = std::numeric_limits<int>::max()) { return (int)(rand_real() * max_val); }
};

// glibc's rand() (random_r.c, TYPE_3: the additive feedback generator x^31 + x^3 + 1 over 31-bit outputs,
// state filled from the seed by the Lehmer generator 16807 mod 2^31 - 1, first 310 outputs discarded).
// tests/test_abi_cpu.py checks it against the C library's own rand().
struct GlibcRand {
    int32_t r[34];
    int f = 3, b = 0;           // front / rear indices into r[3..33]
    explicit GlibcRand(unsigned seed = 1) {
        int32_t *state = r + 3;     // 31 words (r[0..2] unused; keeps the textbook indices readable)
        if (seed == 0) seed = 1;
        state[0] = (int32_t)seed;
        for (int i = 1; i < 31; i++) {
            const long hi = state[i - 1] / 127773, lo = state[i - 1] % 127773;
            long word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            state[i] = (int32_t)word;
        }
        f = 3;
        b = 0;
        for (int i = 0; i < 310; i++) (void)next();
    }
    int next() {
        int32_t *state = r + 3;
        const uint32_t val = (uint32_t)state[f] + (uint32_t)state[b];
        state[f] = (int32_t)val;
        const int result = (int)(val >> 1);
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
        return result;
    }
};
std::mutex g_rand_mutex;
GlibcRand g_reference_rand;       // what rand() would return in a process where the reference library is its only user

// The stream the reference draws from -- the restated libc one (library-wide, as the reference's is process-wide) --
// or, when the caller seeds, a generator of the caller's own.
struct RandStream {
    bool shared;
    GlibcRand own;
    explicit RandStream(long seed) : shared(seed < 0), own(seed < 0 ? 1u : (unsigned)seed + 1u) {}
    int operator()() {
        if (!shared) return own.next();
        std::lock_guard<std::mutex> lock(g_rand_mutex);
        return g_reference_rand.next();
    }
};

constexpr int KM_CH = 32;          // centres per LDS chunk
constexpr int KM_MAX_ITER = 200;   // kmeans.cc:172, :272

// Nearest centre among centres [c_begin, c_end): dist[i] / belong[i] are updated when a centre is STRICTLY
// closer (kmeansII.cc:59-72; with dist preset to DBL_MAX it is the full search of kmeans.cc:87-99).
__global__ __launch_bounds__(256)
void kmeans_assign_kernel(const float *__restrict__ X, long n, int dim, const double *__restrict__ C,
                          int c_begin, int c_end, double *__restrict__ dist, int *__restrict__ belong, int reset) {
    extern __shared__ double cs[];               // [KM_CH][dim]
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const float *x = X + (valid ? i : 0) * dim;
    double best = reset ? 1.7976931348623157e308 : valid ? dist[i] : 0.0;     // reset: a fresh search (DBL_MAX, no centre yet)
    int best_j = reset ? -1 : valid ? belong[i] : 0;
    for (int c0 = c_begin; c0 < c_end; c0 += KM_CH) {
        const int nc = min(KM_CH, c_end - c0);
        __syncthreads();
        for (int e = threadIdx.x; e < nc * dim; e += 256) cs[e] = C[(size_t)c0 * dim + e];
        __syncthreads();
        double acc[KM_CH];
#pragma unroll
        for (int j = 0; j < KM_CH; j++) acc[j] = 0.0;
        for (int d = 0; d < dim; d++) {
            const double xv = (double)x[d];
#pragma unroll
            for (int j = 0; j < KM_CH; j++) {
                if (j < nc) {
                    const double delta = __dsub_rn(xv, cs[j * dim + d]);
                    acc[j] = __dadd_rn(acc[j], __dmul_rn(delta, delta));      // mul, then add: the reference has no FMA
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KM_CH; j++)
            if (j < nc && acc[j] < best) {
                best = acc[j];
                best_j = c0 + j;
            }
    }
    if (valid) {
        dist[i] = best;
        belong[i] = best_j;
    }
}

struct KmWorkspace {
    DevBuf<float> X;
    DevBuf<double> C, dist;
    DevBuf<int> belong;
};
KmWorkspace &kws() { return per_device<KmWorkspace>(); }

void device_assign(long n, int dim, const std::vector<double> &C, int c_begin, int c_end,
                   std::vector<double> &dist, std::vector<int> &belong, bool reset) {
    auto &w = kws();
    w.C.upload(C.data(), (size_t)c_end * dim);
    if (reset) {
        w.dist.ensure((size_t)n);
        w.belong.ensure((size_t)n);
    } else {
        w.dist.upload(dist.data(), (size_t)n);
        w.belong.upload(belong.data(), (size_t)n);
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(kmeans_assign_kernel, dim3(grid), dim3(256), (size_t)KM_CH * dim * sizeof(double), ctx().stream,
                       w.X.p, n, dim, w.C.p, c_begin, c_end, w.dist.p, w.belong.p, reset ? 1 : 0);
    SR_HIP(hipGetLastError());
    w.dist.download(dist.data(), (size_t)n);
    w.belong.download(belong.data(), (size_t)n);
    sync_stream();
}

// sum of v[0..n) as the reference's workers form it: sequentially inside blocks of ceil(n / concurrency),
// then over the blocks (kmeansII.cc:103-129, kmeans.cc:166-187)
double blocked_sum(const std::vector<double> &v, long n, int concurrency) {
    const long block = (long)std::ceil((double)n / concurrency);
    double total = 0;
    for (long b = 0; b < n; b += block) {
        double s = 0;
        const long e = std::min(n, b + block);
        for (long i = b; i < e; i++) s += v[i];
        total += s;
    }
    return total;
}

// Lloyd on the full data (kmeans.cc:150-246).  centroids: [K][dim] in/out.
void lloyd_full(const float *X, long n, int dim, int K, int concurrency, std::vector<double> &centroids, int verbosity) {
    std::vector<double> best_centroids, dist((size_t)n);
    std::vector<int> belong((size_t)n);
    double best = std::numeric_limits<double>::max(), last = std::numeric_limits<double>::max();
    const long block = (long)std::ceil((double)n / concurrency);
    const int n_blocks = (int)((n + block - 1) / block);
    std::vector<std::vector<double>> buf((size_t)n_blocks, std::vector<double>((size_t)K * dim));
    std::vector<std::vector<int>> cnt((size_t)n_blocks, std::vector<int>((size_t)K));
    for (int iter = 0; iter < KM_MAX_ITER; iter++) {
        device_assign(n, dim, centroids, 0, K, dist, belong, true);
        // per worker block: sum of the distances and of the members, in point order (calc_belonging, kmeans.cc:72-107)
        std::vector<double> block_sum((size_t)n_blocks, 0.0);
        auto work = [&](int b) {
            std::fill(buf[b].begin(), buf[b].end(), 0.0);
            std::fill(cnt[b].begin(), cnt[b].end(), 0);
            double s = 0;
            const long e = std::min(n, (long)(b + 1) * block);
            for (long i = (long)b * block; i < e; i++) {
                const int j = belong[i];
                if (j >= 0) {                    // a point no centre is comparable to (NaN centres) has none; the reference indexes [-1] there
                    cnt[b][j] += 1;
                    double *c = buf[b].data() + (size_t)j * dim;
                    const float *x = X + (size_t)i * dim;
                    for (int d = 0; d < dim; d++) c[d] += (double)x[d];
                }
                s += dist[i];
            }
            block_sum[b] = s;
        };
        {
            // the blocks are independent; a few host threads share them (the reference runs one task per block)
            const int n_threads = std::max(1, std::min({n_blocks, (int)std::thread::hardware_concurrency(), 16}));
            auto run = [&](int t) { for (int b = t; b < n_blocks; b += n_threads) work(b); };
            std::vector<std::thread> th;
            if (n > 20000)
                for (int t = 1; t < n_threads; t++) th.emplace_back(run, t);
            else
                for (int t = 1; t < n_threads; t++) run(t);          // small inputs: not worth a thread
            run(0);
            for (auto &t : th) t.join();
        }
        double sum = 0;
        for (int b = 0; b < n_blocks; b++) sum += block_sum[b];
        if (sum < best) {
            best = sum;
            best_centroids = centroids;
        }
        if (verbosity >= 2) printf("k-means iteration %3d: %f\n", iter, sum);
        if (std::fabs(last - sum) < 1e-6) break;
        if (sum > best * 1.5) break;                          // terminate_cost_factor
        std::vector<double> size((size_t)K, 0.0);
        for (int b = 0; b < n_blocks; b++)
            for (int k = 0; k < K; k++) size[k] += cnt[b][k];
        std::fill(centroids.begin(), centroids.end(), 0.0);
        for (int b = 0; b < n_blocks; b++)
            for (size_t e = 0; e < (size_t)K * dim; e++) centroids[e] += buf[b][e];
        for (int k = 0; k < K; k++)
            for (int d = 0; d < dim; d++) centroids[(size_t)k * dim + d] /= size[k];      // an empty cluster divides by 0, as the reference
        last = sum;
    }
    centroids = best_centroids;
}

// squared distance over the coordinates a sparse instance keeps (|x| >= 1e-15, kmeans++.cc:43-51, :53-60)
inline double sparse_distsqr(const double *x, const double *c, int dim) {
    double dist = 0;
    for (int d = 0; d < dim; d++) {
        if (std::fabs(x[d]) < 1e-15) continue;
        const double delta = x[d] - c[d];
        dist += delta * delta;
    }
    return dist;
}

// Weighted Lloyd on the candidate set (kmeans.cc:249-342).
void lloyd_weighted(const std::vector<double> &P, const std::vector<double> &weight, int np, int dim, int K,
                    int concurrency, std::vector<double> &centroids) {
    std::vector<double> best_centroids;
    double best = std::numeric_limits<double>::max(), last = std::numeric_limits<double>::max();
    const int block = (int)std::ceil((double)np / concurrency);
    const int n_blocks = (np + block - 1) / block;
    std::vector<std::vector<double>> buf((size_t)n_blocks, std::vector<double>((size_t)K * dim)), csz((size_t)n_blocks, std::vector<double>((size_t)K));
    for (int iter = 0; iter < KM_MAX_ITER; iter++) {
        std::vector<double> block_sum((size_t)n_blocks, 0.0);
        auto work = [&](int b) {
            std::fill(buf[b].begin(), buf[b].end(), 0.0);
            std::fill(csz[b].begin(), csz[b].end(), 0.0);
            double s = 0;
            for (int i = b * block; i < std::min(np, (b + 1) * block); i++) {
                const double *x = P.data() + (size_t)i * dim;
                double mind = std::numeric_limits<double>::max();
                int id = -1;
                for (int j = 0; j < K; j++) {
                    const double dq = sparse_distsqr(x, centroids.data() + (size_t)j * dim, dim) * weight[i];
                    if (dq < mind) {
                        mind = dq;
                        id = j;
                    }
                }
                if (id >= 0) {
                    csz[b][id] += weight[i];
                    double *c = buf[b].data() + (size_t)id * dim;
                    for (int d = 0; d < dim; d++)
                        if (std::fabs(x[d]) >= 1e-15) c[d] += x[d] * weight[i];
                }
                s += mind;
            }
            block_sum[b] = s;
        };
        {
            const int n_threads = std::max(1, std::min({n_blocks, (int)std::thread::hardware_concurrency(), 16}));
            auto run = [&](int t) { for (int b = t; b < n_blocks; b += n_threads) work(b); };
            std::vector<std::thread> th;
            if ((size_t)np * K * dim > ((size_t)1 << 22))
                for (int t = 1; t < n_threads; t++) th.emplace_back(run, t);
            else
                for (int t = 1; t < n_threads; t++) run(t);
            run(0);
            for (auto &t : th) t.join();
        }
        double sum = 0;
        for (int b = 0; b < n_blocks; b++) sum += block_sum[b];
        if (sum < best) {
            best = sum;
            best_centroids = centroids;
        }
        if (std::fabs(last - sum) < 1e-6) break;
        std::vector<int> size((size_t)K, 0);                  // an int accumulator, as kmeans.cc:303-308
        for (int b = 0; b < n_blocks; b++)
            for (int k = 0; k < K; k++) size[k] += csz[b][k];
        std::fill(centroids.begin(), centroids.end(), 0.0);
        for (int b = 0; b < n_blocks; b++)
            for (size_t e = 0; e < (size_t)K * dim; e++) centroids[e] += buf[b][e];
        for (int k = 0; k < K; k++)
            for (int d = 0; d < dim; d++) centroids[(size_t)k * dim + d] /= size[k];
        last = sum;
    }
    centroids = best_centroids;
}

// KMeansIISolver::cluster (kmeansII.cc:82-171) with its defaults oversampling_factor = size_factor = 2.
std::vector<double> kmeans_parallel_init(const float *X, long n, int dim, int K, int concurrency, RandStream &rs, int verbosity) {
    const double oversampling_factor = 2.0, size_factor = 2.0;
    RefRandom solver_random(rs());                            // KMeansIISolver::random (kmeansII.hh:41)
    std::vector<double> cand;                                 // candidate centres, [count][dim]
    auto push_point = [&](long i) {
        for (int d = 0; d < dim; d++) cand.push_back((double)X[(size_t)i * dim + d]);
    };
    push_point((long)(rs() % n));
    std::vector<double> dist((size_t)n, std::numeric_limits<double>::max());
    std::vector<int> belong((size_t)n, 0);                    // vector<int> belong(n): zero-initialised (kmeansII.cc:97)
    long last_size = 0;
    for (int iter = 0;; iter++) {
        const long size = (long)(cand.size() / dim);
        device_assign(n, dim, cand, (int)last_size, (int)size, dist, belong, false);
        if ((double)size > size_factor * K) break;
        const double distsqr_sum = blocked_sum(dist, n, concurrency);
        last_size = size;
        for (long i = 0; i < n; i++) {
            const double random_weight = rs() / (double)RAND_MAX * distsqr_sum;
            if (random_weight < dist[i] * oversampling_factor * K) push_point(i);
        }
        const long added = (long)(cand.size() / dim) - last_size;
        if (verbosity >= 2) printf("k-means|| round %d: %f, new %ld, all %ld\n", iter, distsqr_sum, added, last_size + added);
        if (added == 0) break;
    }
    while ((double)(cand.size() / dim) <= size_factor * K) push_point(solver_random.rand_int((int)n));
    const int np = (int)(cand.size() / dim);
    std::vector<double> weight((size_t)np, 0.0);
    for (long i = 0; i < n; i++) weight[belong[i]] += 1.0;

    // weighted k-means++ over the candidates (KMeansppSolver::cluster_weighted, kmeans++.cc:161-212)
    RefRandom pp_random(rs());                                // KMeansppSolver::random (kmeans++.hh:32)
    std::vector<double> centroids((size_t)K * dim, 0.0);
    auto copy_sparse = [&](int from, int k) {                 // Vector2Instance drops |x| < 1e-15, Instance2Vector leaves those 0
        for (int d = 0; d < dim; d++) {
            const double v = cand[(size_t)from * dim + d];
            centroids[(size_t)k * dim + d] = std::fabs(v) < 1e-15 ? 0.0 : v;
        }
    };
    copy_sparse(pp_random.rand_int() % np, 0);
    std::vector<double> pdist((size_t)np, std::numeric_limits<double>::max());
    for (int k = 1; k < K; k++) {
        const double *c = centroids.data() + (size_t)(k - 1) * dim;
        for (int i = 0; i < np; i++)
            pdist[i] = std::min(pdist[i], sparse_distsqr(cand.data() + (size_t)i * dim, c, dim) * weight[i]);
        const double distsqr_sum = blocked_sum(pdist, np, concurrency);
        double random_weight = pp_random.rand_int() / (double)RAND_MAX * distsqr_sum;
        for (int i = 0; i < np; i++) {
            random_weight -= pdist[i];
            if (random_weight <= 0) {
                copy_sparse(i, k);
                break;
            }
        }
    }
    lloyd_weighted(cand, weight, np, dim, K, concurrency, centroids);
    lloyd_full(X, n, dim, K, concurrency, centroids, verbosity);
    return centroids;
}

}  // namespace

// The reference library's other draws from rand(), made by the legacy entry points so that the library's stream
// stays in step with the reference's: one per Gaussian that GMM::load constructs (gmm.cc:671-676), one for the
// trainer plus one per Gaussian in train_model_from_ubm (gmmubm.cc:29-38).
void burn_reference_rand(int count) {
    std::lock_guard<std::mutex> lock(g_rand_mutex);
    for (int i = 0; i < count; i++) (void)g_reference_rand.next();
}
// the first `count` values of a fresh generator (test hook: compared with the C library's rand())
void reference_rand_sample(int *out, int count) {
    GlibcRand g;
    for (int i = 0; i < count; i++) out[i] = g.next();
}

// init_gaussians (gmm.cc:306-361).  seed < 0: the reference's stream (the library-wide restated rand()).
void init_gmm_like_reference(GMM &g, const float *X, long n, int dim, const Parameter &param, long seed) {
    const int K = g.nr_mixtures;
    const int concurrency = std::max(1, param.concurrency);
    g.dim = dim;
    RandStream rs(seed);
    RefRandom trainer_random(rs());                           // GMMTrainerBaseline::random, seeded when the trainer is built (pygmm.cc:65)
    for (int k = 0; k < K; k++) (void)rs();                   // every `new Gaussian` seeds a Random of its own (gmm.hh:44, gmm.cc:327-329)
    // data variance, unbiased, one sigma vector shared by all mixtures (gmm.cc:309-325, :352-354)
    std::vector<double> mean(dim, 0.0), var(dim, 0.0);
    for (long i = 0; i < n; i++)
        for (int d = 0; d < dim; d++) mean[d] += X[(size_t)i * dim + d];
    for (int d = 0; d < dim; d++) mean[d] /= (double)n;
    for (long i = 0; i < n; i++)
        for (int d = 0; d < dim; d++) {
            const double v = X[(size_t)i * dim + d] - mean[d];
            var[d] += v * v;
        }
    g.sigma.assign((size_t)K * dim, 0.0);
    for (int k = 0; k < K; k++)
        for (int d = 0; d < dim; d++) g.sigma[(size_t)k * dim + d] = std::sqrt(var[d] * (1.0 / (double)(n - 1)));
    for (double s : g.sigma)
        if (!(s > 0)) fail("training data has zero variance in some dimension");
    g.mean.assign((size_t)K * dim, 0.0);
    if (param.init_with_kmeans > 0) {
        ensure_device();
        kws().X.upload(X, (size_t)n * dim);
        const std::vector<double> c = kmeans_parallel_init(X, n, dim, K, concurrency, rs, param.verbosity);
        for (size_t e = 0; e < (size_t)K * dim; e++) g.mean[e] = c[e];
        for (double v : g.mean)
            if (!std::isfinite(v)) fail("k-means initialisation left an empty cluster (the reference divides by zero here as well)");
    } else {
        for (int k = 0; k < K; k++) {                         // gmm.cc:346-349
            const long pick = trainer_random.rand_int((int)n);
            for (int d = 0; d < dim; d++) g.mean[(size_t)k * dim + d] = X[(size_t)pick * dim + d];
        }
    }
    g.weights.assign(K, 1.0 / K);                             // gmm.cc:356-360
    g.drop_single();
}

}  // namespace sr
