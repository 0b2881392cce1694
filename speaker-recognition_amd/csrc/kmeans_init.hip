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
// build does (no FMA).  The per-cluster sums of a Lloyd step on the full data are formed on the device too
// (round 3), in the reference's order: a stable radix sort of the points by (worker block, cluster) puts every
// (block, cluster)'s members side by side in point order; one thread per (cluster, dimension) then adds each
// (block, cluster)'s members side by side in point order; one thread per (block, cluster, dimension) adds them one
// after the other, one per (cluster, dimension) the blocks' sums in block order -- the additions calc_belonging and
// Lloyds_iteration make (kmeans.cc:72-107, :189-212), bit for bit, at device bandwidth instead of 9 s of host
// loops in front of a 2048-mixture UBM.
// The full data keeps every coordinate (gmm.cc:288-294 dense2sparse; only the CANDIDATE centres of the weighted
// k-means++ / weighted Lloyd stages go through Vector2Instance, which drops |x| < 1e-15, kmeans++.cc:42-51), so
// the distance here is the dense one.
#include "score.hpp"

#include "../../include/pygmm_hip.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>     // the device radix sort of the Lloyd step (rocPRIM: the native library; hipCUB wraps it in CUB's API)

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <new>
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

constexpr int KM_CH = 16;          // centres per LDS chunk
constexpr int KM_PP = 2;           // points per lane: a centre coordinate read from LDS serves both
constexpr int KM_MAX_ITER = 200;   // kmeans.cc:172, :272
typedef double km_d2 __attribute__((ext_vector_type(2)));

// Nearest centre among centres [c_begin, c_end): dist[i] / belong[i] are updated when a centre is STRICTLY
// closer (kmeansII.cc:59-72; with dist preset to DBL_MAX it is the full search of kmeans.cc:87-99).
// float64, dimension by dimension, subtract - multiply - add as the reference's SSE2 build (no FMA): the distances
// and therefore every decision taken on them are the reference's, bit for bit.  A lane keeps two points in registers
// (fp32, widened per use) and 2 x 16 running sums; a chunk of 16 centres sits in LDS transposed ([d][centre]), so one
// 16-byte broadcast read feeds two centres x two points x three operations: the kernel is bound by the fp64 vector rate
// (round 2's one point per lane and one 8-byte read per three operations was bound by LDS issue: 21 ms per pass of
// 1 M points x 2048 centres x 39 dims).
template <int DIM_MAX>
__global__ __launch_bounds__(256, 2)
void kmeans_assign_kernel(const float *__restrict__ X, long n, int dim, const double *__restrict__ C,
                          int c_begin, int c_end, double *__restrict__ dist, int *__restrict__ belong, int reset,
                          const int *__restrict__ list, const int *__restrict__ n_list) {
    // (list != nullptr: only the points list[0 .. *n_list), the ones the fast pass below could not decide)
    const long n_work = list ? (long)*n_list : n;
    __shared__ km_d2 cs[DIM_MAX * KM_CH / 2];    // [d][KM_CH]
    double *csd = reinterpret_cast<double *>(cs);
    constexpr bool XREG = DIM_MAX <= 40;         // wider rows are re-read (L1 / L2) instead of held: 2 x 128 registers would spill
    float x[KM_PP][XREG ? DIM_MAX : 1];
    const float *xsrc[KM_PP];
    long idx[KM_PP];
    bool valid[KM_PP];
    double best[KM_PP];
    int best_j[KM_PP];
#pragma unroll
    for (int p = 0; p < KM_PP; p++) {
        idx[p] = ((long)blockIdx.x * KM_PP + p) * 256 + threadIdx.x;
        valid[p] = idx[p] < n_work;
        if (list) idx[p] = valid[p] ? (long)list[idx[p]] : 0;
        const float *src = X + (valid[p] ? idx[p] : 0) * dim;
        xsrc[p] = src;
        if constexpr (XREG) {
#pragma unroll
            for (int d = 0; d < DIM_MAX; d++) x[p][d] = d < dim ? src[d] : 0.f;
        }
        best[p] = reset ? 1.7976931348623157e308 : valid[p] ? dist[idx[p]] : 0.0;     // reset: a fresh search (DBL_MAX, no centre yet)
        best_j[p] = reset ? -1 : valid[p] ? belong[idx[p]] : 0;
    }
    if ((long)blockIdx.x * KM_PP * 256 >= n_work) return;         // (list mode: the grid is sized for the worst case)
    for (int c0 = c_begin; c0 < c_end; c0 += KM_CH) {
        const int nc = min(KM_CH, c_end - c0);
        double acc[KM_PP][KM_CH];
#pragma unroll
        for (int p = 0; p < KM_PP; p++)
#pragma unroll
            for (int j = 0; j < KM_CH; j++) acc[p][j] = 0.0;
        // (rows wider than DIM_MAX -- the reference has no limit -- take the chunk's centres DIM_MAX dimensions at a time; the
        // running sums go on in dimension order, so the distances are still the reference's bit for bit)
        for (int d0 = 0; d0 < dim; d0 += DIM_MAX) {
        const int dn = min(DIM_MAX, dim - d0);
        __syncthreads();
        for (int e = threadIdx.x; e < KM_CH * dn; e += 256) {
            const int j = e / dn, d = e - j * dn;
            csd[d * KM_CH + j] = j < nc ? C[(size_t)(c0 + j) * dim + d0 + d] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < DIM_MAX; d++) {
            if (d < dn) {
                double xv[KM_PP];
#pragma unroll
                for (int p = 0; p < KM_PP; p++) xv[p] = XREG ? (double)x[p][XREG ? d : 0] : (double)xsrc[p][d0 + d];
#pragma unroll
                for (int j2 = 0; j2 < KM_CH / 2; j2++) {
                    const km_d2 c = cs[d * (KM_CH / 2) + j2];
#pragma unroll
                    for (int p = 0; p < KM_PP; p++) {
                        const double d0 = __dsub_rn(xv[p], c.x), d1 = __dsub_rn(xv[p], c.y);
                        acc[p][2 * j2] = __dadd_rn(acc[p][2 * j2], __dmul_rn(d0, d0));      // mul, then add: the reference has no FMA
                        acc[p][2 * j2 + 1] = __dadd_rn(acc[p][2 * j2 + 1], __dmul_rn(d1, d1));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // (keeps the next dimensions' LDS reads from being hoisted: registers)
            }
        }
        }
#pragma unroll
        for (int p = 0; p < KM_PP; p++)
#pragma unroll
            for (int j = 0; j < KM_CH; j++)
                if (j < nc && acc[p][j] < best[p]) {
                    best[p] = acc[p][j];
                    best_j[p] = c0 + j;
                }
    }
#pragma unroll
    for (int p = 0; p < KM_PP; p++)
        if (valid[p]) {
            dist[idx[p]] = best[p];
            belong[idx[p]] = best_j[p];
        }
}

// ---- the full search, fast (round 3) ----
// 99 Lloyd steps of the exact pass above (10 ms each at K = 2048 on 1 M points: 3 float64 operations per point, centre and
// dimension) were three quarters of the k-means|| start.  The DECISION only needs the exact arithmetic where two centres are
// nearly equally close: ||x - c||^2 - ||x||^2 = ||c||^2 - 2 x.c is ONE fused multiply-add per dimension, and its error is
// bounded by (D + 2) 2^-53 x 2 (||x||^2 + ||c||^2) < 1e-14 (||x||^2 + max ||c||^2).  A point whose best and second-best values
// are further apart than tau = 2^-40 (||x||^2 + max ||c||^2) -- ~50 x that bound on each side, and far above the rounding of
// the reference's own sums -- has the same nearest centre in the reference's arithmetic; its distance is then formed the
// reference's way (subtract, multiply, add, dimension by dimension) for that one centre.  The others (exact ties between
// duplicate centres, the odd near-tie) go on a list and through the exact pass: same belong[] and dist[], bit for bit
// (test_kmeans_fast_assign_equals_the_exact_pass), 4 ms instead of 10.
// chunk records for the fast search: 16 centres per record, [d][16] doubles of -2 c, then the 16 ||c||^2; padded to whole KiB so
// that a record goes into LDS as 1 KiB LDS-DMA pieces
__host__ __device__ constexpr int km_rec_doubles(int dim_max) { return (dim_max * KM_CH + KM_CH + 127) / 128 * 128; }

__global__ __launch_bounds__(256)
void kmeans_centre_prep_kernel(const double *__restrict__ C, int K, int dim, int dim_max, double *__restrict__ rec,
                               double *__restrict__ cn_max_bits) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int k_pad = (K + KM_CH - 1) / KM_CH * KM_CH;
    if (j >= k_pad) return;
    double *r = rec + (size_t)(j / KM_CH) * km_rec_doubles(dim_max);
    const int col = j % KM_CH;
    if (j >= K) {                                       // padding of the last record: never wins, never the runner-up that matters
        for (int d = 0; d < dim_max; d++) r[d * KM_CH + col] = 0.0;
        r[dim_max * KM_CH + col] = 1.7976931348623157e308;
        return;
    }
    double s = 0.0;
    for (int d = 0; d < dim; d++) {
        const double c = C[(size_t)j * dim + d];
        s = fma(c, c, s);
    }
    // a centre with a NaN or infinite coordinate (an empty cluster: the reference divides by zero, kmeans.cc:229-233) is never
    // STRICTLY closer than anything in the exact pass; here it gets the value 1e300 for every point, so that the search below
    // can use plain min / max (no NaN handling) -- and if nothing else exists, the exact pass decides
    const bool finite = s == s && s < 1.0e300;
    for (int d = 0; d < dim_max; d++) r[d * KM_CH + col] = (finite && d < dim) ? -2.0 * C[(size_t)j * dim + d] : 0.0;
    r[dim_max * KM_CH + col] = finite ? s : 1.0e300;
    if (finite)
        atomicMax(reinterpret_cast<unsigned long long *>(cn_max_bits), (unsigned long long)__double_as_longlong(s));   // s >= 0: bit order = value order
}

template <int DIM_MAX>
__global__ __launch_bounds__(256, 2)
void kmeans_assign_fast_kernel(const float *__restrict__ X, long n, int dim, const double *__restrict__ C, const double *__restrict__ rec,
                               const double *__restrict__ cn_max, int K, double *__restrict__ dist,
                               int *__restrict__ belong, int *__restrict__ list, int *__restrict__ n_list) {
    constexpr int REC = km_rec_doubles(DIM_MAX);
    constexpr int PIECES = REC / 128;                       // 1 KiB wave-instructions per record
    __shared__ double buf_a[REC];                           // (two arrays, not one: hipcc then knows a read of one cannot alias the
    __shared__ double buf_b[REC];                           //  LDS-DMA in flight into the other, and does not drain vmcnt in front of it)
    static_assert(DIM_MAX <= 40, "rows held in registers");
    float x[KM_PP][DIM_MAX];
    long idx[KM_PP];
    bool valid[KM_PP];
    double best[KM_PP], second[KM_PP];
    int best_j[KM_PP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int p = 0; p < KM_PP; p++) {
        idx[p] = ((long)blockIdx.x * KM_PP + p) * 256 + threadIdx.x;
        valid[p] = idx[p] < n;
        const float *src = X + (valid[p] ? idx[p] : 0) * dim;
#pragma unroll
        for (int d = 0; d < DIM_MAX; d++) x[p][d] = d < dim ? src[d] : 0.f;
        best[p] = second[p] = 1.7976931348623157e308;
        best_j[p] = -1;
    }
    // A record is fetched by LDS-DMA while the one before it is being used (staging through registers under the arithmetic
    // cost seven registers this kernel does not have: 1.15 s instead of 1.07 for the K = 2048 start; two barriers and an exposed
    // L2 round trip per chunk, as the exact pass has them, are a third of that pass's time).
    auto fetch = [&](double *dst, int chunk) {
        const double *src = rec + (size_t)chunk * REC;
#pragma unroll
        for (int i = 0; i < (PIECES + 3) / 4; i++) {
            const int piece = i * 4 + wave;
            if (piece < PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 128 + lane * 2),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 128), 16, 0, 0);
        }
    };
    auto chunk_pass = [&](const double *cur, int c0) {
        const km_d2 *cs = reinterpret_cast<const km_d2 *>(cur);
        const double *cns = cur + DIM_MAX * KM_CH;
        double acc[KM_PP][KM_CH];
#pragma unroll
        for (int p = 0; p < KM_PP; p++)
#pragma unroll
            for (int j = 0; j < KM_CH; j++) acc[p][j] = cns[j];
#pragma unroll
        for (int d = 0; d < DIM_MAX; d++) {
            if (d < dim) {
                double xv[KM_PP];
#pragma unroll
                for (int p = 0; p < KM_PP; p++) xv[p] = (double)x[p][d];
#pragma unroll
                for (int j2 = 0; j2 < KM_CH / 2; j2++) {
                    const km_d2 c = cs[d * (KM_CH / 2) + j2];
#pragma unroll
                    for (int p = 0; p < KM_PP; p++) {
                        acc[p][2 * j2] = fma(xv[p], c.x, acc[p][2 * j2]);
                        acc[p][2 * j2 + 1] = fma(xv[p], c.y, acc[p][2 * j2 + 1]);
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < KM_PP; p++)
#pragma unroll
            for (int j = 0; j < KM_CH; j++) {                     // (the padding of the last record holds DBL_MAX: it loses to everything)
                const double v = acc[p][j];
                second[p] = fmin(second[p], fmax(v, best[p]));    // the loser of (v, best) against the runner-up so far
                best_j[p] = v < best[p] ? c0 + j : best_j[p];
                best[p] = fmin(best[p], v);
            }
    };
    const int n_chunks = (K + KM_CH - 1) / KM_CH;
    fetch(buf_a, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch = 0; ch < n_chunks; ch += 2) {
        if (ch + 1 < n_chunks) fetch(buf_b, ch + 1);
        chunk_pass(buf_a, ch * KM_CH);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the next record have landed ...
        __syncthreads();                                          // ... and everybody's; and everybody is done with buf_a
        if (ch + 1 >= n_chunks) break;
        if (ch + 2 < n_chunks) fetch(buf_a, ch + 2);
        chunk_pass(buf_b, (ch + 1) * KM_CH);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const double cmax = *cn_max;
#pragma unroll
    for (int p = 0; p < KM_PP; p++) {
        if (!valid[p]) continue;
        double xn = 0.0, d2 = 1.7976931348623157e308;
        if (best_j[p] >= 0) {
            const double *c = C + (size_t)best_j[p] * dim;
            d2 = 0.0;
#pragma unroll
            for (int d = 0; d < DIM_MAX; d++)
                if (d < dim) {
                    const double xd = (double)x[p][d];
                    const double t = __dsub_rn(xd, c[d]);
                    d2 = __dadd_rn(d2, __dmul_rn(t, t));          // the reference's arithmetic for the centre that won
                    xn = fma(xd, xd, xn);
                }
        }
        dist[idx[p]] = d2;
        belong[idx[p]] = best_j[p];
        const double tau = ldexp(xn + cmax, -40);
        if (best_j[p] >= 0 && !(second[p] - best[p] >= tau && best[p] < 1.0e299)) {     // too close to call (or no finite centre): the exact pass decides
            const int at = atomicAdd(n_list, 1);
            list[at] = (int)idx[p];
        }
    }
}

struct KmWorkspace {
    DevBuf<float> X;
    DevBuf<double> C, dist;
    DevBuf<int> belong;
    DevBuf<double> m2c, cn_max;          // the fast full search: its chunk records (-2 c, ||c||^2), the largest ||c||^2
    DevBuf<int> list, n_list;           // points it leaves to the exact pass
    // Lloyd on the full data, cluster sums on the device
    DevBuf<unsigned> key_in, key_out, idx_in, idx_out, seg;
    DevBuf<char> sort_tmp;
    DevBuf<double> csum, bsum, segsum;
    DevBuf<int> csize;
};
KmWorkspace &kws() { return per_device<KmWorkspace>(); }
int &kmeans_assign_engine_option() {        // 0: the fast full search with the exact pass behind it; 1: the exact pass alone
    static int v = 0;
    return v;
}
std::atomic<long> g_km_fast_passes{0}, g_km_fast_rechecked{0};     // full searches taken the fast way; points they left to the exact pass

// key = worker block * K + cluster (a point without a cluster -- NaN centres -- sorts behind everything); value = the point
__global__ __launch_bounds__(256)
void lloyd_keys_kernel(const int *__restrict__ belong, long n, long block, int K, unsigned n_keys, unsigned *__restrict__ key,
                       unsigned *__restrict__ idx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = belong[i];
    key[i] = j >= 0 ? (unsigned)(i / block) * (unsigned)K + (unsigned)j : n_keys;
    idx[i] = (unsigned)i;
}

// seg[k] = first position of key k in the sorted keys (k = 0 .. n_keys inclusive)
__global__ __launch_bounds__(256)
void lloyd_segments_kernel(const unsigned *__restrict__ sorted, long n, unsigned n_keys, unsigned *__restrict__ seg) {
    const unsigned k = blockIdx.x * 256 + threadIdx.x;
    if (k > n_keys) return;
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (sorted[mid] < k) lo = mid + 1; else hi = mid;
    }
    seg[k] = (unsigned)lo;
}

// One thread per (block b, cluster j, dimension d): the members of (b, j) one after the other in point order into a sum
// that starts at 0.0 (calc_belonging's new_centroids, kmeans.cc:78-103); consecutive threads take consecutive dimensions of
// the same member rows.
__global__ __launch_bounds__(256)
void lloyd_segment_sum_kernel(const float *__restrict__ X, int dim, size_t n_seg_elems, const unsigned *__restrict__ seg,
                              const unsigned *__restrict__ idx, double *__restrict__ buf) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_seg_elems) return;
    const size_t k = e / dim;
    const int d = (int)(e - k * dim);
    const unsigned lo = seg[k], hi = seg[k + 1];
    double s = 0.0;
    for (unsigned p = lo; p < hi; p++) s = __dadd_rn(s, (double)X[(size_t)idx[p] * dim + d]);
    buf[e] = s;
}

// One thread per (cluster j, dimension d): the blocks' sums in block order into a total that starts at 0.0 (Lloyds_iteration,
// kmeans.cc:198-204) -- an empty (block, cluster) adds its 0.0 like the reference -- and the cluster's size.
__global__ __launch_bounds__(256)
void lloyd_centroid_kernel(int dim, int K, int n_blocks, const unsigned *__restrict__ seg, const double *__restrict__ buf,
                           double *__restrict__ csum, int *__restrict__ csize) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= K * dim) return;
    const int j = e / dim;
    double total = 0.0;
    int count = 0;
    for (int b = 0; b < n_blocks; b++) {
        total = __dadd_rn(total, buf[(size_t)b * K * dim + e]);
        count += (int)(seg[(size_t)b * K + j + 1] - seg[(size_t)b * K + j]);
    }
    csum[e] = total;
    if (e == j * dim) csize[j] = count;
}

// the sum of a worker block's distances in point order (calc_belonging's distsqr_sum, kmeans.cc:85-106): a workgroup per
// block stages the distances through LDS with coalesced loads, one thread adds them one after the other (a lone thread
// walking global memory pays a cache-miss latency per addend: 8 ms per Lloyd step at 4 k points per block)
constexpr int KM_BS_CHUNK = 4096;
__global__ __launch_bounds__(256)
void lloyd_block_sum_kernel(const double *__restrict__ dist, long n, long block, int n_blocks, double *__restrict__ bsum) {
    __shared__ double stage[KM_BS_CHUNK];
    const int b = blockIdx.x;
    const long lo = (long)b * block, hi = min(n, lo + block);
    double s = 0.0;
    for (long c0 = lo; c0 < hi; c0 += KM_BS_CHUNK) {
        const int m = (int)min((long)KM_BS_CHUNK, hi - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 256) stage[i] = dist[c0 + i];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < m; i++) s = __dadd_rn(s, stage[i]);
    }
    if (threadIdx.x == 0) bsum[b] = s;
}

void device_assign(long n, int dim, const std::vector<double> &C, int c_begin, int c_end,
                   std::vector<double> &dist, std::vector<int> &belong, bool reset, bool download = true) {
    auto &w = kws();
    w.C.upload(C.data(), (size_t)c_end * dim);
    if (reset) {
        w.dist.ensure((size_t)n);
        w.belong.ensure((size_t)n);
    } else {
        w.dist.upload(dist.data(), (size_t)n);
        w.belong.upload(belong.data(), (size_t)n);
    }
    const unsigned grid = (unsigned)((n + 256 * KM_PP - 1) / (256 * KM_PP));
    const bool fast = reset && c_begin == 0 && dim <= 40 && kmeans_assign_engine_option() == 0 && n < ((long)1 << 31);
    if (fast) {
        // one fused multiply-add per point, centre and dimension decides wherever it can; the rest goes through the exact pass
        const int K = c_end;
        const int dim_max = dim <= 16 ? 16 : 40;
        const int n_chunks = (K + KM_CH - 1) / KM_CH;
        w.m2c.ensure((size_t)n_chunks * km_rec_doubles(dim_max));
        w.cn_max.ensure(1);
        w.list.ensure((size_t)n);
        w.n_list.ensure(1);
        SR_HIP(hipMemsetAsync(w.cn_max.p, 0, sizeof(double), ctx().stream));
        SR_HIP(hipMemsetAsync(w.n_list.p, 0, sizeof(int), ctx().stream));
        hipLaunchKernelGGL(kmeans_centre_prep_kernel, dim3((unsigned)((n_chunks * KM_CH + 255) / 256)), dim3(256), 0, ctx().stream, w.C.p, K, dim,
                           dim_max, w.m2c.p, w.cn_max.p);
        if (dim <= 16)
            hipLaunchKernelGGL(kmeans_assign_fast_kernel<16>, dim3(grid), dim3(256), 0, ctx().stream, w.X.p, n, dim, w.C.p, w.m2c.p,
                               w.cn_max.p, K, w.dist.p, w.belong.p, w.list.p, w.n_list.p);
        else
            hipLaunchKernelGGL(kmeans_assign_fast_kernel<40>, dim3(grid), dim3(256), 0, ctx().stream, w.X.p, n, dim, w.C.p, w.m2c.p,
                               w.cn_max.p, K, w.dist.p, w.belong.p, w.list.p, w.n_list.p);
    }
    const int *list = fast ? w.list.p : nullptr;
    const int *n_list = fast ? w.n_list.p : nullptr;
#define SR_KM_LAUNCH(DM)                                                                                                    \
    hipLaunchKernelGGL(kmeans_assign_kernel<DM>, dim3(grid), dim3(256), 0, ctx().stream, w.X.p, n, dim, w.C.p, c_begin, c_end, \
                       w.dist.p, w.belong.p, reset ? 1 : 0, list, n_list)
    if (dim <= 16) SR_KM_LAUNCH(16);
    else if (dim <= 40) SR_KM_LAUNCH(40);
    else if (dim <= 64) SR_KM_LAUNCH(64);
    else SR_KM_LAUNCH(128);
#undef SR_KM_LAUNCH
    SR_HIP(hipGetLastError());
    if (!download) return;                    // (Lloyd on the full data goes on with them on the device)
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
    (void)X;                                   // (resident on the device: kws().X)
    auto &w = kws();
    std::vector<double> best_centroids, dummy_d;
    std::vector<int> dummy_i;
    double best = std::numeric_limits<double>::max(), last = std::numeric_limits<double>::max();
    const long block = (long)std::ceil((double)n / concurrency);
    const int n_blocks = (int)((n + block - 1) / block);
    if ((double)n_blocks * K >= 4.0e9 || n >= ((long)1 << 32)) fail("k-means: %d worker blocks x %d clusters x %ld points do not fit 32-bit sort keys", n_blocks, K, n);
    const unsigned n_keys = (unsigned)n_blocks * (unsigned)K;
    int end_bit = 1;
    while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)n_keys) end_bit++;      // keys 0 .. n_keys inclusive
    w.key_in.ensure((size_t)n);
    w.key_out.ensure((size_t)n);
    w.idx_in.ensure((size_t)n);
    w.idx_out.ensure((size_t)n);
    w.seg.ensure((size_t)n_keys + 2);
    w.csum.ensure((size_t)K * dim);
    w.csize.ensure((size_t)K);
    w.bsum.ensure((size_t)n_blocks);
    w.segsum.ensure((size_t)n_keys * dim);
    size_t tmp_bytes = 0;
    SR_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, w.key_in.p, w.key_out.p, w.idx_in.p, w.idx_out.p, (size_t)n, 0u, (unsigned)end_bit, ctx().stream));
    w.sort_tmp.ensure(tmp_bytes);
    std::vector<double> csum((size_t)K * dim), bsum((size_t)n_blocks);
    std::vector<int> csize((size_t)K);
    for (int iter = 0; iter < KM_MAX_ITER; iter++) {
        device_assign(n, dim, centroids, 0, K, dummy_d, dummy_i, true, false);
        const unsigned g256 = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(lloyd_block_sum_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx().stream, w.dist.p, n, block, n_blocks,
                           w.bsum.p);
        hipLaunchKernelGGL(lloyd_keys_kernel, dim3(g256), dim3(256), 0, ctx().stream, w.belong.p, n, block, K, n_keys, w.key_in.p, w.idx_in.p);
        size_t tb = w.sort_tmp.n;
        SR_HIP(rocprim::radix_sort_pairs(w.sort_tmp.p, tb, w.key_in.p, w.key_out.p, w.idx_in.p, w.idx_out.p, (size_t)n, 0u, (unsigned)end_bit,
                                         ctx().stream));                                     // stable: point order inside a key
        hipLaunchKernelGGL(lloyd_segments_kernel, dim3((n_keys + 1 + 255) / 256), dim3(256), 0, ctx().stream, w.key_out.p, n, n_keys, w.seg.p);
        const size_t n_seg_elems = (size_t)n_keys * dim;
        hipLaunchKernelGGL(lloyd_segment_sum_kernel, dim3((unsigned)((n_seg_elems + 255) / 256)), dim3(256), 0, ctx().stream, w.X.p, dim,
                           n_seg_elems, w.seg.p, w.idx_out.p, w.segsum.p);
        hipLaunchKernelGGL(lloyd_centroid_kernel, dim3((unsigned)((K * dim + 255) / 256)), dim3(256), 0, ctx().stream, dim, K, n_blocks,
                           w.seg.p, w.segsum.p, w.csum.p, w.csize.p);
        SR_HIP(hipGetLastError());
        w.bsum.download(bsum.data(), bsum.size());
        w.csum.download(csum.data(), csum.size());
        w.csize.download(csize.data(), csize.size());
        int rechecked = 0;
        const bool fast_pass = w.n_list.p != nullptr && kmeans_assign_engine_option() == 0 && dim <= 40;
        if (fast_pass) w.n_list.download(&rechecked, 1);
        sync_stream();
        if (fast_pass) {
            g_km_fast_passes++;
            g_km_fast_rechecked += rechecked;
        }
        double sum = 0;
        for (int b = 0; b < n_blocks; b++) sum += bsum[b];
        if (sum < best) {
            best = sum;
            best_centroids = centroids;
        }
        if (verbosity >= 2) printf("k-means iteration %3d: %f\n", iter, sum);
        if (std::fabs(last - sum) < 1e-6) break;
        if (sum > best * 1.5) break;                          // terminate_cost_factor
        for (int k = 0; k < K; k++)
            for (int d = 0; d < dim; d++)
                centroids[(size_t)k * dim + d] = csum[(size_t)k * dim + d] / (double)csize[k];    // an empty cluster divides by 0, as the reference
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
            const int n_threads = std::max(1, std::min({n_blocks, (int)std::thread::hardware_concurrency(), 64}));
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
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
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
    const double t_rounds = now();
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
    // The candidates' distance updates are independent of one another (only the blocked sum below has an order): when the
    // candidate set is large (2 K+ candidates x K centres x dim: 0.24 s on one thread at K = 2048) a few host threads,
    // started once and stepped through the K rounds by a pair of counters, share them.
    const int pp_threads = (size_t)np * dim > ((size_t)1 << 16) ? std::max(1, std::min((int)std::thread::hardware_concurrency(), 16)) : 1;
    std::atomic<int> round_go{0}, round_done{0};
    auto upd = [&](int t, int k) {
        const double *c = centroids.data() + (size_t)(k - 1) * dim;
        const int lo = (int)((long)np * t / pp_threads), hi = (int)((long)np * (t + 1) / pp_threads);
        for (int i = lo; i < hi; i++)
            pdist[i] = std::min(pdist[i], sparse_distsqr(cand.data() + (size_t)i * dim, c, dim) * weight[i]);
    };
    // (joined on every way out: an exception between here and the end -- bad_alloc, a fail() -- would otherwise destroy
    // joinable threads, i.e. std::terminate, with the workers spinning on a round that never comes)
    std::atomic<bool> abort_workers{false};
    struct Joiner {
        std::vector<std::thread> th;
        std::atomic<bool> &abort;
        ~Joiner() {
            abort.store(true, std::memory_order_release);
            for (auto &t : th)
                if (t.joinable()) t.join();
        }
    } workers{{}, abort_workers};
    auto &pp_workers = workers.th;
    for (int t = 1; t < pp_threads; t++)
        pp_workers.emplace_back([&, t] {
            for (int k = 1; k < K; k++) {
                while (round_go.load(std::memory_order_acquire) < k) {
                    if (abort_workers.load(std::memory_order_acquire)) return;
                    std::this_thread::yield();
                }
                upd(t, k);
                round_done.fetch_add(1, std::memory_order_release);
            }
        });
    for (int k = 1; k < K; k++) {
        round_go.store(k, std::memory_order_release);              // centroid k - 1 is in place
        upd(0, k);
        while (round_done.load(std::memory_order_acquire) < (pp_threads - 1) * k) std::this_thread::yield();
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
    for (auto &t : pp_workers) t.join();
    const double t_pp = now();
    lloyd_weighted(cand, weight, np, dim, K, concurrency, centroids);
    const double t_lw = now();
    lloyd_full(X, n, dim, K, concurrency, centroids, verbosity);
    if (verbosity >= 2)
        printf("k-means|| phases: oversampling rounds %.3f s (%d candidates), weighted k-means++ %.3f s, weighted Lloyd %.3f s, "
               "Lloyd on the full data %.3f s\n", t_rounds - t_start, np, t_pp - t_rounds, t_lw - t_pp, now() - t_lw);
    return centroids;
}

}  // namespace

void set_kmeans_assign_engine(int v) { kmeans_assign_engine_option() = v; }
void kmeans_fast_stats(long *passes, long *rechecked) {
    if (passes) *passes = g_km_fast_passes.load();
    if (rechecked) *rechecked = g_km_fast_rechecked.load();
}

// The reference library's other draws from rand(), made by the legacy entry points so that the library's stream
// stays in step with the reference's: one per Gaussian that GMM::load constructs (gmm.cc:671-676), one for the
// trainer plus one per Gaussian in train_model_from_ubm (gmmubm.cc:29-38).
void burn_reference_rand(int count) {
    std::lock_guard<std::mutex> lock(g_rand_mutex);
    for (int i = 0; i < count; i++) (void)g_reference_rand.next();
}
// fork support (common.cpp, fork_proxy.cpp): a forked child gets a fresh lock; the generator's state travels to the helper
// process that computes on a forked child's behalf and back, so that the library-wide stream stays the one the reference's
// process-wide rand() would be
void reference_rand_fork_child() { new (&g_rand_mutex) std::mutex(); }
void reference_rand_state(int32_t *words36, bool set) {
    std::lock_guard<std::mutex> lock(g_rand_mutex);
    if (set) {
        for (int i = 0; i < 34; i++) g_reference_rand.r[i] = words36[i];
        g_reference_rand.f = words36[34];
        g_reference_rand.b = words36[35];
    } else {
        for (int i = 0; i < 34; i++) words36[i] = g_reference_rand.r[i];
        words36[34] = g_reference_rand.f;
        words36[35] = g_reference_rand.b;
    }
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
