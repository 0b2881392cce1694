// em_small.hip -- a WHOLE EM / MAP fit of a speaker-sized model in one launch.
//
// Why: a speaker model (gui/interface.py:55-109 enrols 32 mixtures; configs[0] 16 x 13 on 3000 frames) is fitted by up to 200
// iterations (gmm.cc:600-651) of a few microseconds of arithmetic each.  Iteration at a time (em.hip: pack, upload, score, statistics,
// sums back to the host, M-step there, every second iteration a pass for the total log-likelihood) an iteration costs ~85 us of launches,
// small copies and host waits around ~25 us of kernels: 17-21 ms per fit, 96 % of configs[0]'s enrol + predict time.
// Here the loop of GMMTrainerBaseline::train (gmm.cc:581-653) runs ON the device: one grid that is resident as a whole, a workgroup of 1024 threads per 64
// or 128 frames (kept in its LDS for the whole fit, beside the model), ONE grid-wide barrier per iteration behind which every workgroup
// adds up everybody's sums itself (two, with the addition shared out, when workgroups x sums is large), the host waits once.  16 x 13 on
// 2998 frames: 19.6 us per iteration (96 iteration at a time), a 200-iteration fit 4.05 ms (17.8); what an iteration costs is its trips
// to the memory side -- a device-scope release, the arrival, the poll, the partial sums: ~11 us with ONE workgroup -- and ~0.3 us per
// workgroup that meets at the barrier (scripts/debug/em_small_time.py, profiles/r06_em_small.txt).
//
// Arithmetic: float64 throughout (the reference's own type, gmm.hh:15) -- log densities, responsibilities, the three sums, the M-step.
//   E-step (gmm.cc:439-498): p_ik = w_k N(x_i; mu_k, sigma_k) taken in the log domain, a term below DBL_MIN = exp(-708.396) is 0 as in
//     the reference's linear-domain product under -ffast-math (lse.hpp), a frame without a surviving term carries no responsibility
//     (:482-498: its sum is replaced by 1e-15 and every quotient is 0) and counts ln 1e-15 in the total (safe_log, :34-38).
//   Sums: N_k, sum g (x - mu_k), sum g (x - mu_k)^2 about the CURRENT mean (the variance then needs no cancelling subtraction of large
//     numbers; em.hip's M-step restated), each (mixture, dimension) added up over a workgroup's frames in frame order, the workgroups'
//     partial sums in workgroup order: the same bits on every run.
//   M-step: N_k = 0 -> 1e-6 (:502-509); weights N_k / n normalised by their sum (:388-394); mean, then the variance about the NEW mean
//     (:396-437) floored at sqrt(min_covar); MAP (gmmubm.cc:53-74): means only, relevance 16.
//   Stop rule (:622-650): after every second iteration the total log-likelihood under the updated model -- which is the denominator
//     pass of the NEXT iteration's E-step, so it costs nothing: the next E-step is computed, its total compared, and on "too small an
//     increment" that iteration's M-step is not applied.
// What this kernel leaves to the iteration-at-a-time path (it raises a flag, the host starts over there): a live frame whose largest
// term lies within 110 nats of the underflow boundary -- where the reference's flushes of PARTIAL products decide (lse.hpp,
// gmm_flush.hip) -- and anything that is not finite.
#include "score.hpp"
#include "wave_ops.hpp"

#include "../../include/pygmm_hip.h"

#include <cfloat>
#include <cerrno>
#include <cmath>
#include <atomic>
#include <cstdio>
#include <fcntl.h>
#include <string>
#include <sys/file.h>
#include <unistd.h>
#include <limits>
#include <vector>

namespace sr {

namespace {

constexpr int EMF_THREADS = 1024;                 // a workgroup: 64 or 128 frames x 16 or 8 mixture groups
constexpr int EMF_MAX_K = 32, EMF_MAX_D = 40;
// the iteration's price grows with the workgroups that meet at its barriers (12 us at one, 27 at 47, 46 at 128, 16 x 13); from ~10 k
// frames on an iteration per launch costs the same (20 000 x 32 x 40: 30 ms either way)
constexpr long EMF_MAX_FRAMES = 8192;
constexpr unsigned EMF_POLL_LIMIT = 60000;        // polls of ~1.5 us (a sleep and a device-scope load), ~0.1 s, before a workgroup gives the grid up
constexpr double EMF_MINLOG = -708.396418532264;  // ln DBL_MIN (fastexp.cc:93,105)
constexpr double EMF_BAND = -598.0;               // a live frame below this goes to the path that restates the partial-product flushes
constexpr double EMF_LN_1E_15 = -34.538776394910684;
constexpr double EMF_SQRT_2_PI = 2.5066282746310002;

struct EmSmallArgs {
    const float *X;            // [n][dim]
    int n, dim, K;
    int fr, seg;               // frames per workgroup (64 / 128), segments of a role's sweep (a power of two, fr / seg a multiple of 4)
    int nr_iter;
    int map;                   // means only (gmmubm.cc:53-74)
    int test_absent;           // test hook (option debug_em_small_absent_workgroup): see the barrier
    double threshold, min_sigma, relevance;
    const double *init;        // [K] weights, [K*D] means, [K*D] sigmas, (map) [K*D] the UBM's means
    double *partials;          // [2][E][grid]  (by the iteration's parity; an entry's row: the workgroups side by side)
    double *totals;            // [E]
    double *out;               // [K] weights, [K*D] means, [K*D] sigmas
    double *ll_hist;           // [nr_iter]: total log-likelihood after iteration i (odd i only; NaN elsewhere)
    int *result;               // [0] iterations carried out, [1] flag (1: frames left to the other path, 2: the grid gave up at a barrier)
    unsigned *barrier;         // grid barrier counter, [32] the abort word (0 at launch)
};

// every workgroup of the (resident) grid arrives; `round` = 1, 2, ... over the barriers of the launch.  Every thread releases its own
// stores at device scope before the workgroup's barrier and acquires behind it (the device's eight L2s are not coherent with each
// other for plain accesses); what crosses workgroups is then read with plain loads.  Measured (scripts/debug/em_small_time.py, 16 x 13
// on 47 workgroups / 32 x 40 on 256): two full fences and acquiring polls 49 / 647 us per iteration; release + acquire fences and relaxed
// polls with a longer sleep 38 / 222 (the pollers' traffic on the one address was most of it).  Arrivals dealt to eight group counters on
// lines of their own: 27.0 -> 26.1 / 89 -> 80, not kept.  The device-scope fences by ONE thread (below): 24.7 -> 21.5 with 16 waves.
// Returns false when the grid has to give up: the launch is an ordinary one (a cooperative launch -- the runtime's promise that every
// workgroup is on the chip at once -- makes rocprofv3 crash in the traced process's exit(); measured, 1 October), the grid is at most one
// workgroup per two CUs, and what a promise would rule out -- three or more processes each holding part of the chip with part of such
// a grid, every one waiting for workgroups that cannot start -- ends here instead (and DeviceFitLock below keeps whole-fit grids from meeting each other at all): a workgroup that has polled for ~0.1 s raises the
// abort word (counter[32]), everybody who polls sees it within 64 polls, the kernel returns, the host takes the other path.
__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned round, bool arrive = true) {
    __shared__ int s_ok;
    // every wave's stores are in the L2 (a workgroup-scope release waits for them) before the workgroup's barrier; ONE thread then
    // writes the L2's dirty lines back, arrives, polls and invalidates (the device-scope fences cost per wave that executes them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (arrive) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (released by the fence above)
        unsigned polls = 0;
        int ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * gridDim.x) {
            __builtin_amdgcn_s_sleep(8);
            if ((++polls & 63u) == 0u) {
                if (__hip_atomic_load(counter + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    ok = 0;
                    break;
                }
                if (polls > EMF_POLL_LIMIT) {
                    __hip_atomic_store(counter + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ok = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return s_ok != 0;
}

// sum over the grid's workgroups of entry e: partials[e][0 .. G), in workgroup order, 16 loads in flight at a time (one after
// the other they cost a trip to the memory side each: 47 of them were most of an iteration).  Plain loads: behind the barrier's acquire.
__device__ __forceinline__ double sum_partials(const double *row, int G) {
    double t = 0.0;
    for (int w0 = 0; w0 < G; w0 += 16) {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = w0 + j < G ? row[w0 + j] : 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) t += v[j];
    }
    return t;
}

__global__ __launch_bounds__(EMF_THREADS)
void em_small_fit_kernel(const EmSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) double emf_lds[];
    const int K = a.K, D = a.dim, n = a.n;
    const int FR = a.fr, NG = EMF_THREADS / FR;    // frames of this workgroup; thread = (frame f, mixture group g of NG)
    const int S = a.seg;                           // segments a (mixture, dimension) pair's sweep over the frames is cut into
    const int REC = 2 * D + 1;                     // a mixture's sums: [D] first moments, [D] second moments, N
    const int E = K * REC + 2;                     // + total log-likelihood, + flag count
    const int R = K * (D + 1);                     // roles of the sums: (k, d < D) both moments, (k, D) N_k
    double *s_w = emf_lds;                         // [K]
    double *s_c = s_w + K;                         // [K]   ln w - sum_d ln(sqrt(2 pi) sigma_d)
    double *s_nk = s_c + K;                        // [K]
    double *s_mu = s_nk + K;                       // [K][D]
    double *s_h = s_mu + K * D;                    // [K][D] 1 / (2 sigma^2)
    double *s_sg = s_h + K * D;                    // [K][D]
    double *s_g = s_sg + K * D;                    // [K][FR]  log densities -> exponentials -> responsibilities
    double *s_pm = s_g + K * FR;                   // [NG][FR] the groups' maxima, then their sums
    double *s_ll = s_pm + EMF_THREADS;             // [FR]
    double *s_tot = s_ll + FR;                     // [E]      the grid's sums of an iteration
    double *s_seg = s_tot + E;                     // [R][S][2] the segments' sums of a role
    float *s_x = reinterpret_cast<float *>(s_seg + 2 * R * S);      // [FR][D + 1]
    const int XS = D + 1;

    const int tid = threadIdx.x, f = tid & (FR - 1), g = tid / FR, lane = tid & 63;
    const int G = gridDim.x, wg = blockIdx.x;
    const int f0 = wg * FR;                        // (one chunk per workgroup: it stays in LDS for the whole fit)
    const bool valid = f0 + f < n;

    for (int i = tid; i < K; i += EMF_THREADS) s_w[i] = a.init[i];
    for (int i = tid; i < K * D; i += EMF_THREADS) {
        s_mu[i] = a.init[K + i];
        s_sg[i] = a.init[K + K * D + i];
    }
    for (int i = tid; i < FR * D; i += EMF_THREADS) {
        const int fr = i / D, d = i - fr * D;
        s_x[fr * XS + d] = f0 + fr < n ? a.X[(size_t)(f0 + fr) * D + d] : 0.f;
    }
    __syncthreads();
    auto derive = [&]() {                           // (a barrier in front of it; s_g is free between two E-steps)
        for (int i = tid; i < K * D; i += EMF_THREADS) {
            s_h[i] = 0.5 / (s_sg[i] * s_sg[i]);
            s_g[i] = log(EMF_SQRT_2_PI * s_sg[i]);  // (one logarithm per thread: a mixture's D of them in a row were most of an M-step)
        }
        __syncthreads();
        for (int k = tid; k < K; k += EMF_THREADS) {
            double c = s_w[k] > 0.0 ? log(s_w[k]) : -__builtin_inf();
            for (int d = 0; d < D; d++) c -= s_g[k * D + d];
            s_c[k] = c;
        }
    };
    derive();
    __syncthreads();

    const bool redundant = (long)G * E <= 48 * 1024;
    unsigned arrivals = 0;
    double last_ll = -DBL_MAX;
    int done = a.nr_iter, flagged = 0;
    for (int it = 0;; it++) {
        const bool ll_only = it == a.nr_iter;      // the total after the LAST iteration, when that one is an odd one (gmm.cc:622)
        if (ll_only && ((a.nr_iter - 1) & 1) == 0) break;

        // ---- E-step over this workgroup's frames ----
        double ll_part = 0.0;
        int bad = 0;
        {
            // log densities of this group's mixtures (k = g, g + NG, ...) for frame f
            double pmax = -__builtin_inf();
            for (int k = g; k < K; k += NG) {
                double lp = s_c[k];
                const double *mu = s_mu + k * D, *h = s_h + k * D;
                for (int d = 0; d < D; d++) {
                    const double t = (double)s_x[f * XS + d] - mu[d];
                    lp = fma(-(t * t), h[d], lp);
                }
                s_g[k * FR + f] = lp;
                if (lp >= EMF_MINLOG) pmax = fmax(pmax, lp);
                if (valid && !(lp == lp)) bad = 1;                 // NaN (a non-finite input or parameter)
            }
            s_pm[g * FR + f] = pmax;
            __syncthreads();
            double m = s_pm[f];
            for (int gg = 1; gg < NG; gg++) m = fmax(m, s_pm[gg * FR + f]);
            const bool live = m >= EMF_MINLOG;                     // some term survives (gmm.cc:482-498)
            __syncthreads();                                       // (s_pm is rewritten)
            double psum = 0.0;
            for (int k = g; k < K; k += NG) {
                const double lp = s_g[k * FR + f];
                const double e = live && lp >= EMF_MINLOG ? exp(lp - m) : 0.0;
                s_g[k * FR + f] = e;
                psum += e;
            }
            s_pm[g * FR + f] = psum;
            __syncthreads();
            double s = s_pm[f];
            for (int gg = 1; gg < NG; gg++) s += s_pm[gg * FR + f];                // (the groups in order)
            const double r = live && valid ? 1.0 / s : 0.0;
            for (int k = g; k < K; k += NG) s_g[k * FR + f] *= r;
            if (g == 0) {
                s_ll[f] = valid ? (live ? m + log(s) : EMF_LN_1E_15) : 0.0;
                if (valid && live && m < EMF_BAND) bad = 1;
            }
            __syncthreads();
            if (tid < 64)                                           // wave 0: the frames' values 64 at a time (fixed order, wave_ops.hpp)
                for (int q = 0; q < FR; q += 64) ll_part += wave_sum_f64(s_ll[q + lane]);
            if (!ll_only) {
                // a role's sweep over the frames, cut into S segments on S threads (16 x 13 = 224 roles leave most of 1024 threads
                // idle otherwise); within a segment four running sums per moment (frames i = q mod 4): short chains, a fixed order
                const int L = FR / S;
                for (int item = tid; item < R * S; item += EMF_THREADS) {
                    const int role = item % R, seg = item / R;
                    const int k = role / (D + 1), d = role - k * (D + 1);
                    const double *gam = s_g + k * FR + seg * L;
                    double p1[4] = {0.0, 0.0, 0.0, 0.0}, p2[4] = {0.0, 0.0, 0.0, 0.0};
                    if (d < D) {
                        const double mu = s_mu[k * D + d];
                        const float *xs = s_x + (seg * L) * XS + d;
                        for (int i = 0; i < L; i += 4) {
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const double dv = (double)xs[(i + q) * XS] - mu;
                                const double gd = gam[i + q] * dv;
                                p1[q] += gd;
                                p2[q] = fma(gd, dv, p2[q]);
                            }
                        }
                    } else {
                        for (int i = 0; i < L; i += 4) {
#pragma unroll
                            for (int q = 0; q < 4; q++) p1[q] += gam[i + q];
                        }
                    }
                    s_seg[(role * S + seg) * 2] = (p1[0] + p1[1]) + (p1[2] + p1[3]);
                    s_seg[(role * S + seg) * 2 + 1] = (p2[0] + p2[1]) + (p2[2] + p2[3]);
                }
            }
        }
        __syncthreads();
        // ---- this workgroup's sums out ([entry][workgroup]: a reader's loads take one entry's row), everybody's in ----
        {
            // (two sets of partial sums, by the iteration's parity: with ONE barrier per iteration a workgroup that is still adding up
            // iteration t's sums -- descheduled for another process's kernel, say -- must not see a faster one's sums of t + 1)
            double *mine = a.partials + (size_t)(it & 1) * E * G + wg;
            if (!ll_only)
                for (int role = tid; role < R; role += EMF_THREADS) {
                    const int k = role / (D + 1), d = role - k * (D + 1);
                    double m1 = 0.0, m2 = 0.0;
                    for (int seg = 0; seg < S; seg++) {                             // (the segments in order)
                        m1 += s_seg[(role * S + seg) * 2];
                        m2 += s_seg[(role * S + seg) * 2 + 1];
                    }
                    if (d < D) {
                        mine[(size_t)(k * REC + d) * G] = m1;
                        mine[(size_t)(k * REC + D + d) * G] = m2;
                    } else {
                        mine[(size_t)(k * REC + 2 * D) * G] = m1;
                    }
                }
            if (tid == 0) mine[(size_t)(K * REC) * G] = ll_part;
            const int any_bad = __syncthreads_or(bad);
            if (tid == 0) mine[(size_t)(K * REC + 1) * G] = any_bad ? 1.0 : 0.0;
        }
        // (test hook: workgroup 1 stays away from the third barrier -- what a workgroup that cannot start looks like to the others)
        ++arrivals;
        if (!grid_barrier(a.barrier, arrivals, !(a.test_absent && wg == 1 && arrivals == 3))) {
            flagged = 2;
            break;
        }
        if (redundant) {
            // few workgroups x few entries: every workgroup adds up everything itself (G x E loads, one trip's latency) -- one
            // grid barrier per iteration
            for (int e = tid; e < E; e += EMF_THREADS) s_tot[e] = sum_partials(a.partials + (size_t)(it & 1) * E * G + (size_t)e * G, G);
        } else {
            for (int e = wg * EMF_THREADS + tid; e < E; e += G * EMF_THREADS)
                a.totals[e] = sum_partials(a.partials + (size_t)(it & 1) * E * G + (size_t)e * G, G);
            if (!grid_barrier(a.barrier, ++arrivals)) {
                flagged = 2;
                break;
            }
            for (int e = tid; e < E; e += EMF_THREADS) s_tot[e] = a.totals[e];
        }
        __syncthreads();

        if (s_tot[K * REC + 1] > 0.0) {
            flagged = 1;
            break;
        }
        // the total under the model as iteration it - 1 left it: the reference takes it after odd iterations (gmm.cc:622-650)
        if (it >= 1 && ((it - 1) & 1)) {
            const double ll = s_tot[K * REC];
            if (wg == 0 && tid == 0) a.ll_hist[it - 1] = ll;
            const double ll_diff = ll - last_ll;
            if (fabs(ll_diff) / fabs(ll) < a.threshold && ll_diff < a.threshold) {
                done = it;
                break;
            }
            last_ll = ll;
        }
        if (ll_only) break;

        // ---- M-step (em.hip's host M-step restated; every workgroup forms the same model) ----
        for (int k = tid; k < K; k += EMF_THREADS) {
            double v = s_tot[k * REC + 2 * D];
            if (v == 0.0) v = 1e-6;                                 // min_n_k, gmm.cc:502-509
            s_nk[k] = v;
        }
        __syncthreads();
        if (!a.map) {                                               // update_weights, gmm.cc:388-394 (the quotients side by side, their sum in order)
            for (int k = tid; k < K; k += EMF_THREADS) s_w[k] = s_nk[k] / (double)n;
            __syncthreads();
            double wsum = 0.0;
            for (int k = 0; k < K; k++) wsum += s_w[k];
            __syncthreads();
            for (int k = tid; k < K; k += EMF_THREADS) s_w[k] /= wsum;
        }
        for (int i = tid; i < K * D; i += EMF_THREADS) {
            const int k = i / D, d = i - k * D;
            const double sd = s_tot[k * REC + d], sdd = s_tot[k * REC + D + d];
            const double nk = s_nk[k], mu_old = s_mu[i];
            const double shift = sd / nk;                           // E_k[x] - mu_old
            if (a.map) {                                            // update_means, gmmubm.cc:53-74
                const double alpha = nk / (nk + a.relevance);
                s_mu[i] = alpha * (mu_old + shift) + (1 - alpha) * a.init[K + 2 * K * D + i];
            } else {                                                // gmm.cc:396-437
                s_mu[i] = mu_old + shift;
                double var = sdd / nk - shift * shift;              // sum g (x - mu_new)^2 = sdd - N shift^2
                if (var < 0) var = 0;
                s_sg[i] = fmax(a.min_sigma, sqrt(var));
            }
        }
        __syncthreads();
        derive();
        __syncthreads();                                            // s_h / s_c for the next E-step; derive() is done with s_g
    }

    if (wg == 0) {
        __syncthreads();
        for (int i = tid; i < K; i += EMF_THREADS) a.out[i] = s_w[i];
        for (int i = tid; i < K * D; i += EMF_THREADS) {
            a.out[K + i] = s_mu[i];
            a.out[K + K * D + i] = s_sg[i];
        }
        if (tid == 0) {
            a.result[0] = done;
            a.result[1] = flagged;
        }
    }
}

// ONE whole-fit grid on a device at a time, across processes.  Its workgroups meet at a barrier, so a grid is only safe while all of it
// fits the chip beside whatever else runs -- and a workgroup of 16 waves x 128 registers fills a CU: eleven processes enrolling at once
// (264 workgroups for 256 CUs) could leave every grid waiting for workgroups that cannot start, each round ending only at the poll limit
// with every other kernel of those processes queued behind the spinning ones.  An advisory lock on a per-device file (released by the
// kernel should the holder die) serialises the launches: a fit holds the chip for its few milliseconds
// (scripts/debug/em_small_stress.py).  No lock to be had: no whole-fit launch.
struct DeviceFitLock {
    int fd = -1;
    bool held = false;
    explicit DeviceFitLock(int device) {
        // the file stays open for the life of the process (asking the runtime for the bus id and opening the file cost 2 ms per
        // fit); a forked child opens its own -- a lock belongs to the open file, which a child would share with its parent
        static pid_t owner = 0;
        static int fds[MAX_DEVICES];
        if (owner != getpid()) {
            owner = getpid();
            for (int &f : fds) f = -1;
        }
        if (device < 0 || device >= MAX_DEVICES) return;
        if (fds[device] < 0) {
            char bus[64] = {0};
            if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
                (void)hipGetLastError();
                return;
            }
            for (char *c = bus; *c; c++)
                if (*c == ':' || *c == '.' || *c == '/') *c = '_';
            const std::string path = "/dev/shm/sr_whole_fit_" + std::to_string((unsigned)getuid()) + "_" + bus + ".lock";
            fds[device] = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
        }
        fd = fds[device];
        if (fd < 0) return;
        int rc;
        do rc = flock(fd, LOCK_EX); while (rc != 0 && errno == EINTR);
        held = rc == 0;
    }
    ~DeviceFitLock() {
        if (held) (void)flock(fd, LOCK_UN);
    }
    DeviceFitLock(const DeviceFitLock &) = delete;
    DeviceFitLock &operator=(const DeviceFitLock &) = delete;
};

std::atomic<int> &em_small_test_absent() {
    static std::atomic<int> v{0};
    return v;
}

struct EmSmallWorkspace {
    DevBuf<double> init, partials, totals, out;
    DevBuf<int> result;            // [0..1] the kernel's answer, [32] the barrier's counter, [64] its abort word (a line each)
    PinnedBuf<double> h_out;       // the model and the totals' history, one copy
    PinnedBuf<int> h_result;
};

}  // namespace

// frames per workgroup, segments of a role's sweep, LDS bytes.  The fewer workgroups meet at the barrier the cheaper the iteration
// (~0.3 us each) and the longer a workgroup's own arithmetic: 16 x 13 on 2998 frames 26.8 us per iteration at 64 frames per workgroup,
// 19.5 at 128, 21.5 at 256 -- 128 where the LDS fits and 64 frames would not do with as few workgroups.
struct EmSmallShape {
    int fr, seg, grid;
    size_t lds;
};
static EmSmallShape em_small_shape(int K, int D, long n) {
    EmSmallShape best{0, 1, 0, 0};
    const int R = K * (D + 1);
    for (int fr : {128, 64}) {
        if (fr > 64 && (n + fr / 2 - 1) / (fr / 2) == (n + fr - 1) / fr) continue;      // (half the frames: as many workgroups)
        int seg = 1;
        while (seg * 2 <= fr / 64 && R * seg * 2 <= EMF_THREADS) seg *= 2;
        const size_t lds = (size_t)(3 * K + 3 * K * D + K * fr + EMF_THREADS + fr + K * (2 * D + 1) + 2 + 2 * R * seg) * sizeof(double) +
                           (size_t)fr * (D + 1) * sizeof(float);
        if (lds > 150 * 1024) continue;
        best = {fr, seg, (int)((n + fr - 1) / fr), lds};
        break;
    }
    return best;
}

void set_em_small_test_absent(int v) { em_small_test_absent().store(v); }

bool em_small_eligible(int K, int dim, long n, const Parameter &param) {
    return K >= 1 && K <= EMF_MAX_K && dim >= 1 && dim <= EMF_MAX_D && n >= 1 &&
           n <= EMF_MAX_FRAMES && em_small_shape(K, dim, n).grid >= 1 && em_small_shape(K, dim, n).grid <= ctx().n_cu / 2 && param.nr_iteration >= 1 && param.verbosity < 2;
}

// The fit of `gmm` (its parameters are the start) on the n resident frames dX.  true: done -- gmm holds the result, *iterations the
// count train_em returns; false: the kernel met what it leaves to the iteration-at-a-time path (gmm untouched).
bool train_em_small(GMM &gmm, const GMM *ubm, const float *dX, long n, int dim, const Parameter &param, double relevance, int *iterations) {
    const int K = gmm.nr_mixtures, KD = K * dim;
    int device = 0;
    SR_HIP(hipGetDevice(&device));
    DeviceFitLock fit_lock(device);               // (held until this function has the kernel's answer)
    if (!fit_lock.held) return false;
    auto &w = per_device<EmSmallWorkspace>();
    const EmSmallShape shape = em_small_shape(K, dim, n);
    const int grid = shape.grid;
    const int E = K * (2 * dim + 1) + 2;
    const int nit = param.nr_iteration;

    std::vector<double> init((size_t)K + 2 * (size_t)KD + (ubm ? KD : 0));
    for (int k = 0; k < K; k++) init[k] = gmm.weights[k];
    for (int i = 0; i < KD; i++) {
        init[K + i] = gmm.mean[i];
        init[K + KD + i] = gmm.sigma[i];
        if (ubm) init[K + 2 * KD + i] = ubm->mean[i];
    }
    w.init.upload(init.data(), init.size());
    w.partials.ensure((size_t)2 * grid * E);
    w.totals.ensure((size_t)E);
    w.out.ensure((size_t)K + 2 * (size_t)KD + (size_t)nit);
    w.result.ensure(96);
    w.h_out.ensure((size_t)K + 2 * (size_t)KD + (size_t)nit);
    w.h_result.ensure(4);
    SR_HIP(hipMemsetAsync(w.result.p, 0, 96 * sizeof(int), ctx().stream));
    // (NaN: "no total taken after this iteration")
    SR_HIP(hipMemsetAsync(w.out.p + K + 2 * KD, 0xff, (size_t)nit * sizeof(double), ctx().stream));

    EmSmallArgs a;
    a.X = dX;
    a.n = (int)n;
    a.dim = dim;
    a.K = K;
    a.fr = shape.fr;
    a.seg = shape.seg;
    a.nr_iter = nit;
    a.map = ubm ? 1 : 0;
    a.test_absent = em_small_test_absent().load();
    a.threshold = param.threshold;
    a.min_sigma = std::sqrt(param.min_covar);
    a.relevance = relevance;
    a.init = w.init.p;
    a.partials = w.partials.p;
    a.totals = w.totals.p;
    a.out = w.out.p;
    a.ll_hist = w.out.p + K + 2 * KD;
    a.result = w.result.p;
    a.barrier = reinterpret_cast<unsigned *>(w.result.p + 32);
    const size_t lds = shape.lds;
    if (lds > 64 * 1024)
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&em_small_fit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // an ordinary launch of <= n_cu / 2 workgroups, one per CU: they are all on the chip at once unless other processes hold most of it --
    // and then the barrier's poll limit ends the wait (grid_barrier)
    hipLaunchKernelGGL(em_small_fit_kernel, dim3((unsigned)grid), dim3(EMF_THREADS), lds, ctx().stream, a);
    SR_HIP(hipGetLastError());
    w.out.download(w.h_out.p, (size_t)K + 2 * (size_t)KD + (size_t)nit);
    w.result.download(w.h_result.p, 2);
    sync_stream();
    if (w.h_result.p[1]) return false;
    const int done = w.h_result.p[0];
    for (int k = 0; k < K; k++) gmm.weights[k] = w.h_out.p[k];
    for (int i = 0; i < KD; i++) {
        gmm.mean[i] = w.h_out.p[K + i];
        gmm.sigma[i] = w.h_out.p[K + KD + i];
    }
    gmm.drop_single();
    if (param.verbosity >= 1) {
        const double *hist = w.h_out.p + K + 2 * KD;
        for (int i = 1; i < done && i < nit; i += 2)
            if (hist[i] == hist[i]) printf("iter %d: ll %lf\n", i, hist[i]);
    }
    *iterations = done;
    return true;
}

}  // namespace sr
