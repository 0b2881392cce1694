// ltsd.hip -- long-term spectral divergence of half-overlapped Hann windows, the measure behind the
// reference's voice-activity front end (src/filters/ltsd.py:32-64 -> third-party pyssp.vad.ltsd,
// absent from the tree).  The published algorithm (Ramirez et al. 2004, as pyssp
// states it) is restated in oracle/ltsd_oracle.py -- "parity unpinned": no reference vectors exist.
//   window size N = int(0.04644 fs) (ltsd.py:17,66-69; 743 at 16 kHz, 371 at 8 kHz -- not a power of
//   two, and the transform is the exact length-N DFT), hop N/2, windows = len/(N/2) - 1;
//   amp_l[k]  = |DFT_N(frame_l * hann)|[k];
//   LTSE_l[k] = max_{|j| <= order} amp_{l+j}[k];
//   LTSD_l    = 10 log10( (1/N) sum_{k<N} LTSE_l[k]^2 / noise[k]^2 ),  0 at the first/last `order` windows.
// The spectrum of a real frame is symmetric, so bins 0..N/2 are computed and the mirrored ones
// enter the sum through a weight of 2.
//
// Kernel 1: a workgroup takes 4 consecutive windows; the twiddle ring W_N^j and the 4 windowed
// frames (interleaved, one ds_read_b128 per sample index) sit in LDS; a thread owns bins k, k+256, ...
// and walks the ring with stride k (index arithmetic mod N, no trig in the loop): 8 FMAs per
// 2 LDS reads.  Exact-N DFTs have no radix structure to exploit at N = 743 (prime); the
// direct form is N*(N/2+1) complex MACs per window = 1.1 Mflop at 16 kHz, ~0.1 s of vector-ALU time
// per hour of audio.
// Kernel 2: a workgroup per window takes the 2*order+1 neighbour maxima per bin, divides by the
// noise power and reduces (float64 sum).
#include "batch.hpp"
#include "wave_ops.hpp"

#include <cmath>
#include <map>
#include <memory>
#include <vector>

namespace sr {

constexpr int LTSD_WB = 4;          // windows per workgroup
constexpr int LTSD_MAX_N = 4096;

template <typename PcmT>
__global__ __launch_bounds__(256)
void ltsd_amp_kernel(const PcmT *__restrict__ pcm, const int64_t *__restrict__ sample_off,
                     const int64_t *__restrict__ win_off, int n_utt, int64_t n_windows, int N, int shift,
                     const float *__restrict__ window, const float2 *__restrict__ ring, float *__restrict__ amp,
                     int NB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_ring = reinterpret_cast<float2 *>(smem);              // W_N^j = (cos, sin)(2 pi j / N)
    float4 *s_x = reinterpret_cast<float4 *>(s_ring + N);           // [N] x 4 windows
    const int tid = threadIdx.x;
    const int64_t w0 = (int64_t)blockIdx.x * LTSD_WB;

    // which utterance / sample range each of the 4 windows belongs to
    int64_t base[LTSD_WB], limit[LTSD_WB];
#pragma unroll
    for (int w = 0; w < LTSD_WB; w++) {
        const int64_t gw = w0 + w;
        base[w] = 0;
        limit[w] = 0;
        if (gw < n_windows) {
            int lo = 0, hi = n_utt;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (win_off[mid] <= gw) lo = mid; else hi = mid;
            }
            base[w] = sample_off[lo] + (gw - win_off[lo]) * shift;
            limit[w] = sample_off[lo + 1];
        }
    }
    for (int i = tid; i < N; i += 256) {
        s_ring[i] = ring[i];
        const float wv = window[i];
        float xv[LTSD_WB];
#pragma unroll
        for (int w = 0; w < LTSD_WB; w++) {
            const int64_t s = base[w] + i;
            xv[w] = s < limit[w] ? wv * (float)pcm[s] : 0.0f;      // samples past the end count as zero
        }
        s_x[i] = make_float4(xv[0], xv[1], xv[2], xv[3]);
    }
    __syncthreads();

    for (int k = tid; k < NB; k += 256) {
        float re[LTSD_WB] = {0.f, 0.f, 0.f, 0.f}, im[LTSD_WB] = {0.f, 0.f, 0.f, 0.f};
        int idx = 0;
        for (int n = 0; n < N; n++) {
            const float2 t = s_ring[idx];
            const float4 x = s_x[n];
            re[0] = fmaf(x.x, t.x, re[0]); im[0] = fmaf(x.x, t.y, im[0]);
            re[1] = fmaf(x.y, t.x, re[1]); im[1] = fmaf(x.y, t.y, im[1]);
            re[2] = fmaf(x.z, t.x, re[2]); im[2] = fmaf(x.z, t.y, im[2]);
            re[3] = fmaf(x.w, t.x, re[3]); im[3] = fmaf(x.w, t.y, im[3]);
            idx += k;
            idx = idx >= N ? idx - N : idx;
        }
#pragma unroll
        for (int w = 0; w < LTSD_WB; w++)
            if (w0 + w < n_windows) amp[(w0 + w) * NB + k] = sqrtf(re[w] * re[w] + im[w] * im[w]);
    }
}

__global__ __launch_bounds__(256)
void ltsd_reduce_kernel(const float *__restrict__ amp, const int64_t *__restrict__ win_off, int n_utt,
                        int NB, int N, int order, const float *__restrict__ inv_noise_pow,
                        float *__restrict__ ltsd) {
    const int64_t l = blockIdx.x;
    int lo = 0, hi = n_utt;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (win_off[mid] <= l) lo = mid; else hi = mid;
    }
    const int64_t li = l - win_off[lo], wn = win_off[lo + 1] - win_off[lo];
    if (li < order || li + order >= wn) {            // pyssp: no long-term window at the edges -> 0
        if (threadIdx.x == 0) ltsd[l] = 0.0f;
        return;
    }
    double acc = 0.0;
    for (int k = threadIdx.x; k < NB; k += 256) {
        float m = 0.0f;
        for (int j = -order; j <= order; j++) m = fmaxf(m, amp[(l + j) * NB + k]);
        const double wgt = (k == 0 || (2 * k == N)) ? 1.0 : 2.0;       // mirrored bin N-k
        acc += wgt * (double)m * (double)m * (double)inv_noise_pow[k];
    }
    __shared__ double red[4];
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (red[0] + red[1]) + (red[2] + red[3]);
        ltsd[l] = (float)(10.0 * log10(tot / (double)N));
    }
}

// ---------------- host ----------------

struct LtsdPlan {
    int N = 0;
    DevBuf<float> window;
    DevBuf<float2> ring;
};

static LtsdPlan &plan_for(int N) {
    typedef std::map<std::pair<int, int>, std::unique_ptr<LtsdPlan>> PlanMap;
    PlanMap *plans = &per_device<PlanMap>();                               // leaked on purpose
    auto key = std::make_pair(ctx().device, N);
    auto it = plans->find(key);
    if (it != plans->end()) return *it->second;
    auto p = std::make_unique<LtsdPlan>();
    p->N = N;
    std::vector<float> w(N);
    std::vector<float2> r(N);
    for (int i = 0; i < N; i++) {
        // numpy.hanning(N) (ltsd.py:69): 0.5 - 0.5 cos(2 pi i / (N - 1))
        w[i] = N > 1 ? (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / (double)(N - 1))) : 1.0f;
        const double ang = 2.0 * M_PI * i / (double)N;
        r[i] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    p->window.upload(w.data(), w.size());
    p->ring.upload(r.data(), r.size());
    sync_stream();
    LtsdPlan &ref = *p;
    (*plans)[key] = std::move(p);
    return ref;
}

int64_t ltsd_num_windows(int64_t n_samples, int N) {
    const int shift = N / 2;
    if (shift <= 0) fail("LTSD window of %d samples is too short", N);
    const int64_t wn = n_samples / shift - 1;       // pyssp: len(signal)/(winsize/2) - 1
    return wn > 0 ? wn : 0;
}

struct LtsdWork {
    DevBuf<float> amp, ltsd, inv_noise;
    DevBuf<int64_t> d_win_off;
};
static LtsdWork &lwork() { return per_device<LtsdWork>(); }

// amplitude spectra of every window of every utterance -> device [n_windows][NB]; fills win_off [U+1]
static int64_t ltsd_amplitudes(SRBatch &pcm, int N, std::vector<int64_t> &win_off) {
    ensure_device();
    if (pcm.kind != SRBatch::PCM16 && pcm.kind != SRBatch::PCMF32) fail("LTSD needs a PCM batch");
    if (N < 4 || N > LTSD_MAX_N) fail("LTSD window of %d samples is outside 4..%d", N, LTSD_MAX_N);
    LtsdPlan &pl = plan_for(N);
    const int NB = N / 2 + 1, shift = N / 2;
    win_off.assign(pcm.n_utt + 1, 0);
    for (int u = 0; u < pcm.n_utt; u++)
        win_off[u + 1] = win_off[u] + ltsd_num_windows(pcm.offsets[u + 1] - pcm.offsets[u], N);
    const int64_t nw = win_off[pcm.n_utt];
    auto &w = lwork();
    w.d_win_off.upload(win_off.data(), win_off.size());
    if (nw == 0) {
        sync_stream();
        return 0;
    }
    w.amp.ensure((size_t)nw * NB);
    const size_t lds = (size_t)N * (sizeof(float2) + sizeof(float4));
    dim3 grid((unsigned)((nw + LTSD_WB - 1) / LTSD_WB));
    if (pcm.kind == SRBatch::PCM16) {
        auto kern = ltsd_amp_kernel<int16_t>;
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx().stream, pcm.pcm16.p, pcm.d_offsets.p, w.d_win_off.p,
                           pcm.n_utt, nw, N, shift, pl.window.p, pl.ring.p, w.amp.p, NB);
    } else {
        auto kern = ltsd_amp_kernel<float>;
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx().stream, pcm.data.p, pcm.d_offsets.p, w.d_win_off.p,
                           pcm.n_utt, nw, N, shift, pl.window.p, pl.ring.p, w.amp.p, NB);
    }
    SR_HIP(hipGetLastError());
    return nw;
}

// Mean amplitude spectrum over all windows of the batch (pyssp _compute_noise_avg_spectrum), bins 0..N/2.
void ltsd_noise_spectrum(SRBatch &noise, int N, float *avg_amp_out) {
    std::vector<int64_t> win_off;
    const int64_t nw = ltsd_amplitudes(noise, N, win_off);
    const int NB = N / 2 + 1;
    if (nw == 0) fail("noise signal too short for one LTSD window of %d samples", N);
    std::vector<float> host((size_t)nw * NB);
    lwork().amp.download(host.data(), host.size());
    sync_stream();
    for (int k = 0; k < NB; k++) {
        double a = 0.0;
        for (int64_t l = 0; l < nw; l++) a += (double)host[(size_t)l * NB + k];
        avg_amp_out[k] = (float)(a / (double)nw);
    }
}

// LTSD of every window of every utterance against a noise amplitude spectrum (bins 0..N/2).
void ltsd_compute(SRBatch &pcm, int N, int order, const float *noise_amp, float *ltsd_out,
                  int64_t *win_offsets_out) {
    if (order < 0 || order > 64) fail("LTSD order %d is outside 0..64", order);
    std::vector<int64_t> win_off;
    const int64_t nw = ltsd_amplitudes(pcm, N, win_off);
    const int NB = N / 2 + 1;
    if (win_offsets_out)
        for (size_t i = 0; i < win_off.size(); i++) win_offsets_out[i] = win_off[i];
    if (nw == 0) return;
    auto &w = lwork();
    std::vector<float> inv(NB);
    for (int k = 0; k < NB; k++) {
        const double p = (double)noise_amp[k] * (double)noise_amp[k];
        inv[k] = (float)(1.0 / p);                   // a zero noise bin gives inf, as the division in pyssp does
    }
    w.inv_noise.upload(inv.data(), inv.size());
    w.ltsd.ensure((size_t)nw);
    hipLaunchKernelGGL(ltsd_reduce_kernel, dim3((unsigned)nw), dim3(256), 0, ctx().stream, w.amp.p,
                       w.d_win_off.p, pcm.n_utt, NB, N, order, w.inv_noise.p, w.ltsd.p);
    SR_HIP(hipGetLastError());
    w.ltsd.download(ltsd_out, (size_t)nw);
    sync_stream();
}

}  // namespace sr
