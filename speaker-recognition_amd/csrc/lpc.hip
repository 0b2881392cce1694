// lpc.hip -- LPC-15 features per frame (the second half of the reference's mix_feature,
// src/feature/__init__.py:25-30 -> src/feature/LPC.py:14-75): frame, Hamming window and
// pre-emphasis as in MFCC.py (LPC.py:49-53), then scikits.talkbox's lpc(frame, order)[0][1:]
// (LPC.py:40-42) = biased autocorrelation + Levinson-Durbin; NaN -> 0 (LPC.py:56).
// talkbox is third-party and absent: the algorithm is restated from its published form
// (oracle/lpc_oracle.py, "parity unpinned") and cross-checked against a Toeplitz solve.
//
// The Toeplitz systems of voiced frames are ill-conditioned (cond ~1e4..1e6), so the whole chain
// -- windowed samples, lag products, recursion -- runs in float64; fp32 would lose 3 digits.
// Mapping: one wave64 per frame, a contiguous run of samples per lane; the frame goes through LDS
// once so every lane can read its run plus the 15-sample lag halo; 16 lag sums by DPP/readlane
// wave reductions; the order-15 recursion is evaluated redundantly by every lane (wave-uniform).
#include "batch.hpp"
#include "mfcc.hpp"
#include "wave_ops.hpp"

#include <algorithm>
#include <cmath>

namespace sr {

constexpr int LPC_MAX_SPL = 32;        // samples per lane: frames up to 2048 samples

template <typename PcmT, int SPL, int ORDER>
__global__ __launch_bounds__(256)
void lpc_frames_kernel(const PcmT *__restrict__ pcm, const int64_t *__restrict__ sample_off,
                       const int64_t *__restrict__ frame_off, int n_utt, int64_t n_frames,
                       int64_t frames_per_wave, const double *__restrict__ window, int frame_len,
                       int frame_shift, double pre_emph, float *__restrict__ out, int out_stride,
                       int col_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int YLEN = 64 * SPL + ORDER + 1;
    double *y = reinterpret_cast<double *>(smem) + (size_t)wave * YLEN;
    const int L = frame_len;

    const int64_t gwave = (int64_t)blockIdx.x * 4 + wave;
    const int64_t f_begin = gwave * frames_per_wave;
    const int64_t f_end = f_begin + frames_per_wave < n_frames ? f_begin + frames_per_wave : n_frames;
    if (f_begin >= f_end) return;

    // lane-constant window taps of this lane's run [lane*SPL, lane*SPL + SPL) and its predecessor
    double w0[SPL], wm[SPL];
#pragma unroll
    for (int j = 0; j < SPL; j++) {
        const int i = lane * SPL + j;
        w0[j] = i < L ? window[i] : 0.0;
        wm[j] = (i > 0 && i < L) ? window[i - 1] * pre_emph : 0.0;
    }
    // the lag halo beyond the frame is zero for every frame
    for (int i = lane; i < ORDER + 1; i += 64) y[64 * SPL + i] = 0.0;

    int utt = 0;
    {
        int lo = 0, hi = n_utt;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (frame_off[mid] <= f_begin) lo = mid; else hi = mid;
        }
        utt = lo;
    }
    int64_t utt_f0 = frame_off[utt], utt_f1 = frame_off[utt + 1], utt_s0 = sample_off[utt];

    for (int64_t frame = f_begin; frame < f_end; frame++) {
        while (frame >= utt_f1) {
            utt++;
            utt_f0 = utt_f1;
            utt_f1 = frame_off[utt + 1];
            utt_s0 = sample_off[utt];
        }
        const PcmT *fp = pcm + utt_s0 + (frame - utt_f0) * frame_shift;
        // ---- y[i] = w[i] x[i] - pre w[i-1] x[i-1]  (window first, then pre-emphasis; LPC.py:50-53) ----
        double yy[SPL];
        {
            const int i0 = lane * SPL;
            double prev = (i0 > 0 && i0 - 1 < L) ? (double)fp[i0 - 1] : 0.0;
#pragma unroll
            for (int j = 0; j < SPL; j++) {
                const int i = i0 + j;
                const double cur = i < L ? (double)fp[i] : 0.0;
                yy[j] = w0[j] * cur - wm[j] * prev;
                prev = cur;
            }
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < SPL; j++) y[lane * SPL + j] = yy[j];
        wave_sync();
        // ---- lag products of this lane's run against the run + halo ----
        double acc[ORDER + 1];
#pragma unroll
        for (int k = 0; k <= ORDER; k++) acc[k] = 0.0;
        {
            double win[SPL + ORDER];
#pragma unroll
            for (int j = 0; j < SPL + ORDER; j++) win[j] = y[lane * SPL + j];
#pragma unroll
            for (int j = 0; j < SPL; j++)
#pragma unroll
                for (int k = 0; k <= ORDER; k++) acc[k] = fma(win[j], win[j + k], acc[k]);
        }
        // ---- biased autocorrelation r[k] = sum / L (talkbox acorr_lpc), wave-uniform ----
        double r[ORDER + 1];
#pragma unroll
        for (int k = 0; k <= ORDER; k++) r[k] = wave_sum_f64(acc[k]) / (double)L;
        // ---- Levinson-Durbin (talkbox levinson_1d) ----
        double a[ORDER + 1], t[ORDER + 1];
        a[0] = 1.0;
        double e = r[0];
#pragma unroll
        for (int i = 1; i <= ORDER; i++) {
            double s = r[i];
#pragma unroll
            for (int j = 1; j < i; j++) s = fma(a[j], r[i - j], s);
            const double kk = -s / e;
            a[i] = kk;
#pragma unroll
            for (int j = 1; j < i; j++) t[j] = a[j];
#pragma unroll
            for (int j = 1; j < i; j++) a[j] = fma(kk, t[i - j], a[j]);
            e *= 1.0 - kk * kk;
        }
        if (lane < ORDER) {
            double v = 0.0;
#pragma unroll
            for (int j = 1; j <= ORDER; j++)
                if (lane == j - 1) v = a[j];
            out[frame * out_stride + col_off + lane] = (v == v) ? (float)v : 0.0f;   // NaN -> 0 (LPC.py:56)
        }
    }
}

template <typename PcmT, int SPL>
static void launch_lpc_order(int order, dim3 grid, size_t lds, const PcmT *pcm, const int64_t *soff,
                             const int64_t *foff, int U, int64_t NF, int64_t fpw, const double *win,
                             int L, int shift, double pre, float *out, int stride, int col) {
#define SR_LPC_CASE(O)                                                                                 \
    case O: {                                                                                          \
        auto kern = lpc_frames_kernel<PcmT, SPL, O>;                                                   \
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                               \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx().stream, pcm, soff, foff, U, NF, fpw, win, \
                           L, shift, pre, out, stride, col);                                           \
    } break;
    switch (order) {
        SR_LPC_CASE(10) SR_LPC_CASE(12) SR_LPC_CASE(15) SR_LPC_CASE(16) SR_LPC_CASE(20)
        default: fail("LPC order %d is not instantiated (10, 12, 15, 16, 20)", order);
    }
#undef SR_LPC_CASE
}

// Device window table (float64) of an extractor, created on first use.
static const double *lpc_window(SRMfcc &m) {
    std::shared_ptr<void> &slot = m.dev_window_f64[current_device()];
    if (!slot) {
        auto buf = std::make_shared<DevBuf<double>>();
        buf->upload(m.window.data(), m.window.size());
        sync_stream();
        slot = buf;
    }
    return std::static_pointer_cast<DevBuf<double>>(slot)->p;
}

// LPC coefficients of every frame of `pcm` into columns [col_off, col_off + n_lpc) of `out`.
void lpc_extract_into(SRMfcc &m, SRBatch &pcm, const int64_t *d_frame_off, int64_t n_frames, int n_lpc,
                      float *out, int out_stride, int col_off) {
    if (n_frames <= 0) return;
    const double *win = lpc_window(m);
    const int spl = (m.frame_len + 63) / 64;
    const int spl_t = spl <= 8 ? 8 : spl <= 16 ? 16 : 32;
    if (spl > LPC_MAX_SPL) fail("frame of %d samples is too long for the LPC kernel", m.frame_len);
    if (n_lpc >= m.frame_len) fail("LPC order %d needs a longer frame", n_lpc);
    const size_t lds = (size_t)4 * (64 * spl_t + n_lpc + 1) * sizeof(double);
    const int64_t max_waves = (int64_t)ctx().n_cu * 4 * 8;
    const int64_t fpw = std::max<int64_t>(8, (n_frames + max_waves - 1) / max_waves);
    const int64_t n_waves = (n_frames + fpw - 1) / fpw;
    dim3 grid((unsigned)((n_waves + 3) / 4));
    ScopedKernelTimer t(T_MFCC);
#define SR_LPC_SPL(S)                                                                                     \
    if (pcm.kind == SRBatch::PCM16)                                                                       \
        launch_lpc_order<int16_t, S>(n_lpc, grid, lds, pcm.pcm16.p, pcm.d_offsets.p, d_frame_off, pcm.n_utt, \
                                     n_frames, fpw, win, m.frame_len, m.frame_shift, m.pre_emph, out,     \
                                     out_stride, col_off);                                                \
    else                                                                                                  \
        launch_lpc_order<float, S>(n_lpc, grid, lds, pcm.data.p, pcm.d_offsets.p, d_frame_off, pcm.n_utt,  \
                                   n_frames, fpw, win, m.frame_len, m.frame_shift, m.pre_emph, out,       \
                                   out_stride, col_off);
    switch (spl_t) {
        case 8: SR_LPC_SPL(8) break;
        case 16: SR_LPC_SPL(16) break;
        default: SR_LPC_SPL(32) break;
    }
#undef SR_LPC_SPL
    SR_HIP(hipGetLastError());
}

}  // namespace sr
