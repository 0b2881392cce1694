// mfcc.hip -- the reference's own MFCC chain on gfx950: framing, Hamming window,
// pre-emphasis on the WINDOWED frame, zero-padded real FFT, power spectrum, melfb.m-style
// filterbank, ln, DCT-II (c0 dropped), per-utterance CMVN, first-difference deltas.
// Restates src/feature/MFCC.py:14-16 (window), :20-41 (constants), :49-79 (chain),
// :81-105 (filterbank), :107-113 (DCT) and src/feature/utils.py:24-31 (deltas).
//
// Mapping: one wave64 per frame (4 frames per 256-thread workgroup in flight); the N/2-point
// complex Stockham FFT (radix 4, one trailing radix-2 pass when log2 is odd) ping-pongs
// between two LDS slabs private to the wave; twiddles live in LDS; the filterbank is a
// sparse row sweep with a wave64 shuffle reduction per band; constants are fp32 copies of
// float64 tables built on the host exactly as the reference builds them.
#include "batch.hpp"
#include "gmm_model.hpp"
#include "mfcc.hpp"
#include "mfcc_dev.hpp"
#include "wave_ops.hpp"

#include <algorithm>
#include <cmath>

namespace sr {

// ---------------- host: tables (float64, as the reference) ----------------

static std::vector<double> hamming(int n) {  // MFCC.py:14-16
    std::vector<double> w(n);
    for (int i = 0; i < n; i++) w[i] = 0.54 - 0.46 * std::cos(2 * M_PI / n * (i + 0.5));
    return w;
}

static std::vector<double> dct_rows(int n_bands, int n_ceps) {  // MFCC.py:107-113 + :36-37
    std::vector<double> d((size_t)n_ceps * n_bands);
    for (int y = 1; y <= n_ceps; y++)
        for (int x = 0; x < n_bands; x++)
            d[(size_t)(y - 1) * n_bands + x] =
                std::sqrt(2.0 / n_bands) * std::cos(M_PI * (2 * x + 1) * y / (2.0 * n_bands));
    return d;  // row 0 (the one divided by sqrt 2) is c0, which the reference drops
}

static std::vector<double> mel_bank(double fs, int fft_size, int n_bands) {  // MFCC.py:81-105
    const double f0 = 700.0 / fs;
    const int fn2 = fft_size / 2;
    const double lr = std::log(1 + 0.5 / f0) / (n_bands + 1);
    auto bl = [&](int i) { return fft_size * f0 * (std::exp(i * lr) - 1); };
    const int b1 = (int)std::floor(bl(0)) + 1;
    const int b2 = (int)std::ceil(bl(1));
    const int b3 = (int)std::floor(bl(n_bands));
    const int b4 = std::min(fn2, (int)std::ceil(bl(n_bands + 1))) - 1;
    const int n = b4 - b1 + 1;
    std::vector<double> fp(n), pm(n);
    for (int i = 0; i < n; i++) {
        const double pf = std::log(1 + (double)(b1 + i) / f0 / fft_size) / lr;
        fp[i] = std::floor(pf);
        pm[i] = pf - fp[i];
    }
    std::vector<double> M((size_t)n_bands * (fn2 + 1), 0.0);
    auto at = [&](int r, int c) -> double & {
        if (r < 0 || r >= n_bands || c < 0 || c > fn2) fail("mel filterbank index out of range");
        return M[(size_t)r * (fn2 + 1) + c];
    };
    for (int c = b2 - 1; c < b4; c++) at((int)fp[c] - 1, c + 1) += 2 * (1 - pm[c]);
    for (int c = 0; c < b3; c++) at((int)fp[c], c + 1) += 2 * pm[c];
    return M;
}

}  // namespace sr

SRMfcc::SRMfcc(double fs_, double win_length_ms, double win_shift_ms, int fft_size_,
               int n_filters_, int n_ceps_, double pre_emph_) {
    using namespace sr;
    fs = fs_;
    fft_size = fft_size_;
    n_filters = n_filters_;
    n_ceps = n_ceps_;
    pre_emph = pre_emph_;
    frame_len = (int)(win_length_ms / 1000.0 * fs);    // MFCC.py:28
    frame_shift = (int)(win_shift_ms / 1000.0 * fs);   // MFCC.py:29
    if (fft_size < 32 || fft_size > 4096 || (fft_size & (fft_size - 1)))
        fail("FFT_SIZE must be a power of two in [32, 4096], got %d", fft_size);
    if (frame_len <= 0 || frame_shift <= 0) fail("empty frame (len %d shift %d)", frame_len, frame_shift);
    if (frame_len > fft_size) fail("frame of %d samples does not fit FFT_SIZE %d", frame_len, fft_size);
    if (n_filters < 2 || n_filters > 64) fail("n_filters must be in [2, 64], got %d", n_filters);
    if (n_ceps < 1 || n_ceps >= n_filters) fail("n_ceps must be in [1, n_filters), got %d", n_ceps);
    window = hamming(frame_len);
    melbank = mel_bank(fs, fft_size, n_filters);
    dct = dct_rows(n_filters, n_ceps);
}

namespace sr {

bool mfcc_force_generic();
int mfcc_waves_per_block();

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// W_NFFT^i for i in [0, NFFT): the table stores the first half, the second is its negative.
__device__ __forceinline__ float2 tw(const float2 *t, int i, int nc) {
    float2 w = t[i & (nc - 1)];
    if (i & nc) w = make_float2(-w.x, -w.y);
    return w;
}

template <typename PcmT>
__global__ __launch_bounds__(256)
void mfcc_frames_kernel(const PcmT *__restrict__ pcm, const int64_t *__restrict__ sample_off,
                        const int64_t *__restrict__ frame_off, int n_utt, int64_t n_frames,
                        MfccDev p, float *__restrict__ raw /* [n_frames][n_ceps] */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nc = p.fft_size >> 1;           // complex FFT length
    float2 *s_tw = reinterpret_cast<float2 *>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float2 *buf_a = s_tw + nc + (size_t)wave * 2 * nc;
    float2 *buf_b = buf_a + nc;
    float *s_lm = reinterpret_cast<float *>(s_tw + nc + (size_t)4 * 2 * nc) + wave * 64;

    for (int i = threadIdx.x; i < nc; i += 256) s_tw[i] = p.twiddle[i];
    __syncthreads();

    const int64_t frames_per_iter = (int64_t)gridDim.x * 4;
    const int64_t iters = (n_frames + frames_per_iter - 1) / frames_per_iter;
    for (int64_t it = 0; it < iters; it++) {
        const int64_t frame = (it * gridDim.x + blockIdx.x) * 4 + wave;
        const bool active = frame < n_frames;   // wave-uniform

        // ---- frame -> (utterance, local index): binary search in frame_off ----
        int64_t base = 0;
        if (active) {
            int lo = 0, hi = n_utt;               // frame_off[lo] <= frame < frame_off[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (frame_off[mid] <= frame) lo = mid; else hi = mid;
            }
            base = sample_off[lo] + (frame - frame_off[lo]) * p.frame_shift;
        }

        // ---- window, then pre-emphasis on the windowed samples (MFCC.py:61-64); pack the
        // zero-padded real frame as z[n] = y[2n] + i y[2n+1] ----
        if (active) {
            const int L = p.frame_len;
            for (int n = lane; n < nc; n += 64) {
                float re = 0.f, im = 0.f;
                const int i0 = 2 * n;
                if (i0 < L) {
                    const float c0 = (float)pcm[base + i0] * p.window[i0];
                    const float pm1 = i0 > 0 ? (float)pcm[base + i0 - 1] * p.window[i0 - 1] : 0.f;
                    re = i0 > 0 ? c0 - pm1 * p.pre_emph : c0;
                    if (i0 + 1 < L) {
                        const float c1 = (float)pcm[base + i0 + 1] * p.window[i0 + 1];
                        im = c1 - c0 * p.pre_emph;
                    }
                }
                buf_a[n] = make_float2(re, im);
            }
        }
        __syncthreads();

        // ---- Stockham autosort FFT of length nc ----
        float2 *in = buf_a, *out = buf_b;
        int ns = 1;
        for (; ns * 4 <= nc; ns *= 4) {
            if (active) {
                const int T = nc >> 2;
                const int tstep = p.fft_size / (4 * ns) ;   // W_nc^(k*nc/(4ns)) = W_NFFT^(k*NFFT/(4ns))
                for (int j = lane; j < T; j += 64) {
                    const int k = j & (ns - 1);
                    const int q = k * tstep;
                    const float2 v0 = in[j];
                    const float2 v1 = cmul(in[j + T], tw(s_tw, q, nc));
                    const float2 v2 = cmul(in[j + 2 * T], tw(s_tw, 2 * q, nc));
                    const float2 v3 = cmul(in[j + 3 * T], tw(s_tw, 3 * q, nc));
                    const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
                    const float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
                    const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
                    const float2 a3 = make_float2(v1.y - v3.y, v3.x - v1.x);   // (v1-v3)*(-i)
                    const int j0 = ((j - k) << 2) + k;
                    out[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
                    out[j0 + ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
                    out[j0 + 2 * ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
                    out[j0 + 3 * ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
                }
            }
            __syncthreads();
            float2 *t = in; in = out; out = t;
        }
        if (ns < nc) {   // one radix-2 pass (ns == nc/2)
            if (active) {
                const int T = nc >> 1;
                for (int j = lane; j < T; j += 64) {
                    const float2 v0 = in[j];
                    const float2 v1 = cmul(in[j + T], tw(s_tw, 2 * j, nc));   // W_nc^j
                    out[j] = make_float2(v0.x + v1.x, v0.y + v1.y);
                    out[j + T] = make_float2(v0.x - v1.x, v0.y - v1.y);
                }
            }
            __syncthreads();
            float2 *t = in; in = out; out = t;
        }

        // ---- real-FFT untangle + power spectrum (MFCC.py:66), bins 0..nc into `out` as floats ----
        float *pw = reinterpret_cast<float *>(out);
        if (active) {
            for (int k = lane; k <= nc; k += 64) {
                const float2 zk = in[k & (nc - 1)];
                const float2 zr = in[(nc - k) & (nc - 1)];
                const float2 e = make_float2(0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y));
                const float2 o = make_float2(0.5f * (zk.y + zr.y), -0.5f * (zk.x - zr.x));
                const float2 w = tw(s_tw, k, nc);
                const float2 xo = cmul(w, o);
                const float xr = e.x + xo.x, xi = e.y + xo.y;
                pw[k] = xr * xr + xi * xi;
            }
        }
        __syncthreads();

        // ---- mel filterbank (sparse rows, wave shuffle reduction) + ln (MFCC.py:67-69) ----
        if (active) {
            float mine = 0.f;
            for (int b = 0; b < p.n_filters; b++) {
                float acc = 0.f;
                const int e1 = p.mel_row[b + 1];
                for (int e = p.mel_row[b] + lane; e < e1; e += 64) acc = fmaf(p.mel_val[e], pw[p.mel_col[e]], acc);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                if (lane == b) mine = acc > 0.f ? logf(acc) : p.mel_floor[b];
            }
            s_lm[lane] = mine;
        }
        __syncthreads();

        // ---- DCT-II rows 1..n_ceps ----
        if (active && lane < p.n_ceps) {
            float acc = 0.f;
            const float *drow = p.dct + lane * p.n_filters;
            for (int b = 0; b < p.n_filters; b++) acc = fmaf(drow[b], s_lm[b], acc);
            raw[frame * p.n_ceps + lane] = acc;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path for FFT_SIZE 2048 (the reference's default, MFCC.py:116): the 1024-point complex
// FFT of the packed frame is done by ONE wave with 16 points per lane in registers --
//   n = 64 n1 + lane : 16-point DFT over n1 in registers (pruned: a frame of <= 512 samples
//   fills only n1 < 4), twiddle W_1024^(lane k1), LDS transpose, radix-4 over a (n2 = 16a+b),
//   twiddle W_64^(bc), LDS transpose, 16-point DFT over b in registers  ->  Z[k1 + 16c + 64d]
// i.e. three register passes and two LDS exchanges instead of five LDS passes; the waves of a
// workgroup never synchronise with each other (wave-local LDS slab, wavefront-scope fences).
// Mel rows are contiguous column runs (melfb.m structure), lane = band; DCT lane = coefficient.

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

// y_r = sum_p x_p (-i)^(p r)
__device__ __forceinline__ void radix4(float2 a, float2 b, float2 c, float2 d, float2 &y0, float2 &y1,
                                       float2 &y2, float2 &y3) {
    const float2 s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = mul_mi(csub(b, d));
    y0 = cadd(s0, s2);
    y1 = cadd(s1, s3);
    y2 = csub(s0, s2);
    y3 = csub(s1, s3);
}

// In-register 16-point forward DFT, natural order in and out (4x4 Cooley-Tukey).
// NZ = number of leading nonzero inputs the caller guarantees (4 or 16).
template <int NZ>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, H = 0.70710678118654752440f;
    float2 u[4][4];   // u[q][r]
    if constexpr (NZ <= 4) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) u[q][r] = v[q];
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) radix4(v[q], v[4 + q], v[8 + q], v[12 + q], u[q][0], u[q][1], u[q][2], u[q][3]);
    }
    // twiddles W_16^(q r)
    u[1][1] = cmul(u[1][1], make_float2(C1, -S1));
    u[1][2] = cmul(u[1][2], make_float2(H, -H));
    u[1][3] = cmul(u[1][3], make_float2(S1, -C1));
    u[2][1] = cmul(u[2][1], make_float2(H, -H));
    u[2][2] = mul_mi(u[2][2]);
    u[2][3] = cmul(u[2][3], make_float2(-H, -H));
    u[3][1] = cmul(u[3][1], make_float2(S1, -C1));
    u[3][2] = cmul(u[3][2], make_float2(-H, -H));
    u[3][3] = cmul(u[3][3], make_float2(-C1, S1));
#pragma unroll
    for (int r = 0; r < 4; r++) radix4(u[0][r], u[1][r], u[2][r], u[3][r], v[r], v[r + 4], v[r + 8], v[r + 12]);
}

// In-register forward DFT of N1 = 16, 8 or 4 points, natural order in and out (pass 1 of the register-resident FFT:
// N1 = complex points / 64).  NZ = leading nonzero inputs the caller guarantees.
template <int N1, int NZ>
__device__ __forceinline__ void dft_n1(float2 (&v)[16]) {
    if constexpr (N1 == 16) {
        dft16<NZ>(v);
    } else if constexpr (N1 == 8) {
        constexpr float H = 0.70710678118654752440f;
        float2 e[4], o[4];
        radix4(v[0], v[2], v[4], v[6], e[0], e[1], e[2], e[3]);
        radix4(v[1], v[3], v[5], v[7], o[0], o[1], o[2], o[3]);
        o[1] = cmul(o[1], make_float2(H, -H));
        o[2] = mul_mi(o[2]);
        o[3] = cmul(o[3], make_float2(-H, -H));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = cadd(e[k], o[k]);
            v[k + 4] = csub(e[k], o[k]);
        }
    } else {
        static_assert(N1 == 4, "pass-1 DFT sizes: 16, 8, 4");
        float2 y0, y1, y2, y3;
        radix4(v[0], v[1], v[2], v[3], y0, y1, y2, y3);
        v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3;
    }
}

// One wave = one contiguous range of frames (binary search for the utterance once, then walk);
// per-lane constants (window taps, pass-1/2 twiddles) live in registers, the untangle twiddles
// and the mel / DCT tables in LDS; the next frame's samples are prefetched while the current
// frame is transformed.
// WPB = waves per workgroup: 4 (per-lane twiddle constants in registers, 2 waves/SIMD) or 12 (the
// pass-1 and untangle twiddles read from LDS instead: <= 168 VGPRs, 3 waves/SIMD; one workgroup per
// CU, its tables shared by 12 waves).
// N1 = complex points / 64: 16 (FFT_SIZE 2048, the reference's default), 8 (1024) or 4 (512).  The transform is
// n = 64 n1 + lane -> N1-point DFT over n1 -> twiddle -> 64-point DFT over the lane index as 4 x 16 through two
// LDS exchanges; with N1 < 16 pass 2 has N1/4 rows per lane and pass 3 runs on the first 4 N1 lanes.
template <typename PcmT, int NZ1, int MP, int WPB, int N1 = 16>
__global__ __launch_bounds__(64 * WPB, WPB == 4 ? 2 : 3)
void mfcc_frames_fft2048_kernel(const PcmT *__restrict__ pcm, const int64_t *__restrict__ sample_off,
                                const int64_t *__restrict__ frame_off, int n_utt, int64_t n_frames,
                                int64_t frames_per_wave, MfccDev p, MelRuns mr, float *__restrict__ raw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NC = 64 * N1;                                         // complex points; FFT_SIZE = 2 NC
    static_assert(N1 == 16 || N1 == 8 || N1 == 4, "FFT_SIZE 2048, 1024 or 512");
    static_assert(NZ1 <= N1 && (MP == 0 || N1 == 16), "the mel presets belong to FFT_SIZE 2048");
    float2 *s_tw = reinterpret_cast<float2 *>(smem);                    // W_{2 NC}^k, k < NC
    float *s_melval = reinterpret_cast<float *>(s_tw + NC);
    float *s_dct = s_melval + mr.pad_floats;                         // [16][DCT_LD], zero padded
    constexpr int DCT_LD = MFCC_DCT_LD;
    const int dct_pad = 16 * DCT_LD;
    // Frames of more than 4 rows of samples (NZ1 > 4: windows longer than 512 samples) hold 4 NZ1 window taps, 3 NZ1 sample
    // offsets and 3 NZ1 prefetched samples per lane on top of the transform's 32 registers: through round 3 those variants
    // spilled 460-680 bytes per lane.  They keep the twiddle constants AND the window taps in LDS (one 16-byte read per row and
    // frame), form the offsets where they are used, and come in the 4-wave shape only (whose LDS has the room).
    constexpr bool LDS_WIN = NZ1 > 4;
    static_assert(!LDS_WIN || WPB == 4, "long frames: the 4-wave workgroup shape only");
    constexpr bool LDS_TW = WPB != 4 || LDS_WIN;                     // twiddle constants in LDS, not registers
    float2 *s_wk = reinterpret_cast<float2 *>(s_dct + dct_pad);      // [N1][64] W_NC^(lane*k1) (LDS_TW only)
    float4 *s_win = reinterpret_cast<float4 *>(s_wk + (LDS_TW ? N1 * 64 : 0));     // [NZ1][64] {w0, wm, w1, w0p} (LDS_WIN only)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float2 *slab = reinterpret_cast<float2 *>(s_win + (LDS_WIN ? NZ1 * 64 : 0)) + (size_t)wave * WAVE_SLAB_C;
    float *pbuf = reinterpret_cast<float *>(slab);          // power spectrum, 1025 floats
    float *s_lm = pbuf + 1100;                              // log-mel energies, 64 floats

    for (int i = threadIdx.x; i < NC; i += 64 * WPB) s_tw[i] = p.twiddle[i];
    for (int i = threadIdx.x; i < mr.pad_floats; i += 64 * WPB) s_melval[i] = mr.pad_val[i];
    for (int i = threadIdx.x; i < 16 * DCT_LD; i += 64 * WPB) {
        const int c = i / DCT_LD, b = i - c * DCT_LD;
        s_dct[i] = (c < p.n_ceps && b < p.n_filters) ? p.dct[c * p.n_filters + b] : 0.f;
    }
    __syncthreads();
    if (LDS_TW) {
        for (int i = threadIdx.x; i < N1 * 64; i += 64 * WPB) s_wk[i] = tw(s_tw, (2 * (i & 63) * (i >> 6)) & (2 * NC - 1), NC);
        __syncthreads();
    }

    // ---- per-lane constants ----
    float2 wk[LDS_TW ? 1 : N1];        // W_NC^(lane*k1)
    if (!LDS_TW) {
#pragma unroll
        for (int k1 = 0; k1 < N1; k1++) wk[k1] = tw(s_tw, (2 * lane * k1) & (2 * NC - 1), NC);
    }
    const int bb = lane & 15, gg = lane >> 4;
    float2 w64[4];        // W_64^(b*c)
#pragma unroll
    for (int c = 0; c < 4; c++) w64[c] = tw(s_tw, (2 * N1 * bb * c) & (2 * NC - 1), NC);
    float2 utw[LDS_TW ? 1 : N1];       // untangle twiddles W_{2 NC}^(lane + 64 j)
    if (!LDS_TW) {
#pragma unroll
        for (int j = 0; j < N1; j++) utw[j] = s_tw[lane + 64 * j];
    }
    const int L = p.frame_len;
    // window taps of this lane's samples: y[i0] = w0 x[i0] - wm x[i0-1], y[i0+1] = w1 x[i0+1] - w0p x[i0]
    constexpr int NW = LDS_WIN ? 1 : NZ1;
    float win_m[NW], win_0[NW], win_1[NW], win_0p[NW];
    auto taps = [&](int i0, float &w0, float &wm, float &w1, float &w0p) {
        w0 = i0 < L ? p.window[i0] : 0.f;
        wm = (i0 > 0 && i0 < L) ? p.window[i0 - 1] * p.pre_emph : 0.f;
        w1 = i0 + 1 < L ? p.window[i0 + 1] : 0.f;
        w0p = i0 + 1 < L ? w0 * p.pre_emph : 0.f;
    };
    if constexpr (LDS_WIN) {
        for (int i = threadIdx.x; i < NZ1 * 64; i += 64 * WPB) {
            float w0, wm, w1, w0p;
            taps(2 * i, w0, wm, w1, w0p);                   // (i = 64 n1 + lane)
            s_win[i] = make_float4(w0, wm, w1, w0p);
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int n1 = 0; n1 < NZ1; n1++) taps(2 * (64 * n1 + lane), win_0[n1], win_m[n1], win_1[n1], win_0p[n1]);
    }
    // mel: 4 lanes per band, 16 bands per pass
    const int m_part = lane & 3, m_bl = lane >> 2;
    int m_c0[4];
    float m_floor[4];
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        const int band = 16 * ps + m_bl;
        m_c0[ps] = mr.col0[band];
        m_floor[ps] = band < p.n_filters ? p.mel_floor[band] : 0.f;
    }

    const int64_t gwave = (int64_t)blockIdx.x * WPB + wave;
    const int64_t f_begin = gwave * frames_per_wave;
    const int64_t f_end = f_begin + frames_per_wave < n_frames ? f_begin + frames_per_wave : n_frames;
    if (f_begin >= f_end) return;      // whole wave idle (no workgroup barrier below this point)

    int utt = 0;
    {
        int lo = 0, hi = n_utt;        // frame_off[lo] <= f_begin < frame_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (frame_off[mid] <= f_begin) lo = mid; else hi = mid;
        }
        utt = lo;
    }
    int64_t utt_f0 = frame_off[utt], utt_f1 = frame_off[utt + 1], utt_s0 = sample_off[utt];

    // Sample fetch for one frame: three taps per row (previous, even, odd sample).  Addresses are
    // clamped into the frame, never predicated: out-of-frame taps carry a zero window weight, and
    // all loads of a frame are in flight together (a predicated load forces a wait each).
    int off_m[NW], off_0[NW], off_1[NW];
    auto offsets = [&](int n1, int &o0, int &om, int &o1) {
        const int i0 = 2 * (64 * n1 + lane);
        o0 = i0 < L ? i0 : L - 1;
        om = i0 - 1 < 0 ? 0 : (i0 - 1 < L ? i0 - 1 : L - 1);
        o1 = i0 + 1 < L ? i0 + 1 : L - 1;
    };
    if constexpr (!LDS_WIN) {
#pragma unroll
        for (int n1 = 0; n1 < NZ1; n1++) offsets(n1, off_0[n1], off_m[n1], off_1[n1]);
    }
    // (held as float: a 32-bit register per tap, so the prefetched samples are not re-packed --
    // packing would put a wait right behind the loads instead of one iteration later)
    auto fetch = [&](int64_t base, float (&xm)[NZ1], float (&x0)[NZ1], float (&x1)[NZ1]) {
        const PcmT *fp = pcm + base;
#pragma unroll
        for (int n1 = 0; n1 < NZ1; n1++) {
            int o0, om, o1;
            if constexpr (LDS_WIN) offsets(n1, o0, om, o1);
            else {
                o0 = off_0[n1];
                om = off_m[n1];
                o1 = off_1[n1];
            }
            x0[n1] = (float)fp[o0];
            xm[n1] = (float)fp[om];
            x1[n1] = (float)fp[o1];
        }
    };
    float cm[NZ1], c0[NZ1], c1[NZ1];
    fetch(utt_s0 + (f_begin - utt_f0) * p.frame_shift, cm, c0, c1);

    for (int64_t frame = f_begin; frame < f_end; frame++) {
        if constexpr (LDS_WIN) {
            // long frames: 3 x 16 prefetched samples live across the whole transform are what used to spill; they are
            // fetched where they are consumed instead (the wave beside this one on the SIMD covers the latency)
            if (frame != f_begin) {
                while (frame >= utt_f1) {
                    utt++;
                    utt_f0 = utt_f1;
                    utt_f1 = frame_off[utt + 1];
                    utt_s0 = sample_off[utt];
                }
                fetch(utt_s0 + (frame - utt_f0) * p.frame_shift, cm, c0, c1);
            }
        }
        // ---- window + pre-emphasis on the windowed samples (MFCC.py:61-64), packed z = y[2n] + i y[2n+1] ----
        float2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) v[n1] = make_float2(0.f, 0.f);
#pragma unroll
        for (int n1 = 0; n1 < NZ1; n1++) {
            float w0, wm, w1, w0p;
            if constexpr (LDS_WIN) {
                const float4 t = s_win[n1 * 64 + lane];
                w0 = t.x; wm = t.y; w1 = t.z; w0p = t.w;
            } else {
                w0 = win_0[n1]; wm = win_m[n1]; w1 = win_1[n1]; w0p = win_0p[n1];
            }
            v[n1] = make_float2(w0 * (float)c0[n1] - wm * (float)cm[n1], w1 * (float)c1[n1] - w0p * (float)c0[n1]);
        }
        // ---- prefetch the next frame's samples (in flight during the transform) ----
        if (!LDS_WIN && frame + 1 < f_end) {
            int64_t nf = frame + 1;
            while (nf >= utt_f1) {     // utterances with zero frames are skipped
                utt++;
                utt_f0 = utt_f1;
                utt_f1 = frame_off[utt + 1];
                utt_s0 = sample_off[utt];
            }
            fetch(utt_s0 + (nf - utt_f0) * p.frame_shift, cm, c0, c1);
        }
        // ---- pass 1: N1-point DFT over n1, twiddle, exchange ----
        dft_n1<N1, (NZ1 <= 4 ? 4 : 16)>(v);
#pragma unroll
        for (int k1 = 1; k1 < N1; k1++) v[k1] = cmul(v[k1], LDS_TW ? s_wk[k1 * 64 + lane] : wk[k1]);
        wave_sync();      // previous frame's readers are done with the slab
#pragma unroll
        for (int k1 = 0; k1 < N1; k1++) slab[k1 * 68 + lane] = v[k1];
        wave_sync();
        // ---- pass 2: radix-4 over a for (k1 = (N1/4) g + i, b), twiddle W_64^(bc), exchange ----
#pragma unroll
        for (int i = 0; i < N1 / 4; i++) {
            const float2 *rowp = slab + ((N1 / 4) * gg + i) * 68 + bb;
            radix4(rowp[0], rowp[16], rowp[32], rowp[48], v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            v[4 * i + 1] = cmul(v[4 * i + 1], w64[1]);
            v[4 * i + 2] = cmul(v[4 * i + 2], w64[2]);
            v[4 * i + 3] = cmul(v[4 * i + 3], w64[3]);
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < N1 / 4; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) slab[(((N1 / 4) * gg + i) + N1 * c) * 17 + bb] = v[4 * i + c];
        wave_sync();
        // ---- pass 3: lane l = k1 + N1 c (the first 4 N1 lanes) holds C[k1][b][c], b = 0..15 -> Z[l + 4 N1 d] ----
        if (N1 == 16 || lane < 4 * N1) {
#pragma unroll
            for (int b = 0; b < 16; b++) v[b] = slab[lane * 17 + b];
            dft16<16>(v);
        }
        wave_sync();
        if (N1 == 16 || lane < 4 * N1) {
#pragma unroll
            for (int d = 0; d < 16; d++) slab[lane + 4 * N1 * d] = v[d];
        }
        wave_sync();
        // ---- real-FFT untangle + power spectrum (MFCC.py:66): bins lane + 64 j, and bin NC ----
        float pw[N1];
#pragma unroll
        for (int j = 0; j < N1; j++) {
            const int k = lane + 64 * j;
            float2 zk;
            if constexpr (N1 == 16) zk = v[j]; else zk = slab[k];   // N1 = 16: Z[lane + 64 j] is this lane's own output
            const float2 zr = slab[(NC - k) & (NC - 1)];
            // 2 X[k] = (Zk + conj Zr) - i W^k (Zk - conj Zr): the factor 1/2 is left out here and the
            // resulting 4x in the power is folded into the mel weights on the host (x 0.25)
            const float ex = zk.x + zr.x, ey = zk.y - zr.y;
            const float ox = zk.y + zr.y, oy = zr.x - zk.x;
            const float2 ut = LDS_TW ? s_tw[lane + 64 * j] : utw[j];
            const float xr = fmaf(-ut.y, oy, fmaf(ut.x, ox, ex));
            const float xi = fmaf(ut.y, ox, fmaf(ut.x, oy, ey));
            pw[j] = fmaf(xr, xr, xi * xi);
        }
        const float2 z0 = slab[0];
        const float nyq = 4.0f * (z0.x - z0.y) * (z0.x - z0.y);   // X[NC] = Re Z0 - Im Z0 (x4: same scale as the other bins)
        wave_sync();
#pragma unroll
        for (int j = 0; j < N1; j++) pbuf[lane + 64 * j] = pw[j];
        if (lane == 0) pbuf[NC] = nyq;
        wave_sync();
        // ---- mel filterbank (MFCC.py:67-69): 4 lanes sweep one band's column run, 16 bands per pass ----
        float lm[4];
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const int len = mr.pass_len[ps];
            // one ds_read_b128 of weights and one of the power spectrum per 4 FMAs: lane (band, part)
            // takes bins c0 + 16 it + 4 part + {0..3}; c0 is a multiple of 4, so both are 16-byte aligned
            const float4 *mv4 = reinterpret_cast<const float4 *>(s_melval + mr.pass_base[ps]) + lane;
            const float4 *pp4 = reinterpret_cast<const float4 *>(pbuf + m_c0[ps]) + m_part;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            if constexpr (MP != 0) {
#pragma unroll
                for (int it = 0; it < mel_preset_steps(MP, ps); it++) {
                    const float4 wv = mv4[it * 64];
                    const float4 xv = pp4[it * 4];
                    a0 = fmaf(wv.x, xv.x, a0);
                    a1 = fmaf(wv.y, xv.y, a1);
                    a2 = fmaf(wv.z, xv.z, a2);
                    a3 = fmaf(wv.w, xv.w, a3);
                }
            } else {
                for (int it = 0; it < (len >> 4); it++) {     // zero-padded runs: no bounds logic in the loop
                    const float4 wv = mv4[it * 64];
                    const float4 xv = pp4[it * 4];
                    a0 = fmaf(wv.x, xv.x, a0);
                    a1 = fmaf(wv.y, xv.y, a1);
                    a2 = fmaf(wv.z, xv.z, a2);
                    a3 = fmaf(wv.w, xv.w, a3);
                }
            }
            float acc = (a0 + a1) + (a2 + a3);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            lm[ps] = acc > 0.f ? logf(acc) : m_floor[ps];
        }
        wave_sync();
        if (m_part == 0) {
#pragma unroll
            for (int ps = 0; ps < 4; ps++) s_lm[16 * ps + m_bl] = lm[ps];
        }
        wave_sync();
        // ---- DCT-II rows 1..n_ceps: 4 lanes per coefficient, zero-padded table [16][64] ----
        {
            const int cidx = lane >> 2;
            // lane (coefficient, part) takes bands 16 it + 4 part + {0..3}: one 16-byte read of the row
            // and one of the log-mel vector per 4 FMAs
            const float4 *drow = reinterpret_cast<const float4 *>(s_dct + cidx * DCT_LD) + m_part;
            const float4 *lp = reinterpret_cast<const float4 *>(s_lm) + m_part;
            float o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const float4 dv = drow[it * 4];
                const float4 lv = lp[it * 4];
                o0 = fmaf(dv.x, lv.x, o0);
                o1 = fmaf(dv.y, lv.y, o1);
                o0 = fmaf(dv.z, lv.z, o0);
                o1 = fmaf(dv.w, lv.w, o1);
            }
            float o = o0 + o1;
            o += __shfl_xor(o, 1, 64);
            o += __shfl_xor(o, 2, 64);
            if (m_part == 0 && cidx < p.n_ceps) raw[frame * p.n_ceps + cidx] = o;
        }
    }
}

// Per-utterance CMVN (mean, population std, no epsilon -- MFCC.py:74-77) and causal
// first-difference deltas AFTER the normalisation (utils.py:24-31).  Statistics in float64.
constexpr int CMVN_THREADS = 1024;    // frames of an utterance go round-robin to CMVN_THREADS / 16 stripes (13 coefficients): one short utterance
                                      // -- a serving decision -- is 5 frames per thread instead of 19 with 256 (16 -> 6 us)
__global__ __launch_bounds__(CMVN_THREADS)
void cmvn_delta_kernel(const float *__restrict__ raw, const int64_t *__restrict__ raw_off,
                       const int64_t *__restrict__ out_off, int n_ceps, int nd, int cmvn,
                       float *__restrict__ out, int out_stride) {
    const int u = blockIdx.x;
    const int64_t r0 = raw_off[u];
    const int64_t T = raw_off[u + 1] - r0;
    const int64_t o0 = out_off[u];
    const int64_t To = out_off[u + 1] - o0;
    __shared__ double red[CMVN_THREADS];
    __shared__ double s_mean[64], s_inv[64];
    const int tid = threadIdx.x;
    // thread = (stripe of frames, coefficient): consecutive threads read consecutive floats
    const int cp = n_ceps <= 16 ? 16 : n_ceps <= 32 ? 32 : 64;
    const int n_stripes = CMVN_THREADS / cp;
    const int c = tid & (cp - 1);
    const int stripe = tid / cp;
    const bool live = c < n_ceps;
    if (tid < 64) {
        s_mean[tid] = 0.0;
        s_inv[tid] = 1.0;
    }
    __syncthreads();
    if (cmvn && T > 1) {
        double acc = 0.0;
        if (live)
            for (int64_t t = stripe; t < T; t += n_stripes) acc += (double)raw[(r0 + t) * n_ceps + c];
        red[tid] = acc;
        __syncthreads();
        if (tid < cp) {
            double tot = 0.0;
            for (int s = 0; s < n_stripes; s++) tot += red[s * cp + tid];
            s_mean[tid] = tot / (double)T;
        }
        __syncthreads();
        acc = 0.0;
        if (live) {
            const double mu = s_mean[c];
            for (int64_t t = stripe; t < T; t += n_stripes) {
                const double dv = (double)raw[(r0 + t) * n_ceps + c] - mu;
                acc += dv * dv;
            }
        }
        red[tid] = acc;
        __syncthreads();
        if (tid < cp) {
            double tot = 0.0;
            for (int s = 0; s < n_stripes; s++) tot += red[s * cp + tid];
            s_inv[tid] = 1.0 / sqrt(tot / (double)T);      // no epsilon, as the reference
        }
        __syncthreads();
    }
    if (live) {
        const double mu = s_mean[c], inv = s_inv[c];
        for (int64_t t = stripe; t < To; t += n_stripes) {
            const int64_t tr = t + nd;   // row of the normalised features this output row ends on
            float *dst = out + (o0 + t) * out_stride;
            const double z0 = ((double)raw[(r0 + tr) * n_ceps + c] - mu) * inv;
            dst[c] = (float)z0;
            if (nd >= 1) {
                const double z1 = ((double)raw[(r0 + tr - 1) * n_ceps + c] - mu) * inv;
                dst[n_ceps + c] = (float)(z0 - z1);
                if (nd >= 2) {
                    const double z2 = ((double)raw[(r0 + tr - 2) * n_ceps + c] - mu) * inv;
                    dst[2 * n_ceps + c] = (float)((z0 - z1) - (z1 - z2));
                }
            }
        }
    }
}

// ---------------- host: device tables + launch ----------------

MfccDev upload_tables(SRMfcc &m) {
    std::shared_ptr<void> &slot = m.dev[current_device()];
    if (!slot) {
        auto t = std::make_shared<MfccDeviceTables>();
        const int L = m.frame_len, NF = m.fft_size, nc = NF / 2, B = m.n_filters, C = m.n_ceps;
        std::vector<float> w(L), dctf((size_t)C * B), val, floor_ln(B);
        std::vector<float2> twd(nc);
        std::vector<int> row(B + 1, 0), col, col0(64, 0), cnt(64, 0);
        for (int i = 0; i < L; i++) w[i] = (float)m.window[i];
        for (size_t i = 0; i < dctf.size(); i++) dctf[i] = (float)m.dct[i];
        for (int k = 0; k < nc; k++) {
            const double ang = -2.0 * M_PI * k / NF;
            twd[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
        for (int b = 0; b < B; b++) {
            double rs = 0.0;
            for (int c = 0; c <= nc; c++) {
                const double v = m.melbank[(size_t)b * (nc + 1) + c];
                if (v != 0.0) {
                    col.push_back(c);
                    val.push_back((float)v);
                    rs += v;
                }
            }
            row[b + 1] = (int)col.size();
            cnt[b] = row[b + 1] - row[b];
            col0[b] = cnt[b] ? col[row[b]] : 0;
            if (cnt[b] && col[row[b + 1] - 1] - col0[b] + 1 != cnt[b]) t->runs_contiguous = false;
            t->max_cnt = std::max(t->max_cnt, cnt[b]);
            t->pass_len[b / 16] = std::max(t->pass_len[b / 16], ((cnt[b] + 15) / 16) * 16);
            floor_ln[b] = (float)std::log(1e-100 * rs);   // POWER_SPECTRUM_FLOOR, MFCC.py:8,67
        }
        if (col.empty()) fail("empty mel filterbank");
        t->window.upload(w.data(), w.size());
        t->dct.upload(dctf.data(), dctf.size());
        t->twiddle.upload(twd.data(), twd.size());
        t->mel_row.upload(row.data(), row.size());
        t->mel_col.upload(col.data(), col.size());
        t->mel_val.upload(val.data(), val.size());
        t->mel_floor.upload(floor_ln.data(), floor_ln.size());
        {   // float64 copies for the float64-spectrum kernels (mfcc_f64.hip)
            std::vector<double2> twd64(nc);
            for (int k = 0; k < nc; k++) {
                const double ang = -2.0 * M_PI * k / NF;
                twd64[k] = make_double2(std::cos(ang), std::sin(ang));
            }
            std::vector<double> val64, floor64(64, 0.0), dpad((size_t)4 * 64 * 4, 0.0);
            for (int b = 0; b < B; b++) {
                double rs = 0.0;
                for (int c = 0; c <= nc; c++) {
                    const double v = m.melbank[(size_t)b * (nc + 1) + c];
                    if (v != 0.0) {
                        val64.push_back(v);
                        rs += v;
                    }
                }
                floor64[b] = std::log(1e-100 * rs);
            }
            if (C <= 16)
                for (int it = 0; it < 4; it++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int q = 0; q < 4; q++) {
                            const int ci = lane >> 2, band = 16 * it + 4 * (lane & 3) + q;
                            if (ci < C && band < B) dpad[((size_t)it * 64 + lane) * 4 + q] = m.dct[(size_t)ci * B + band];
                        }
            t->window64.upload(m.window.data(), m.window.size());
            t->twiddle64.upload(twd64.data(), twd64.size());
            t->mel_val64.upload(val64.data(), val64.size());
            t->mel_floor64.upload(floor64.data(), floor64.size());
            t->dct64.upload(m.dct.data(), m.dct.size());
            t->dct_pad64.upload(dpad.data(), dpad.size());
        }
        // Padded re-layout for the fast kernels: pass ps holds bands 16ps..16ps+15, four lanes sweep a band with one
        // ds_read_b128 each per step.  That instruction is served in four groups of 16 lanes -- {0-3,12-15,20-27},
        // {4-11,16-19,28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- over 16 slots of 16 bytes (bank = dword address mod
        // 64), i.e. the bands {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15} of a pass are served together, each covering the four
        // consecutive slots from (start / 4) mod 16.  A band's sweep start may move DOWN in steps of 4 columns (leading zero
        // weights) as long as its padded run still fits the pass's length: every group's starts are chosen -- exhaustively, a
        // few thousand candidates, once per extractor -- for the fewest extra LDS cycles, then the least padding.  (Through
        // round 5 the starts avoided conflicts of a 32-lane / 8-window model that is not this instruction's: 4-6 extra cycles
        // per read in the first three passes of the 16 kHz bank, SQ_LDS_BANK_CONFLICT 7 % of the kernel's LDS cycles; now 2.)
        std::vector<int> start(64, 0), lead(64, 0);
        mel_sweep_starts(col0.data(), cnt.data(), B, t->pass_len, start.data());       // (gmm_model.cpp: host-only, under the sanitizers in tests/host)
        for (int b = 0; b < B; b++) lead[b] = cnt[b] ? col0[b] - start[b] : 0;
        for (int ps = 0; ps < 4; ps++) t->pass_len[ps] = 0;
        for (int b = 0; b < B; b++)
            t->pass_len[b / 16] = std::max(t->pass_len[b / 16], ((lead[b] + cnt[b] + 15) / 16) * 16);
        int total = 0;
        for (int ps = 0; ps < 4; ps++) {
            t->pass_base[ps] = total;
            total += 16 * t->pass_len[ps];
        }
        t->pad_floats = (total + 3) & ~3;
        std::vector<float> padv((size_t)std::max(4, t->pad_floats), 0.0f);
        for (int b = 0; b < B; b++) {
            const int ps = b / 16, bl = b % 16;
            for (int i = 0; i < cnt[b]; i++) {
                const int e = lead[b] + i;
                padv[(size_t)t->pass_base[ps] + (size_t)(e >> 4) * 256 + ((size_t)bl * 4 + ((e >> 2) & 3)) * 4 + (e & 3)] = 0.25f * val[row[b] + i];   // the fast kernel's power spectrum is 4 |X|^2
            }
            if (start[b] + t->pass_len[ps] + 3 > 1100) t->runs_contiguous = false;   // padded sweep must stay inside the slab's power-spectrum region
        }
        for (int b = 0; b < 64; b++) col0[b] = start[b];   // the kernel sweeps from the aligned start
        t->mel_pad.upload(padv.data(), padv.size());
        t->mel_col0.upload(col0.data(), col0.size());
        t->mel_cnt.upload(cnt.data(), cnt.size());
        t->nnz = (int)col.size();
        sync_stream();
        slot = t;
    }
    auto &t = *std::static_pointer_cast<MfccDeviceTables>(slot);
    MfccDev d;
    d.window = t.window.p;
    d.twiddle = t.twiddle.p;
    d.mel_row = t.mel_row.p;
    d.mel_col = t.mel_col.p;
    d.mel_val = t.mel_val.p;
    d.mel_floor = t.mel_floor.p;
    d.dct = t.dct.p;
    d.frame_len = m.frame_len;
    d.frame_shift = m.frame_shift;
    d.fft_size = m.fft_size;
    d.n_filters = m.n_filters;
    d.n_ceps = m.n_ceps;
    d.pre_emph = (float)m.pre_emph;
    return d;
}

MfccDev64 device_tables_f64(SRMfcc &m) {
    auto &t = device_tables(m);
    MfccDev64 d;
    d.window = t.window64.p;
    d.twiddle = t.twiddle64.p;
    d.mel_val = t.mel_val64.p;
    d.mel_floor = t.mel_floor64.p;
    d.dct = t.dct64.p;
    d.dct_pad = t.dct_pad64.p;
    d.pre_emph = m.pre_emph;
    return d;
}

int64_t mfcc_num_frames(const SRMfcc &m, int64_t n_samples) {
    if (n_samples <= 5 * (int64_t)m.frame_len) return 0;           // MFCC.py:56 (assert there)
    return (n_samples - m.frame_len) / m.frame_shift + 1;          // MFCC.py:57
}

static bool &mfcc_force_generic_flag() {
    static bool f = false;
    return f;
}
bool mfcc_force_generic() { return mfcc_force_generic_flag(); }
// (12-wave workgroups of the fp32 FFT-2048 kernel: the A/B against 4-wave ones was settled in round 2 and its switch went in round 6)
int mfcc_waves_per_block() { return 12; }
void mfcc_set_force_generic(bool on) { mfcc_force_generic_flag() = on; }

struct MfccScratch {
    DevBuf<float> raw;
    DevBuf<int64_t> raw_off;
    std::vector<int64_t> raw_off_host;   // what raw_off currently holds (skip the upload + sync when unchanged)
    StagedUpload<int64_t> stage_raw_off; // (a workspace lives as long as its device: its table always travels this way when small)
};
MfccScratch *mfcc_scratch_new() { return new MfccScratch(); }
void mfcc_scratch_delete(MfccScratch *s) { delete s; }

// PCM batch -> feature batch.  `out` is reused when it is large enough (serving loop).
void mfcc_extract_batch(SRMfcc &m, SRBatch &pcm, int nd, int cmvn, SRBatch &out) {
    mfcc_extract_with(m, pcm, nd, cmvn, out, &per_device<MfccScratch>());     // (the device's own: leaked on purpose, no hipFree at exit)
}

void mfcc_extract_with(SRMfcc &m, SRBatch &pcm, int nd, int cmvn, SRBatch &out, MfccScratch *scratch) {
    ensure_device();
    if (!scratch) fail("null feature workspace");
    const int u0 = 0, u1 = pcm.n_utt;
    if (pcm.kind != SRBatch::PCM16 && pcm.kind != SRBatch::PCMF32) fail("MFCC needs a PCM batch");
    pcm.bind_device();
    out.bind_device();
    if (nd < 0 || nd > 2) fail("delta order must be 0, 1 or 2");
    if (m.n_lpc > 0 && nd != 0) fail("LPC columns (mix_feature) come without deltas: use nd = 0");
    const MfccDev dev = upload_tables(m);
    const int U = u1 - u0;
    if (m.n_lpc > 0 && (u0 != 0 || u1 != pcm.n_utt)) fail("LPC columns need the whole batch");
    const int64_t *d_pcm_off = pcm.d_offsets.p + u0;      // offsets stay absolute into the PCM buffer
    std::vector<int64_t> raw_off(U + 1, 0), out_off(U + 1, 0);
    for (int u = 0; u < U; u++) {
        const int64_t T = mfcc_num_frames(m, pcm.offsets[u0 + u + 1] - pcm.offsets[u0 + u]);
        raw_off[u + 1] = raw_off[u] + T;
        out_off[u + 1] = out_off[u] + std::max<int64_t>(0, T - nd);
    }
    const int64_t NF = raw_off[U];
    auto &w = *scratch;
    w.raw.ensure((size_t)std::max<int64_t>(1, NF) * m.n_ceps);
    bool uploaded = false;
    const bool small_tables = raw_off.size() * sizeof(int64_t) <= STAGED_TABLE_MAX_BYTES;
    if (w.raw_off_host != raw_off) {
        w.raw_off_host = raw_off;
        if (small_tables) {
            w.stage_raw_off.send(w.raw_off, w.raw_off_host.data(), w.raw_off_host.size());
        } else {
            w.raw_off.upload(w.raw_off_host.data(), w.raw_off_host.size());
            uploaded = true;
        }
    }

    const int dim_out = m.n_ceps * (nd + 1) + m.n_lpc;
    const bool same_shape = out.kind == SRBatch::FEATURES && out.offsets == out_off && out.dim == dim_out;
    out.kind = SRBatch::FEATURES;
    out.n_utt = U;
    out.dim = dim_out;
    out.n_rows = out_off[U];
    out.data.ensure((size_t)std::max<int64_t>(1, out.n_rows) * out.dim);
    if (!same_shape) {
        const bool refilled = !out.offsets.empty() && out.d_offsets.p != nullptr;      // (a batch this stage has filled before)
        out.offsets = out_off;
        out.invalidate_tiles();
        if (refilled && small_tables) {
            out.stage_offsets.send(out.d_offsets, out.offsets.data(), out.offsets.size());
        } else {
            out.d_offsets.upload(out.offsets.data(), out.offsets.size());
            uploaded = true;
        }
    }

    if (NF > 0) {
        auto &tabs = *std::static_pointer_cast<MfccDeviceTables>(m.dev[current_device()]);
        // the register-resident kernel: FFT_SIZE 2048 (the reference's default), 1024 or 512, frames that fit the transform
        const int n1 = m.fft_size / 128;               // complex points / 64
        const bool fast = (m.fft_size == 2048 || m.fft_size == 1024 || m.fft_size == 512) && m.frame_len <= m.fft_size &&
                          tabs.runs_contiguous && m.n_ceps <= 16 && !mfcc_force_generic();
        ScopedKernelTimer t(T_MFCC);
        if (mfcc_precision() == 2) {
            // float64 spectrum for every frame (MFCC.py:59-70 computes in float64): mfcc_f64.hip
            mfcc_launch_f64(m, dev, pcm.kind, pcm.kind == SRBatch::PCM16 ? (const void *)pcm.pcm16.p : (const void *)pcm.data.p, d_pcm_off,
                            w.raw_off.p, U, NF, w.raw.p);
        } else if (fast) {
            MelRuns mr;
            mr.col0 = tabs.mel_col0.p;
            mr.pad_val = tabs.mel_pad.p;
            mr.pad_floats = tabs.pad_floats;
            for (int ps = 0; ps < 4; ps++) {
                mr.pass_base[ps] = tabs.pass_base[ps];
                mr.pass_len[ps] = tabs.pass_len[ps];
            }
            const int nz1 = (m.frame_len + 127) / 128;      // rows n1 with any nonzero sample
            const int nz_inst = nz1 <= 4 ? 4 : n1;          // the instantiated NZ1
            const bool long_frames = nz_inst > 4;           // window taps + twiddles in LDS, 4-wave workgroups only (see the kernel)
            auto lds_for = [&](int w) {
                return (size_t)(64 * n1) * sizeof(float2) + (size_t)(tabs.pad_floats + 16 * MFCC_DCT_LD) * sizeof(float) +
                       ((w == 4 && !long_frames) ? 0 : (size_t)n1 * 64 * sizeof(float2)) +
                       (long_frames ? (size_t)nz_inst * 64 * sizeof(float4) : 0) + (size_t)w * WAVE_SLAB_C * sizeof(float2);
            };
            int wpb = long_frames ? 4 : mfcc_waves_per_block();
            if (wpb == 12 && lds_for(12) > (size_t)160 * 1024) wpb = 4;      // a very wide filterbank: tables too big for one 12-wave workgroup
            const size_t lds = lds_for(wpb);
            // one contiguous frame range per wave; enough waves to fill the chip a few times over
            const int blocks_per_cu = std::max<int>(1, std::min<int>(3, (int)(160 * 1024 / lds)));
            const int64_t max_waves = (int64_t)ctx().n_cu * blocks_per_cu * wpb * 4;
            // A wave walks its frames one after the other.  Large batches: enough waves to fill the chip four times over, at least 8
            // frames each (the per-workgroup table setup amortised).  Small ones -- one serving utterance, a streaming window --
            // spread over ONE round of waves instead, down to a frame per wave (through round 3 the minimum of 8 made 300 frames
            // 38 waves on 4 CUs: 57 us of a 270 us decision; now 17 us).
            const int64_t one_round = (int64_t)ctx().n_cu * blocks_per_cu * wpb;
            int64_t frames_per_wave = std::max<int64_t>(1, (NF + one_round - 1) / one_round);
            if (frames_per_wave > 8) frames_per_wave = std::max<int64_t>(8, (NF + max_waves - 1) / max_waves);
            const int64_t n_waves = (NF + frames_per_wave - 1) / frames_per_wave;
            const int grid = (int)((n_waves + wpb - 1) / wpb);
            int preset = 0;
            for (int pr = 1; pr <= 2 && !preset && n1 == 16; pr++) {
                bool same = true;
                for (int ps = 0; ps < 4; ps++) same = same && tabs.pass_len[ps] == 16 * mel_preset_steps(pr, ps);
                if (same) preset = pr;
            }
#define SR_LAUNCH_FAST(PT, NZ, PCMPTR)                                                              \
    do {                                                                                             \
        if (preset == 1) SR_LAUNCH_FAST_P(PT, NZ, 1, 16, PCMPTR);                                    \
        else if (preset == 2) SR_LAUNCH_FAST_P(PT, NZ, 2, 16, PCMPTR);                               \
        else SR_LAUNCH_FAST_P(PT, NZ, 0, 16, PCMPTR);                                                \
    } while (0)
#define SR_LAUNCH_FAST_P(PT, NZ, MPV, N1V, PCMPTR)                                                  \
    do {                                                                                             \
        if constexpr ((NZ) <= 4) {                                                                   \
            if (wpb == 12) SR_LAUNCH_FAST_W(PT, NZ, MPV, 12, N1V, PCMPTR); else SR_LAUNCH_FAST_W(PT, NZ, MPV, 4, N1V, PCMPTR); \
        } else {                                                                                     \
            SR_LAUNCH_FAST_W(PT, NZ, MPV, 4, N1V, PCMPTR);                                           \
        }                                                                                            \
    } while (0)
#define SR_LAUNCH_FAST_W(PT, NZ, MPV, W, N1V, PCMPTR)                                               \
    do {                                                                                             \
        auto kern = mfcc_frames_fft2048_kernel<PT, NZ, MPV, W, N1V>;                                 \
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * W), lds, ctx().stream, PCMPTR, d_pcm_off, \
                           w.raw_off.p, U, NF, frames_per_wave, dev, mr, w.raw.p);                   \
    } while (0)
#define SR_LAUNCH_FAST_N(PT, PCMPTR)                                                                \
    do {                                                                                             \
        if (n1 == 16) { if (nz1 <= 4) SR_LAUNCH_FAST(PT, 4, PCMPTR); else SR_LAUNCH_FAST(PT, 16, PCMPTR); }                     \
        else if (n1 == 8) { if (nz1 <= 4) SR_LAUNCH_FAST_P(PT, 4, 0, 8, PCMPTR); else SR_LAUNCH_FAST_P(PT, 8, 0, 8, PCMPTR); } \
        else SR_LAUNCH_FAST_P(PT, 4, 0, 4, PCMPTR);                                                  \
    } while (0)
            if (pcm.kind == SRBatch::PCM16) SR_LAUNCH_FAST_N(int16_t, pcm.pcm16.p);
            else SR_LAUNCH_FAST_N(float, pcm.data.p);
#undef SR_LAUNCH_FAST_N
#undef SR_LAUNCH_FAST
#undef SR_LAUNCH_FAST_P
#undef SR_LAUNCH_FAST_W
        } else {
            const int nc = m.fft_size / 2;
            const size_t lds = (size_t)nc * sizeof(float2) * (1 + 4 * 2) + 4 * 64 * sizeof(float);
            const int64_t blocks_needed = (NF + 3) / 4;
            const int blocks_per_cu = std::max<int>(1, (int)(160 * 1024 / lds));
            const int grid = (int)std::min<int64_t>(blocks_needed, (int64_t)ctx().n_cu * std::min(blocks_per_cu, 8));
            if (pcm.kind == SRBatch::PCM16) {
                auto kern = mfcc_frames_kernel<int16_t>;
                SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx().stream, pcm.pcm16.p,
                                   d_pcm_off, w.raw_off.p, U, NF, dev, w.raw.p);
            } else {
                auto kern = mfcc_frames_kernel<float>;
                SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx().stream, pcm.data.p,
                                   d_pcm_off, w.raw_off.p, U, NF, dev, w.raw.p);
            }
        }
        SR_HIP(hipGetLastError());
    }
    if (U > 0 && out.n_rows > 0) {
        ScopedKernelTimer t(T_CMVN);
        hipLaunchKernelGGL(cmvn_delta_kernel, dim3(U), dim3(CMVN_THREADS), 0, ctx().stream, w.raw.p,
                           w.raw_off.p, out.d_offsets.p, m.n_ceps, nd, cmvn, out.data.p, out.dim);
        SR_HIP(hipGetLastError());
    }
    if (m.n_lpc > 0 && NF > 0)   // mix_feature: LPC columns next to the cepstra (same frames, nd == 0)
        lpc_extract_into(m, pcm, w.raw_off.p, NF, m.n_lpc, out.data.p, out.dim, m.n_ceps);
    // Same shapes as the previous call (the serving loop): nothing was uploaded, nothing to wait for --
    // the kernels stay queued behind whatever the caller does next on the stream.
    if (uploaded) sync_stream();
}

}  // namespace sr
