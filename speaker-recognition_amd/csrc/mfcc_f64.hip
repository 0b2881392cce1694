// mfcc_f64.hip -- the MFCC chain with a float64 spectrum (src/feature/MFCC.py:59-70 computes every frame in float64:
// `fft.fft(frame, FFT_SIZE)` at :66, floor 1e-100 at :67, `D . ln(M . X)` at :69).
//
// Why: an fp32 FFT leaves a noise floor ~156 dB under a frame's strongest bin.  A mel band b of energy E_b then carries a
// relative error ~2 * 1.6e-8 * sqrt(E_max / E_b); voices whose bands lie 60..120 dB apart (SURVEY.md 8d's synthetic speakers
// do: narrow 80 Hz formants over a noise source) come out with CMVN-normalised cepstra 1e-5 (mean) to 2e-2 (max) away from
// the reference, against SURVEY 8d's gate of 1e-5 / 1e-3.  fp32 cannot close that; ten more mantissa bits through the
// window, the butterflies and the twiddles do.  gfx950's vector ALU runs fp64 at 16 lanes per clock (half the fp32 rate),
// and the transform is bound by its LDS exchanges (stores: ~79 B/clk/CU), so the price is ~1.5x the fp32 kernel's time.
//
// What is float64 here: window x sample, pre-emphasis, the 1024-point complex FFT of the packed frame, the real-FFT
// untangle and |X|^2; then ln(E_b) and the DCT.  The power spectrum is rounded to fp32 and the mel sums run in fp32, as in
// mfcc.hip (a sum of <= 104 positive terms: relative error ~1e-7, i.e. 1e-7 absolute in ln E).  Measured against the
// float64 numpy oracle on SURVEY 8d's voices: 2.5e-7 mean, 1.9e-6 max after CMVN (fp32 output rounding included).
//
// Two kernels:
//   mfcc_frames_fft2048_f64_kernel  FFT_SIZE 2048 (the reference's default), frames of <= 512 samples: one wave per frame,
//       16 points per lane in registers, two LDS exchanges (mfcc.hip's scheme) + a HALF exchange for the untangle: bin k and
//       bin 1024-k come from the same two values, so one lane computes both powers and only 8 of a lane's 16 values travel.
//   mfcc_frames_f64_kernel          any other shape (Stockham radix-4 through LDS), float64 throughout incl. the mel sums.
#include "batch.hpp"
#include "mfcc.hpp"
#include "mfcc_dev.hpp"
#include "wave_ops.hpp"

#include <algorithm>

namespace sr {

static int &precision_flag() {
    static int v = 2;
    return v;
}
int mfcc_precision() { return precision_flag(); }
void mfcc_set_precision(int mode) { precision_flag() = mode == 0 ? 0 : 2; }

using d2 = double2;

__device__ __forceinline__ d2 mk(double x, double y) { return make_double2(x, y); }
__device__ __forceinline__ d2 dadd(d2 a, d2 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ d2 dsub(d2 a, d2 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ d2 dmul(d2 a, d2 b) { return mk(fma(-a.y, b.y, a.x * b.x), fma(a.y, b.x, a.x * b.y)); }
__device__ __forceinline__ d2 dmul_mi(d2 a) { return mk(a.y, -a.x); }   // a * (-i)
constexpr double DH = 0.70710678118654752440;
__device__ __forceinline__ d2 dmul_h_mh(d2 a) { return mk(DH * (a.x + a.y), DH * (a.y - a.x)); }      // a * (H - iH)
__device__ __forceinline__ d2 dmul_mh_mh(d2 a) { return mk(DH * (a.y - a.x), -DH * (a.x + a.y)); }    // a * (-H - iH)

// W_NFFT^i for i in [0, NFFT): the table stores the first half, the second is its negative.
__device__ __forceinline__ d2 tw64(const d2 *t, int i, int nc) {
    d2 w = t[i & (nc - 1)];
    if (i & nc) w = mk(-w.x, -w.y);
    return w;
}

// y_r = sum_p x_p (-i)^(p r)
__device__ __forceinline__ void radix4d(d2 a, d2 b, d2 c, d2 d, d2 &y0, d2 &y1, d2 &y2, d2 &y3) {
    const d2 s0 = dadd(a, c), s1 = dsub(a, c), s2 = dadd(b, d), s3 = dmul_mi(dsub(b, d));
    y0 = dadd(s0, s2);
    y1 = dadd(s1, s3);
    y2 = dsub(s0, s2);
    y3 = dsub(s1, s3);
}

// In-register 16-point forward DFT, natural order in and out (4x4 Cooley-Tukey); NZ = leading nonzero inputs (4 or 16).
template <int NZ>
__device__ __forceinline__ void dft16d(d2 (&v)[16]) {
    constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178;
    d2 u[4][4];   // u[q][r]
    if constexpr (NZ <= 4) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) u[q][r] = v[q];
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) radix4d(v[q], v[4 + q], v[8 + q], v[12 + q], u[q][0], u[q][1], u[q][2], u[q][3]);
    }
    // twiddles W_16^(q r)
    u[1][1] = dmul(u[1][1], mk(C1, -S1));
    u[1][2] = dmul_h_mh(u[1][2]);
    u[1][3] = dmul(u[1][3], mk(S1, -C1));
    u[2][1] = dmul_h_mh(u[2][1]);
    u[2][2] = dmul_mi(u[2][2]);
    u[2][3] = dmul_mh_mh(u[2][3]);
    u[3][1] = dmul(u[3][1], mk(S1, -C1));
    u[3][2] = dmul_mh_mh(u[3][2]);
    u[3][3] = dmul(u[3][3], mk(-C1, S1));
#pragma unroll
    for (int r = 0; r < 4; r++) radix4d(u[0][r], u[1][r], u[2][r], u[3][r], v[r], v[r + 4], v[r + 8], v[r + 12]);
}

// value of `x` in lane (l ^ 1) / (l ^ 2): DPP quad_perm, stays on the vector ALU
template <int CTRL>
__device__ __forceinline__ float quad_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double quad_f64(double x) {
    union { double d; int i[2]; } a, b;
    a.d = x;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xf, 0xf, true);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xf, 0xf, true);
    return b.d;
}
constexpr int QUAD_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int QUAD_XOR2 = 0x4E;   // quad_perm [2,3,0,1]

// cos(pi d / 16), d = 0..8 (sin(pi d / 16) = cos(pi (8 - d) / 16)): W_32^d = (W32_COS[d], -W32_COS[8 - d])
__device__ constexpr double W32_COS[9] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                          0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                                          0.19509032201612826785, 0.0};

// ln of a positive fp32 value in float64, to ~1e-15: e = m 2^k with m in [0.707, 1.414), ln m = 2 atanh(s) = 2 s (1 + s^2/3 + ... +
// s^14/15), s = (m - 1) / (m + 1), |s| <= 0.172 (the next term is 5e-15); the quotient from the fp32 reciprocal + one float64
// Newton step.  43 vector instructions where the library's log(double) takes 95 -- of 920 per frame (round 6).  The argument is a
// band energy summed in fp32 (1e-7 relative): nothing here is visible next to that.
__device__ __forceinline__ double ln_pos_f32(float e) {
    int ex;
    float m = frexpf(e, &ex);                         // [0.5, 1)
    if (m < 0.70710678f) {
        m *= 2.0f;
        ex -= 1;
    }
    const double md = (double)m, den = md + 1.0;
    double r = (double)__builtin_amdgcn_rcpf((float)den);
    r = fma(r, fma(-den, r, 1.0), r);
    const double s = (md - 1.0) * r, s2 = s * s;
    // (the coefficients as scalar registers formed HERE: left to itself the compiler hoists them out of the frame loop into 16 vector
    // registers held across the transform -- the run-time-length variant then spilled)
    double c15 = 1.0 / 15.0, c13 = 1.0 / 13.0, c11 = 1.0 / 11.0, c9 = 1.0 / 9.0, c7 = 1.0 / 7.0, c5 = 1.0 / 5.0, c3 = 1.0 / 3.0, ln2 = 0.69314718055994530942;
    asm volatile("" : "+s"(c15), "+s"(c13), "+s"(c11), "+s"(c9), "+s"(c7), "+s"(c5), "+s"(c3), "+s"(ln2));
    double p = c15;
    p = fma(p, s2, c13);
    p = fma(p, s2, c11);
    p = fma(p, s2, c9);
    p = fma(p, s2, c7);
    p = fma(p, s2, c5);
    p = fma(p, s2, c3);
    p = fma(p, s2, 1.0);
    return fma((double)ex, ln2, 2.0 * s * p);
}

constexpr int F64_WIN_BYTES = 4160;               // 520 float64 window taps in LDS (frames of <= 512 samples)
constexpr int F64_WPB = 8;                       // waves per workgroup: 8 x 17 KB of exchange slab + the mel table fill a CU's LDS
constexpr int F64_U_ELEMS = 576;                 // half-spectrum exchange: [8][64] float64 complex (+ lane 0's displaced read)
constexpr int F64_PBUF_BYTE = F64_U_ELEMS * 16;  // power spectrum, fp32[1100]
constexpr int F64_SE_BYTE = F64_PBUF_BYTE + 1100 * 4;    // band energies fp32[64]
constexpr int F64_SLM_BYTE = F64_SE_BYTE + 272;          // ln E, float64[64] (32-byte aligned)
static_assert(F64_SLM_BYTE % 32 == 0 && F64_SLM_BYTE + 512 <= WAVE_SLAB_C * 16, "per-wave slab layout");

// One wave = one contiguous range of frames, as in mfcc.hip's fft2048 kernel.  MP = mel preset (mel_preset_steps).
template <typename PcmT, int MP>
__global__ __launch_bounds__(64 * F64_WPB)
void mfcc_frames_fft2048_f64_kernel(const PcmT *__restrict__ pcm, const int64_t *__restrict__ sample_off,
                                    const int64_t *__restrict__ frame_off, int n_utt, int64_t n_frames,
                                    int64_t frames_per_wave, MfccDev p, MfccDev64 q, MelRuns mr, float *__restrict__ raw) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    constexpr int NC = 1024;                                  // complex points; FFT_SIZE = 2 NC
    float *s_melval = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // window taps, float64, shifted by one and zero-padded: s_win[1 + i] = window[i] (i < L), 0 elsewhere -- a lane's three taps
    // (i0 - 1, i0, i0 + 1) are three consecutive entries starting at the even index i0
    double *s_win = reinterpret_cast<double *>(smem + (((size_t)mr.pad_floats * 4 + 31) & ~(size_t)31));
    char *slab_b = reinterpret_cast<char *>(s_win) + F64_WIN_BYTES + (size_t)wave * WAVE_SLAB_C * sizeof(d2);
    d2 *slab = reinterpret_cast<d2 *>(slab_b);
    float *pbuf = reinterpret_cast<float *>(slab_b + F64_PBUF_BYTE);
    float *s_e = reinterpret_cast<float *>(slab_b + F64_SE_BYTE);
    double *s_lm = reinterpret_cast<double *>(slab_b + F64_SLM_BYTE);

    for (int i = threadIdx.x; i < mr.pad_floats; i += 64 * F64_WPB) s_melval[i] = mr.pad_val[i];
    for (int i = threadIdx.x; i < F64_WIN_BYTES / 8; i += 64 * F64_WPB) s_win[i] = (i >= 1 && i <= p.frame_len) ? q.window[i - 1] : 0.0;
    __syncthreads();

    // ---- per-lane constants ----
    const d2 w1 = q.twiddle[2 * lane];                  // W_1024^lane  (pass-1 twiddles are its powers)
    const d2 wl = q.twiddle[lane];                      // W_2048^lane  (untangle twiddles are wl * W_32^d)
    const int bb = lane & 15, gg = lane >> 4;
    const d2 w64_1 = q.twiddle[32 * bb];                // W_64^b; its square and cube are formed where they are used (8 registers)
    const int L = p.frame_len;
    const double lm_floor = q.mel_floor[lane];          // ln(1e-100 * row sum) of band `lane` (0 beyond n_filters)
    // mel: 4 lanes per band, 16 bands per pass
    const int m_part = lane & 3, m_bl = lane >> 2;
    int m_c0[4];
#pragma unroll
    for (int ps = 0; ps < 4; ps++) m_c0[ps] = mr.col0[16 * ps + m_bl];

    const int64_t gwave = (int64_t)blockIdx.x * F64_WPB + wave;
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

    // sample fetch: three taps per row, addresses clamped into the frame (what an out-of-frame tap reads is never used);
    // the offsets are re-formed per frame from the lane index (held, they would cost 12 registers)
    PcmT cm[4], c0[4], c1[4];
    auto fetch = [&](int64_t base) {
        const PcmT *fp = pcm + base;
#pragma unroll
        for (int n1 = 0; n1 < 4; n1++) {
            const int i0 = 2 * (64 * n1 + lane);
            c0[n1] = fp[min(i0, L - 1)];
            cm[n1] = fp[max(min(i0 - 1, L - 1), 0)];
            c1[n1] = fp[min(i0 + 1, L - 1)];
        }
    };
    fetch(utt_s0 + (f_begin - utt_f0) * p.frame_shift);

    for (int64_t frame = f_begin; frame < f_end; frame++) {
        // ---- window + pre-emphasis on the windowed samples (MFCC.py:61-64), packed z = y[2n] + i y[2n+1] ----
        d2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 4; n1++) {
            const int i0 = 2 * (64 * n1 + lane);
            const d2 wm0 = *reinterpret_cast<const d2 *>(s_win + i0);      // window[i0 - 1], window[i0]
            const double wp1 = s_win[i0 + 2];                              // window[i0 + 1]
            const double x0 = (double)c0[n1], xm = (double)cm[n1], x1 = (double)c1[n1];
            const double y0 = wm0.y * x0;
            const double re = fma(-(wm0.x * q.pre_emph), xm, y0);
            const double im = fma(-q.pre_emph, y0, wp1 * x1);
            v[n1] = mk(i0 < L ? re : 0.0, i0 + 1 < L ? im : 0.0);
        }
        // ---- prefetch the next frame's samples (in flight during the transform) ----
        if (frame + 1 < f_end) {
            int64_t nf = frame + 1;
            while (nf >= utt_f1) {     // utterances with zero frames are skipped
                utt++;
                utt_f0 = utt_f1;
                utt_f1 = frame_off[utt + 1];
                utt_s0 = sample_off[utt];
            }
            fetch(utt_s0 + (nf - utt_f0) * p.frame_shift);
        }
        // ---- pass 1: 16-point DFT over n1 (rows 4..15 are zero), twiddle W_1024^(lane k1) by recurrence, exchange ----
        dft16d<4>(v);
        {
            d2 cur = w1;
            v[1] = dmul(v[1], cur);
#pragma unroll
            for (int k1 = 2; k1 < 16; k1++) {
                cur = dmul(cur, w1);
                v[k1] = dmul(v[k1], cur);
            }
        }
        wave_sync();      // previous frame's readers are done with the slab
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) slab[k1 * 68 + lane] = v[k1];
        wave_sync();
        // ---- pass 2: radix-4 over a for (k1 = 4 g + i, b), twiddle W_64^(bc), exchange ----
        {
            d2 w64_2 = dmul(w64_1, w64_1);
            asm volatile("" : "+v"(w64_2.x), "+v"(w64_2.y));       // (not loop-invariant as far as the compiler can tell)
            const d2 w64_3 = dmul(w64_2, w64_1);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const d2 *rowp = slab + (4 * gg + i) * 68 + bb;
                radix4d(rowp[0], rowp[16], rowp[32], rowp[48], v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
                v[4 * i + 1] = dmul(v[4 * i + 1], w64_1);
                v[4 * i + 2] = dmul(v[4 * i + 2], w64_2);
                v[4 * i + 3] = dmul(v[4 * i + 3], w64_3);
            }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) slab[((4 * gg + i) + 16 * c) * 17 + bb] = v[4 * i + c];
        wave_sync();
        // ---- pass 3: lane l = k1 + 16 c holds C[k1][b][c], b = 0..15 -> Z[l + 64 d] ----
#pragma unroll
        for (int b = 0; b < 16; b++) v[b] = slab[lane * 17 + b];
        dft16d<16>(v);
        wave_sync();
        // ---- real-FFT untangle + power spectrum (MFCC.py:66).  With E' = Zk + conj Zr, O' = -i (Zk - conj Zr), T = W^k O'
        // (r = 1024 - k):  2 X[k] = E' + T  and  2 X[r] = conj(E' - T)  -- the lane that owns Z[k], k = lane + 64 d, d < 8, fetches
        // Zr (lane 64 - l's register 15 - d) and produces both powers; the factor 4 is folded into the mel weights. ----
#pragma unroll
        for (int d = 8; d < 16; d++) slab[(d - 8) * 64 + lane] = v[d];
        wave_sync();
        {
            const int src = (64 - lane) & 63;
            const int shift = lane == 0 ? 64 : 0;        // lane 0 pairs with its own registers 16 - d (bins 64 d <-> 1024 - 64 d)
#pragma unroll
            for (int d = 0; d < 8; d++) {
                const d2 zk = v[d];
                d2 zr = slab[(7 - d) * 64 + shift + src];
                if (d == 0 && lane == 0) zr = zk;        // bins 0 and 1024: (Re Z0 + Im Z0)^2 and (Re Z0 - Im Z0)^2
                const d2 e = mk(zk.x + zr.x, zk.y - zr.y);
                const d2 o = mk(zk.y + zr.y, zr.x - zk.x);
                // W_2048^(lane + 64 d) = wl * W_32^d; (wl * (c * o)) keeps the product out of the loop-invariant registers
                d2 t;
                if (d == 0) t = dmul(wl, o);
                else if (d == 4) t = dmul(wl, dmul_h_mh(o));          // W_32^4 = W_8
                else t = dmul(wl, dmul(mk(W32_COS[d], -W32_COS[8 - d]), o));
                const d2 xp = dadd(e, t), xm = dsub(e, t);
                const int k = lane + 64 * d;
                pbuf[k] = (float)fma(xp.x, xp.x, xp.y * xp.y);
                pbuf[NC - k] = (float)fma(xm.x, xm.x, xm.y * xm.y);
            }
            if (lane == 0) pbuf[512] = (float)(4.0 * fma(v[8].x, v[8].x, v[8].y * v[8].y));   // bin 512 pairs with itself
            // the padded mel sweeps read past bin 1024 with zero weights: what lies there must be finite (stale float64 halves are not)
            pbuf[1025 + lane] = 0.f;
            if (lane < 11) pbuf[1089 + lane] = 0.f;
        }
        // DCT weights of this lane (coefficient lane / 4, bands 16 it + 4 (lane % 4) + {0..3}): 128 B, in flight during the mel sweep
        // (the address goes through an opaque register: hoisted out of the frame loop the 16 weights would hold 32 VGPRs for good)
        double dw[4][4];
        int dct_lane = lane;
        asm volatile("" : "+v"(dct_lane));
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const double4 wv = reinterpret_cast<const double4 *>(q.dct_pad)[it * 64 + dct_lane];
            dw[it][0] = wv.x; dw[it][1] = wv.y; dw[it][2] = wv.z; dw[it][3] = wv.w;
        }
        wave_sync();
        // ---- mel filterbank in fp32 (MFCC.py:67-69): 4 lanes sweep one band's column run, 16 bands per pass ----
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const int len = mr.pass_len[ps];
            const float4 *mv4 = reinterpret_cast<const float4 *>(s_melval + mr.pass_base[ps]) + lane;
            int c0v = m_c0[ps];
            if constexpr (MP == 0) {
                // (the run-time-length variant is one register over: its sweep starts are re-read per frame -- L1 -- through an
                // opaque index instead of held across the transform)
                int bl = 16 * ps + m_bl;
                asm volatile("" : "+v"(bl));
                c0v = mr.col0[bl];
            }
            const float4 *pp4 = reinterpret_cast<const float4 *>(pbuf + c0v) + m_part;
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
            acc += quad_f32<QUAD_XOR1>(acc);
            acc += quad_f32<QUAD_XOR2>(acc);
            if (m_part == 0) s_e[16 * ps + m_bl] = acc;
        }
        wave_sync();
        // ---- ln E in float64, one band per lane (a band whose every bin sits at the 1e-100 floor sums to 0 in fp32) ----
        {
            const float e = s_e[lane];
            s_lm[lane] = e > 0.f ? ln_pos_f32(e) : lm_floor;
        }
        wave_sync();
        // ---- DCT-II rows 1..n_ceps in float64: 4 lanes per coefficient ----
        {
            const int cidx = lane >> 2;
            const double4 *lp = reinterpret_cast<const double4 *>(s_lm) + m_part;
            double o0 = 0.0, o1 = 0.0;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const double4 lv = lp[it * 4];
                o0 = fma(dw[it][0], lv.x, o0);
                o1 = fma(dw[it][1], lv.y, o1);
                o0 = fma(dw[it][2], lv.z, o0);
                o1 = fma(dw[it][3], lv.w, o1);
            }
            double o = o0 + o1;
            o += quad_f64<QUAD_XOR1>(o);
            o += quad_f64<QUAD_XOR2>(o);
            if (m_part == 0 && cidx < p.n_ceps) raw[frame * p.n_ceps + cidx] = (float)o;
        }
    }
}

// Any FFT_SIZE / frame length: Stockham autosort FFT through LDS (mfcc.hip's generic kernel), float64 throughout --
// the floor at 1e-100 and the dense-matrix semantics of M . X included.
template <typename PcmT, int WPB>
__global__ __launch_bounds__(64 * WPB)
void mfcc_frames_f64_kernel(const PcmT *__restrict__ pcm, const int64_t *__restrict__ sample_off,
                            const int64_t *__restrict__ frame_off, int n_utt, int64_t n_frames,
                            MfccDev p, MfccDev64 q, float *__restrict__ raw /* [n_frames][n_ceps] */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nc = p.fft_size >> 1;           // complex FFT length
    d2 *s_tw = reinterpret_cast<d2 *>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    d2 *buf_a = s_tw + nc + (size_t)wave * 2 * nc;
    d2 *buf_b = buf_a + nc;
    double *s_lm = reinterpret_cast<double *>(s_tw + nc + (size_t)WPB * 2 * nc) + wave * 64;

    for (int i = threadIdx.x; i < nc; i += 64 * WPB) s_tw[i] = q.twiddle[i];
    __syncthreads();

    const int64_t frames_per_iter = (int64_t)gridDim.x * WPB;
    const int64_t iters = (n_frames + frames_per_iter - 1) / frames_per_iter;
    for (int64_t it = 0; it < iters; it++) {
        const int64_t frame = (it * gridDim.x + blockIdx.x) * WPB + wave;
        const bool active = frame < n_frames;   // wave-uniform

        int64_t base = 0;
        if (active) {
            int lo = 0, hi = n_utt;               // frame_off[lo] <= frame < frame_off[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (frame_off[mid] <= frame) lo = mid; else hi = mid;
            }
            base = sample_off[lo] + (frame - frame_off[lo]) * p.frame_shift;
        }

        // ---- window, then pre-emphasis on the windowed samples (MFCC.py:61-64); z[n] = y[2n] + i y[2n+1] ----
        if (active) {
            const int L = p.frame_len;
            for (int n = lane; n < nc; n += 64) {
                double re = 0.0, im = 0.0;
                const int i0 = 2 * n;
                if (i0 < L) {
                    const double c0 = (double)pcm[base + i0] * q.window[i0];
                    const double pm1 = i0 > 0 ? (double)pcm[base + i0 - 1] * q.window[i0 - 1] : 0.0;
                    re = i0 > 0 ? c0 - pm1 * q.pre_emph : c0;
                    if (i0 + 1 < L) {
                        const double c1 = (double)pcm[base + i0 + 1] * q.window[i0 + 1];
                        im = c1 - c0 * q.pre_emph;
                    }
                }
                buf_a[n] = mk(re, im);
            }
        }
        __syncthreads();

        // ---- Stockham autosort FFT of length nc ----
        d2 *in = buf_a, *out = buf_b;
        int ns = 1;
        for (; ns * 4 <= nc; ns *= 4) {
            if (active) {
                const int T = nc >> 2;
                const int tstep = p.fft_size / (4 * ns);   // W_nc^(k*nc/(4ns)) = W_NFFT^(k*NFFT/(4ns))
                for (int j = lane; j < T; j += 64) {
                    const int k = j & (ns - 1);
                    const int qq = k * tstep;
                    const d2 v0 = in[j];
                    const d2 v1 = dmul(in[j + T], tw64(s_tw, qq, nc));
                    const d2 v2 = dmul(in[j + 2 * T], tw64(s_tw, 2 * qq, nc));
                    const d2 v3 = dmul(in[j + 3 * T], tw64(s_tw, 3 * qq, nc));
                    d2 y0, y1, y2, y3;
                    radix4d(v0, v1, v2, v3, y0, y1, y2, y3);
                    const int j0 = ((j - k) << 2) + k;
                    out[j0] = y0;
                    out[j0 + ns] = y1;
                    out[j0 + 2 * ns] = y2;
                    out[j0 + 3 * ns] = y3;
                }
            }
            __syncthreads();
            d2 *t = in; in = out; out = t;
        }
        if (ns < nc) {   // one radix-2 pass (ns == nc/2)
            if (active) {
                const int T = nc >> 1;
                for (int j = lane; j < T; j += 64) {
                    const d2 v0 = in[j];
                    const d2 v1 = dmul(in[j + T], tw64(s_tw, 2 * j, nc));   // W_nc^j
                    out[j] = dadd(v0, v1);
                    out[j + T] = dsub(v0, v1);
                }
            }
            __syncthreads();
            d2 *t = in; in = out; out = t;
        }

        // ---- real-FFT untangle + power spectrum (MFCC.py:66) floored at 1e-100 (:67), bins 0..nc into `out` ----
        double *pw = reinterpret_cast<double *>(out);
        if (active) {
            for (int k = lane; k <= nc; k += 64) {
                const d2 zk = in[k & (nc - 1)];
                const d2 zr = in[(nc - k) & (nc - 1)];
                const d2 e = mk(0.5 * (zk.x + zr.x), 0.5 * (zk.y - zr.y));
                const d2 o = mk(0.5 * (zk.y + zr.y), -0.5 * (zk.x - zr.x));
                const d2 x = dadd(e, dmul(tw64(s_tw, k, nc), o));
                const double pk = fma(x.x, x.x, x.y * x.y);
                pw[k] = pk < 1e-100 ? 1e-100 : pk;
            }
        }
        __syncthreads();

        // ---- mel filterbank (sparse rows, fixed-order wave sum) + ln (MFCC.py:67-69) ----
        if (active) {
            double mine = 0.0;
            for (int b = 0; b < p.n_filters; b++) {
                double acc = 0.0;
                const int e1 = p.mel_row[b + 1];
                for (int e = p.mel_row[b] + lane; e < e1; e += 64) acc = fma(q.mel_val[e], pw[p.mel_col[e]], acc);
                acc = wave_sum_f64(acc);
                if (lane == b) mine = acc > 0.0 ? log(acc) : q.mel_floor[b];
            }
            s_lm[lane] = mine;
        }
        __syncthreads();

        // ---- DCT-II rows 1..n_ceps ----
        if (active && lane < p.n_ceps) {
            double acc = 0.0;
            const double *drow = q.dct + lane * p.n_filters;
            for (int b = 0; b < p.n_filters; b++) acc = fma(drow[b], s_lm[b], acc);
            raw[frame * p.n_ceps + lane] = (float)acc;
        }
        __syncthreads();
    }
}

void mfcc_launch_f64(SRMfcc &m, const MfccDev &dev, int pcm_kind, const void *pcm, const int64_t *d_pcm_off, const int64_t *d_raw_off,
                     int n_utt, int64_t n_frames, float *raw) {
    auto &tabs = device_tables(m);
    const MfccDev64 dev64 = device_tables_f64(m);
    const size_t mel_bytes = ((size_t)tabs.pad_floats * 4 + 31) & ~(size_t)31;
    const size_t lds_fast = mel_bytes + F64_WIN_BYTES + (size_t)F64_WPB * WAVE_SLAB_C * sizeof(d2);
    const bool fast = m.fft_size == 2048 && m.frame_len <= 512 && tabs.runs_contiguous && m.n_ceps <= 16 &&
                      lds_fast <= (size_t)160 * 1024 && !mfcc_force_generic();
    if (fast) {
        MelRuns mr;
        mr.col0 = tabs.mel_col0.p;
        mr.pad_val = tabs.mel_pad.p;
        mr.pad_floats = tabs.pad_floats;
        for (int ps = 0; ps < 4; ps++) {
            mr.pass_base[ps] = tabs.pass_base[ps];
            mr.pass_len[ps] = tabs.pass_len[ps];
        }
        // one contiguous frame range per wave; one 8-wave workgroup per CU (its LDS)
        // Frames per wave: the chip holds ONE round of waves at a time (a workgroup per CU), every wave walks its frames one after the
        // other, so a pass costs rounds x frames per wave.  Of 1..4 rounds the cheapest (64 utterances x 300
        // frames: one round of 10 frames per wave, not 1.17 rounds of 8 -- 0.108 -> 0.07 ms); large batches end up with four rounds
        // of equal waves, which evens out what the scheduler does to them.
        const int64_t one_round = (int64_t)ctx().n_cu * F64_WPB;
        int64_t frames_per_wave = 1, best_cost = -1;
        for (int64_t r = 1; r <= 4; r++) {
            const int64_t fpw = std::max<int64_t>(1, (n_frames + r * one_round - 1) / (r * one_round));
            const int64_t waves = (n_frames + fpw - 1) / fpw;
            const int64_t cost = ((waves + one_round - 1) / one_round) * fpw;
            // (more rounds of shorter waves are preferred within 2 % while a wave still has >= 32 frames to amortise its start on)
            if (best_cost < 0 || cost < best_cost - best_cost / 50 || (cost <= best_cost + best_cost / 50 && fpw >= 32)) {
                best_cost = cost;
                frames_per_wave = fpw;
            }
        }
        const int64_t n_waves = (n_frames + frames_per_wave - 1) / frames_per_wave;
        const int grid = (int)((n_waves + F64_WPB - 1) / F64_WPB);
        int preset = 0;
        for (int pr = 1; pr <= 2 && !preset; pr++) {
            bool same = true;
            for (int ps = 0; ps < 4; ps++) same = same && tabs.pass_len[ps] == 16 * mel_preset_steps(pr, ps);
            if (same) preset = pr;
        }
#define SR_LAUNCH_F64(PT, MPV)                                                                              \
    do {                                                                                                     \
        auto kern = mfcc_frames_fft2048_f64_kernel<PT, MPV>;                                                 \
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * F64_WPB), lds_fast, ctx().stream, static_cast<const PT *>(pcm), d_pcm_off,      \
                           d_raw_off, n_utt, n_frames, frames_per_wave, dev, dev64, mr, raw);                \
    } while (0)
#define SR_LAUNCH_F64_P(PT)                                                                                 \
    do {                                                                                                     \
        if (preset == 1) SR_LAUNCH_F64(PT, 1); else if (preset == 2) SR_LAUNCH_F64(PT, 2); else SR_LAUNCH_F64(PT, 0); \
    } while (0)
        if (pcm_kind == SRBatch::PCM16) SR_LAUNCH_F64_P(int16_t); else SR_LAUNCH_F64_P(float);
#undef SR_LAUNCH_F64_P
#undef SR_LAUNCH_F64
        return;
    }
    const int nc = m.fft_size / 2;
    // waves per workgroup: as many as the LDS takes (float64 twiddles + two slabs per wave)
    auto lds_for = [&](int w) { return (size_t)nc * sizeof(d2) * (1 + 2 * w) + (size_t)w * 64 * sizeof(double); };
    const int wpb = lds_for(4) <= (size_t)160 * 1024 ? 4 : lds_for(2) <= (size_t)160 * 1024 ? 2 : 1;
    const size_t lds = lds_for(wpb);
    const int64_t blocks_needed = (n_frames + wpb - 1) / wpb;
    const int blocks_per_cu = std::max<int>(1, (int)(160 * 1024 / lds));
    const int grid = (int)std::min<int64_t>(blocks_needed, (int64_t)ctx().n_cu * std::min(blocks_per_cu, 8));
#define SR_LAUNCH_G64(PT, W)                                                                                \
    do {                                                                                                     \
        auto kern = mfcc_frames_f64_kernel<PT, W>;                                                           \
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * W), lds, ctx().stream, static_cast<const PT *>(pcm), d_pcm_off, d_raw_off, \
                           n_utt, n_frames, dev, dev64, raw);                                                \
    } while (0)
#define SR_LAUNCH_G64_W(PT)                                                                                 \
    do {                                                                                                     \
        if (wpb == 4) SR_LAUNCH_G64(PT, 4); else if (wpb == 2) SR_LAUNCH_G64(PT, 2); else SR_LAUNCH_G64(PT, 1); \
    } while (0)
    if (pcm_kind == SRBatch::PCM16) SR_LAUNCH_G64_W(int16_t); else SR_LAUNCH_G64_W(float);
#undef SR_LAUNCH_G64_W
#undef SR_LAUNCH_G64
}

}  // namespace sr
