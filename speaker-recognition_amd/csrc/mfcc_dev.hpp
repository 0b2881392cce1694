// mfcc_dev.hpp -- device-side tables of one MFCC extractor, shared by the fp32 kernels (mfcc.hip) and the
// float64-spectrum kernels (mfcc_f64.hip).
#pragma once

#include "common.hpp"
#include "mfcc.hpp"

#include <hip/hip_runtime.h>

namespace sr {

struct MfccDev {
    const float *window;      // [L]
    const float2 *twiddle;    // [NFFT/2]  W_NFFT^k
    const int *mel_row;       // [n_filters+1]
    const int *mel_col;       // [nnz]
    const float *mel_val;     // [nnz]
    const float *mel_floor;   // [n_filters]  ln(1e-100 * row sum): the reference's floored silence
    const float *dct;         // [n_ceps][n_filters]
    int frame_len, frame_shift, fft_size, n_filters, n_ceps;
    float pre_emph;
};

// Mel rows are contiguous column runs; for the fast kernel they are re-laid as 4 passes x 16 bands,
// every run of a pass zero-padded to the same multiple of 16 columns, so that 4 lanes sweep a band
// with plain (unclamped) reads; element i of band b sits at pass_base[pass] + (i/4)*64 + (b%16)*4 + i%4,
// i.e. one sweep step of all 64 lanes reads 64 consecutive floats (bank-conflict free).
struct MelRuns {
    const int *col0;        // [64] first column of the band's run (0 for absent bands)
    const float *pad_val;   // padded weights
    int pad_floats;         // total floats in pad_val
    int pass_base[4];       // float offset of each pass
    int pass_len[4];        // padded run length of the pass (multiple of 16)
};

constexpr int MFCC_DCT_LD = 80;     // row stride of the zero-padded DCT table in LDS: 320 B = 64 B mod 256, so the 4 rows x 4
                                    // parts of a 16-lane ds_read_b128 phase cover 16 distinct 16-byte windows (64 floats would put
                                    // all 16 rows on the same banks)
constexpr int WAVE_SLAB_C = 1088;   // complex slots per wave: max(16*68, 64*17, 1024)

// Mel sweep lengths (16-bin steps per pass of 16 bands) of the two common filterbanks, known at
// compile time so that the sweep unrolls completely and its LDS reads are issued ahead of their use;
// preset 0 takes the lengths from MelRuns at run time (any other fs / n_filters).
__host__ __device__ constexpr int mel_preset_steps(int preset, int pass) {
    return preset == 1 ? (pass == 0 ? 2 : pass == 1 ? 3 : pass == 2 ? 6 : 7)      // fs 16 kHz, 50 filters, FFT 2048
         : preset == 2 ? (pass == 0 ? 2 : pass == 1 ? 4 : 6)                       // fs  8 kHz, 50 filters, FFT 2048
                       : 0;
}

// float64 copies of the host tables for the float64-spectrum path (MFCC.py computes in float64 throughout)
struct MfccDev64 {
    const double *window;     // [L]
    const double2 *twiddle;   // [NFFT/2]  W_NFFT^k
    const double *mel_val;    // [nnz]  (CSR with MfccDev::mel_row / mel_col)
    const double *mel_floor;  // [64]  ln(1e-100 * row sum), 0 beyond n_filters
    const double *dct;        // [n_ceps][n_filters]
    const double *dct_pad;    // [4 parts][4 its][16 coefficients][4]: lane (coefficient, part) reads bands 16 it + 4 part + {0..3}
    double pre_emph;
};

struct MfccDeviceTables {
    DevBuf<double> window64, mel_val64, mel_floor64, dct64, dct_pad64;
    DevBuf<double2> twiddle64;
    DevBuf<float> window, mel_val, mel_floor, dct;
    DevBuf<float2> twiddle;
    DevBuf<int> mel_row, mel_col, mel_col0, mel_cnt;
    DevBuf<float> mel_pad;
    int device = -1;
    int nnz = 0, max_cnt = 0;
    int pass_base[4] = {0, 0, 0, 0}, pass_len[4] = {0, 0, 0, 0}, pad_floats = 0;
    bool runs_contiguous = true;
};

MfccDev upload_tables(SRMfcc &m);           // creates the calling device's tables on first use
MfccDev64 device_tables_f64(SRMfcc &m);     // (after upload_tables)
inline MfccDeviceTables &device_tables(SRMfcc &m) { return *std::static_pointer_cast<MfccDeviceTables>(m.dev[current_device()]); }

// mfcc_f64.hip
bool mfcc_force_generic();
void mfcc_launch_f64(SRMfcc &m, const MfccDev &dev, int pcm_kind, const void *pcm, const int64_t *d_pcm_off, const int64_t *d_raw_off,
                     int n_utt, int64_t n_frames, float *raw);

}  // namespace sr
