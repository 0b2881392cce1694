// mfcc.hpp -- MFCC extractor object behind the C ABI handle `SRMfcc *`.
#pragma once

#include "common.hpp"

#include <memory>
#include <vector>

struct SRBatch;

// Constants of one extractor, as MFCCExtractor.__init__ builds them
// (src/feature/MFCC.py:20-41): float64 on the host, fp32 copies on the device.
struct SRMfcc {
    double fs = 0, pre_emph = 0;
    int fft_size = 0, n_filters = 0, n_ceps = 0, frame_len = 0, frame_shift = 0;
    std::vector<double> window;    // [frame_len]
    std::vector<double> melbank;   // [n_filters][fft_size/2+1]
    std::vector<double> dct;       // [n_ceps][n_filters]  (DCT-II rows 1..n_ceps)
    std::shared_ptr<void> dev[sr::MAX_DEVICES];              // device tables, created on first use on each device
    std::shared_ptr<void> dev_window_f64[sr::MAX_DEVICES];   // float64 window for the LPC kernel
    int n_lpc = 0;                 // > 0: append LPC columns (mix_feature, feature/__init__.py:25-30)
    SRMfcc(double fs, double win_length_ms, double win_shift_ms, int fft_size, int n_filters,
           int n_ceps, double pre_emph);
};

namespace sr {
int64_t mfcc_num_frames(const SRMfcc &m, int64_t n_samples);
void mfcc_extract_batch(SRMfcc &m, SRBatch &pcm, int nd, int cmvn, SRBatch &out);
// The feature stage's device workspace (raw cepstra + their frame offsets).  mfcc_extract_batch uses the calling device's own;
// a caller that alternates between batches of different shapes (the pieces of sr_multi_predict_pcm) keeps one per shape so that
// the cached offset table is not re-uploaded -- and the stream not synchronised -- on every call.
struct MfccScratch;
MfccScratch *mfcc_scratch_new();
void mfcc_scratch_delete(MfccScratch *s);
void mfcc_extract_with(SRMfcc &m, SRBatch &pcm, int nd, int cmvn, SRBatch &out, MfccScratch *scratch);
void mfcc_set_force_generic(bool on);
void mfcc_set_precision(int mode);       // 2: float64 spectrum / ln / DCT for every frame (default), 0: fp32 throughout
int mfcc_precision();
void lpc_extract_into(SRMfcc &m, SRBatch &pcm, const int64_t *d_frame_off, int64_t n_frames, int n_lpc,
                      float *out, int out_stride, int col_off);   // A/B: route FFT_SIZE 2048 through the generic LDS kernel
// ltsd.hip: long-term spectral divergence (voice-activity front end)
int64_t ltsd_num_windows(int64_t n_samples, int N);
void ltsd_noise_spectrum(SRBatch &noise, int N, float *avg_amp_out);
void ltsd_compute(SRBatch &pcm, int N, int order, const float *noise_amp, float *ltsd_out,
                  int64_t *win_offsets_out);

}  // namespace sr
