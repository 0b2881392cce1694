// score.hpp -- host entry points of the scoring path shared between translation units.
#pragma once

#include "batch.hpp"
#include "gmm_model.hpp"

namespace sr {

struct ScoreOptions {
    int frames_per_lane = 0;   // 0 = auto; 1, 2 or 4 frames resident per lane
    int model_groups = 0;      // 0 = auto; workgroups per frame tile along the model axis
    int packed = 0;            // -1 = scalar v_fma_f32; 0 (auto) / 1 = v_pk_fma_f32, two frames per VGPR pair
};
ScoreOptions &score_options();

// Device-resident results of the last scoring call (valid until the next one).
struct ScoreResult {
    const double *d_sums = nullptr;    // [U][S]
    const int *d_argmax = nullptr;     // [U]
    const float *d_frame_ll = nullptr; // [S][n_frames] when requested
};

// Scores every utterance of `feat` against every model of `set`; leaves results on the device.
ScoreResult score_device(SRModelSet &set, SRBatch &feat, bool want_frame_ll, int flags);
// Same, then copies what the caller asked for to host memory.
void score_batch_set(SRModelSet &set, SRBatch &feat, double *sums_out, int *argmax_out,
                     float *frame_ll_out, int flags);
// Packs + uploads a model set on the current device.
void upload_model_set(SRModelSet &s);

}  // namespace sr
