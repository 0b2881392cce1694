// fork_proxy.hpp -- compute on behalf of a process that lost its GPU runtime to fork() (fork_proxy.cpp, common.hpp).
#pragma once

#include "gmm_model.hpp"

struct Parameter;

namespace sr {

// the frames of X against one model: per-frame values and / or their sum (what score_all / score_batch / score_instance /
// sr_score_frames_f32 return), computed by this process's helper
void fork_proxy_score(GMM *g, const float *X, long n, int dim, float *ll_out, double *sum_out, int flags);
// train_model / train_model_from_ubm / sr_train_f32: EM or MAP in the helper; `gmm` receives the result.  Returns the iterations run.
int fork_proxy_train(GMM &gmm, const GMM *ubm, const float *X, long n, int dim, const Parameter &param, long seed);
// every successful sr_set_option is recorded (a helper starts from the library's defaults and is told) and, in a forked child with a
// live helper, forwarded
void fork_proxy_score_models(GMM *const *models, int n_models, const float *X, long n, int dim, double *sums_out, int flags);
void fork_proxy_note_option(const char *key, long value);
// test hook: how many models a helper keeps (0: the default, HELPER_MAX_MODELS); sr_set_option("debug_helper_max_models", n)
void fork_proxy_set_max_models(long n);
// pthread_atfork child handler: a fresh mutex for the helper record
void fork_proxy_atfork_child();

}  // namespace sr
