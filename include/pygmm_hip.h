/*
 * include/pygmm_hip.h -- C ABI of lib/pygmm.so (MI355X / gfx950 build).
 *
 * Part 1 is a drop-in for the reference's FFI, /root/reference/src/gmm/src/pygmm.hh:11-43:
 * the same ten symbols, the same signatures and argument meaning, so the reference's ctypes
 * caller (src/gmm/python/pygmm.py:16 `cdll.LoadLibrary('../lib/pygmm.so')`) binds it
 * unchanged.  Differences in behaviour, all deliberate:
 *   - nothing throws across the boundary (reference: `throw "literal"`, gmm.cc:45,59,585);
 *     legacy entry points report through sr_last_error() and return NULL / NaN / no-op;
 *   - `concurrency` is accepted and ignored (the HIP grid replaces the thread pool,
 *     gmm.cc:533-560);
 *   - score_instance works (the reference aborts on assert(buffer != NULL),
 *     pygmm.cc:113-117 -> gmm.cc:185).
 * Part 2 (sr_ prefix) is the contiguous / batched / device-resident interface the hot path
 * actually wants; the Python mirror of the reference surface calls these.
 *
 * Plain C types only; no HIP or C++ types cross this boundary.
 */
#ifndef PYGMM_HIP_H
#define PYGMM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct GMM GMM; /* opaque; reference: class GMM, src/gmm/src/gmm.hh:130-173 */

/* src/gmm/src/pygmm.hh:12-26 (mirrored by ctypes at src/gmm/python/pygmm.py:18-27) */
struct Parameter {
    int nr_instance;
    int nr_dim;
    int nr_mixture;
    double min_covar;
    double threshold;
    int nr_iteration;
    int init_with_kmeans;
    int concurrency;
    int verbosity;          /* >= 1: the reference's progress lines (gmm.cc:588-590, :641); >= 2 (extension): the phase
                               times of every EM / MAP iteration on stdout */
};

/* ---------------- Part 1: legacy symbols (pygmm.hh:28-41) ---------------- */

/* pygmm.hh:28 / pygmm.cc:50-52.  covariance_type must be 1 (COVTYPE_DIAGONAL, gmm.hh:18-22). */
GMM *new_gmm(int nr_mixture, int covariance_type);
/* pygmm.hh:29 / pygmm.cc:54-56 -> GMM::load, gmm.cc:664-682 (text format, gmm.cc:101-150). */
GMM *load(const char *model_file);
/* pygmm.hh:31 / pygmm.cc:58-61 -> GMM::dump, gmm.cc:655-662 (6 significant digits). */
void dump(GMM *gmm, const char *model_file);
/* pygmm.hh:33 / pygmm.cc:63-83 -> GMMTrainerBaseline::train, gmm.cc:581-653. X_in: nr_instance row pointers. */
void train_model(GMM *gmm, double **X_in, struct Parameter *param);
/* pygmm.hh:34 / pygmm.cc:85-96 -> MAP means-only adaptation, gmmubm.cc:29-81. */
void train_model_from_ubm(GMM *gmm, GMM *ubm, double **X_in, struct Parameter *param);
/* pygmm.hh:36 / pygmm.cc:98-102: sum over frames of the per-frame log-likelihood. */
double score_all(GMM *gmm, double **X_in, int nr_instance, int nr_dim, int concurrency);
/* pygmm.hh:37 / pygmm.cc:104-111: per-frame log-likelihoods into caller-owned prob_out[nr_instance]. */
void score_batch(GMM *gmm, double **X_in, double *prob_out, int nr_instance, int nr_dim, int concurrency);
/* pygmm.hh:38 / pygmm.cc:113-117. */
double score_instance(GMM *gmm, double *x_in, int nr_dim);
/* pygmm.hh:40-41 / pygmm.cc:119-120. */
int get_dim(GMM *gmm);
int get_nr_mixtures(GMM *gmm);

/* ---------------- Part 2: extensions ---------------- */

/* Status: 0 = ok, negative = error (text via sr_last_error(), thread-local). */
const char *sr_last_error(void);

/* Processes.  The reference's drivers fit in the parent and THEN fork a multiprocessing.Pool whose workers score
 * (src/test/test-nperson.py:126-139, src/test/test-gmm.py:120-133); a HIP runtime does not survive fork().  So:
 *   - the runtime is initialised lazily, at the first call that needs the device: a pool forked BEFORE that gives
 *     every worker a runtime of its own;
 *   - a process forked AFTER its parent used the GPU is detected (pthread_atfork + pid).  There the reference's ten
 *     symbols above, sr_score_frames_f32 and sr_train_f32 keep working -- they are served by a helper process
 *     (lib/sr_fork_helper next to lib/pygmm.so, spawned at the first such call, one per forked child; csrc/fork_proxy.cpp)
 *     with the results of the in-process path; every other entry point that needs the device returns its error
 *     status with a message naming the remedy, without touching HIP; host-only entry points (load / dump / new_gmm /
 *     get_* / sr_gmm_* / sr_mfcc_tables ...) work as always; inherited device handles may be freed (a no-op there).
 * sr_gpu_runtime_lost() = 1 in such a process.  Pinned by tests/test_fork.py (CPU) and tests/test_gpu_fork.py. */
int sr_gpu_runtime_lost(void);

/* Device plumbing.  One process may drive several GPUs:
 * every host thread has a current device; stream, workspaces, timers and the entry-point lock exist
 * once per device, so threads on different devices run in parallel and threads sharing a device are
 * serialised.  Handles that own device memory (SRBatch, SRModelSet, SRStream) belong to the device
 * they were created on and refuse calls from a thread that is on another one.
 * sr_set_device: the process default (what threads that never chose get) and the calling thread's;
 * sr_set_thread_device: the calling thread's only. */
int sr_device_count(void);
int sr_set_device(int device);
int sr_set_thread_device(int device);
int sr_get_device(void);
int sr_device_synchronize(void);
int sr_device_name(char *buf, int buflen);
/* NUMA: the node of `device` per sysfs (-1: unknown); sr_bind_thread_near_device pins the CALLING host thread to that node's
 * cores and returns the node (-1: left alone).  The slot threads of sr_multi_* do it themselves. */
int sr_device_numa_node(int device);
int sr_bind_thread_near_device(int device);

/* Model handles beyond the legacy constructors. */
void sr_free_gmm(GMM *gmm);
GMM *sr_gmm_from_arrays(int nr_mixture, int nr_dim, const double *weights, const double *mean,
                        const double *sigma);                       /* sigma = std deviations */
int sr_gmm_get_params(GMM *gmm, double *weights, double *mean, double *sigma);
int sr_gmm_dumps(GMM *gmm, char *buf, long buflen, long *needed);     /* text format into memory */
GMM *sr_gmm_loads(const char *text);

/* Contiguous fp32 scoring of ONE model (frames row-major [n][dim], host memory).
 * flags: SR_CLAMP_COMPAT reproduces the reference's underflow behaviour: its densities and its mixture
 * sum are formed in the linear domain under FTZ arithmetic, so a term w_k p_k(x) below DBL_MIN =
 * exp(-708.396) counts as 0 and an all-zero sum returns ln(1e-15) (gmm.cc:34-38, :237-244; pinned by
 * reference-DSO vectors, tests/golden/make_clamp_golden.py) -- and so does a term any PARTIAL product of
 * which dips below DBL_MIN (gmm.cc:192-195), or one dimension of which reaches the exponent floor of
 * fastexp.cc:104-131, although its full product would be representable.  The frames this can matter
 * for (log-likelihood within a few tens of nats of -708.4: frames ~37 sigma from every mixture) are found
 * by every engine and re-evaluated with the reference's own arithmetic (csrc/gmm_flush.hip; pinned by
 * tests/golden/make_flush_golden.py on the DSO).  Which partial products exist is the compiler's choice under
 * the reference's -ffast-math: option "flush_order" 2 (default) = the DSO as g++ 11 builds it from the
 * reference's flags (even / odd dimension lanes), 1 = the source's order.
 * SR_SCORE_PRECISE keeps to the fp32-grade engines (the split-fp16 ones carry 22 significand bits). */
#define SR_CLAMP_COMPAT 1
#define SR_SCORE_PRECISE 0x200
int sr_score_frames_f32(GMM *gmm, const float *X, long n, int dim, float *ll_out, double *sum_out,
                        int flags);

/* All of `models` on ONE utterance in one fused pass: sums_out[i] = the sum over the frames of model i's log-likelihood -- what
 * the reference's per-speaker loop of score_all calls computes (gmmset.py:95-99).  The packed set of the last model list is
 * kept (keyed by the handles and their parameters' state), so a loop over utterances packs once.  Unlike SRModelSet handles this
 * entry point also works in a process forked after its parent used the GPU (the reference's Pool drivers,
 * test-nperson.py:126-146): there it is ONE conversation with the process's helper instead of one per model. */
int sr_score_models_f32(GMM *const *models, int n_models, const float *X /* [n_frames][dim] */, long n_frames, int dim,
                        double *sums_out /* [n_models] */, int flags /* SR_CLAMP_COMPAT | ... */);

/* Speaker set: S models packed once, resident in HBM (replaces the per-speaker ABI loop of
 * src/testbench/gmmset.py:59-64,95-99). */
typedef struct SRModelSet SRModelSet;
SRModelSet *sr_modelset_create(GMM *const *models, int n_models);
void sr_modelset_free(SRModelSet *set);
int sr_modelset_size(SRModelSet *set);
/* Conditioning of the packed set as the engine dispatcher sees it: out8[0] = amp = max_k sum_d
 * ((mu_kd - centre_d) / sigma_kd)^2 (the cancellation the expanded quadratic form has to survive in
 * fp32), [1] = dead fraction of the 32-mixture tiles, [2] = largest per-dimension sigma ratio,
 * [3] = largest scaled coefficient of the fp16 layouts, [4] = 1 if sigma and weights are shared,
 * [5] = models, [6] = device, [7] = mixtures (of the largest model) that the hybrid form sends to the
 * direct-form vector engine because the expanded form would cancel for them (0: no hybrid form; the other
 * entries then describe the whole set).  A set that is ill conditioned because of a few mixtures only -- collapsed
 * components at the sigma floor -- is scored as two sub-sets whose per-frame values are merged by a log-add-exp. */
int sr_modelset_info(SRModelSet *set, double *out8);
int sr_modelset_dim(SRModelSet *set);

/* Utterance batch resident in HBM: either PCM (int16, concatenated, sample_offsets[U+1]) or
 * features (fp32 [n][dim], frame_offsets[U+1]). */
typedef struct SRBatch SRBatch;
SRBatch *sr_batch_from_pcm(const int16_t *pcm, const int64_t *sample_offsets, int n_utt);
SRBatch *sr_batch_from_pcm_f32(const float *pcm, const int64_t *sample_offsets, int n_utt);
SRBatch *sr_batch_from_features(const float *X, int64_t n_frames, int dim,
                                const int64_t *frame_offsets, int n_utt);
/* Overwrite the samples of a PCM batch in place (same utterance layout, same or fewer samples per
 * utterance is NOT supported: the layout must match exactly) -- the serving loop's H2D, no allocation.
 * Up to 4 MB of samples are copied to a page-locked staging area and the call returns with the transfer in
 * flight on the library's stream (the caller's buffer may be reused at once; whatever is queued next on the
 * stream -- sr_predict_pcm_batch, sr_mfcc_extract_batch -- runs behind it); larger updates return when the
 * samples are on the device. */
int sr_batch_update_pcm(SRBatch *b, const int16_t *pcm, int64_t n_samples);
/* New contents AND a new utterance layout in the same handle: device buffers are reused (they only
 * grow), so a serving loop whose batches change shape allocates nothing in steady state.  Up to 4 MB of samples travel as
 * sr_batch_update_pcm's do -- page-locked copy, transfer left in flight, the caller's buffers free on return -- together with
 * their offsets; the feature stage's and the scoring pass's tables of such a batch are rebuilt the same way (no host wait). */
int sr_batch_reset_pcm(SRBatch *b, const int16_t *pcm, const int64_t *sample_offsets, int n_utt);
/* The same for a feature batch: new frames and a new utterance layout (any dim) in the handle's buffers -- what keeps the
 * reference's per-utterance loop (gmmset.py:62-64, :95-99: one scoring call per utterance) free of allocations.  Up to 4 MB of
 * frames are copied to a page-locked staging area and the call returns with the transfer in flight on the library's stream (X is
 * the caller's again at once; sr_score_batch_set and whatever else is queued next runs behind it); larger refills return when the
 * frames are on the device. */
int sr_batch_reset_features(SRBatch *b, const float *X, int64_t n_frames, int dim,
                            const int64_t *frame_offsets, int n_utt);
void sr_batch_free(SRBatch *b);
int sr_batch_num_utterances(SRBatch *b);
int64_t sr_batch_num_rows(SRBatch *b);           /* samples (PCM) or frames (features) */
int sr_batch_dim(SRBatch *b);                    /* 0 for PCM */
int sr_batch_offsets(SRBatch *b, int64_t *offsets_out /* [U+1] */);
int sr_batch_download(SRBatch *b, float *out);   /* features -> host, [rows][dim] */

/* Score every utterance of a feature batch against every model of the set in one fused pass:
 * sums_out[U][S] = sum_t LL_s(x_t) (double), argmax_out[U] = first maximum (gmmset.py:62-64),
 * frame_ll_out (optional, may be NULL) = [S][n_frames] fp32 per-frame log-likelihoods. */
int sr_score_batch_set(SRModelSet *set, SRBatch *features, double *sums_out, int *argmax_out,
                       float *frame_ll_out, int flags);

/* MFCC extractor (src/feature/MFCC.py:20-41 constants; :49-79 chain; utils.py:24-31 deltas). */
typedef struct SRMfcc SRMfcc;
SRMfcc *sr_mfcc_create(double fs, double win_length_ms, double win_shift_ms, int fft_size,
                       int n_filters, int n_ceps, double pre_emphasis);
/* n_lpc > 0: append LPC-n_lpc columns to every frame (the reference's mix_feature,
 * src/feature/__init__.py:25-30 -> LPC.py:40-57; same framing, no deltas); 0 switches them off. */
int sr_mfcc_set_lpc(SRMfcc *m, int n_lpc);
void sr_mfcc_free(SRMfcc *m);
int sr_mfcc_frame_len(SRMfcc *m);
int sr_mfcc_frame_shift(SRMfcc *m);
int64_t sr_mfcc_num_frames(SRMfcc *m, int64_t n_samples);   /* MFCC.py:57; 0 if too short (:56) */
int sr_mfcc_tables(SRMfcc *m, double *window /*[L]*/, double *melbank /*[n_filters][fft/2+1]*/,
                   double *dct /*[n_ceps][n_filters]*/);    /* host float64 tables, for tests */
/* PCM batch -> feature batch (CMVN per utterance, then delta order nd in {0,1,2});
 * cmvn=0 skips the normalisation (raw cepstra, for tests). Returns a new device batch. */
SRBatch *sr_mfcc_extract_batch(SRMfcc *m, SRBatch *pcm, int nd, int cmvn);

/* Fused serving step on resident inputs: PCM batch -> MFCC -> CMVN/delta -> scoring -> argmax. */
int sr_predict_pcm_batch(SRMfcc *m, SRModelSet *set, SRBatch *pcm, int nd, double *sums_out,
                         int *argmax_out, int flags);

/* Every GPU of the node from ONE host process (no torch, no MPI, no collective): utterances are dealt
 * to n_slots slots by length, slot i lives on device i % sr_device_count() with its own replica of
 * the models and extractor tables, one host thread per slot runs upload -> MFCC -> CMVN/deltas -> all
 * models -> sums + argmax, and the per-utterance rows are gathered on the host (the reference:
 * Threadpool inside the call, gmm.cc:533-560, and multiprocessing.Pool over utterances,
 * test-gmm.py:128-133).  n_slots = 0 means one slot per visible device; more slots than devices is
 * legal: slots that share a device are one queue on it (the first of them takes their work and the others report 0
 * seconds; sr_set_option("multi_merge_same_device", 0) gives every slot its own thread and share, serialised by the
 * device's lock).  slot_seconds_out (optional, [n_slots]) receives each slot's wall time for the pass. */
typedef struct SRMulti SRMulti;
SRMulti *sr_multi_create(GMM *const *models, int n_models, double fs, double win_length_ms,
                         double win_shift_ms, int fft_size, int n_filters, int n_ceps,
                         double pre_emphasis, int n_slots);
void sr_multi_free(SRMulti *m);
/* Page-lock caller memory (hipHostRegister) -- a serving loop's PCM ring, say -- so that sr_multi_predict_pcm's copy
 * engines read it in place instead of going through a staging buffer; sr_host_unregister before it is freed. */
int sr_host_register(void *p, size_t bytes);
int sr_host_unregister(void *p);
int sr_multi_slots(SRMulti *m);
int sr_multi_slot_device(SRMulti *m, int slot);
int sr_multi_slot_numa_node(SRMulti *m, int slot);      /* where the slot's host thread was pinned in its last pass (-1: nowhere) */
int sr_multi_predict_pcm(SRMulti *m, const int16_t *pcm, const int64_t *sample_offsets, int n_utt,
                         int nd, double *sums_out /*[U][S]*/, int *argmax_out /*[U]*/,
                         double *slot_seconds_out, int flags);

/* Measured device-to-device copy rate (bytes read + written per second, GB/s) of a `bytes`-sized
 * buffer over `iters` copies on the library's stream: the HBM ceiling bench.py quotes beside the
 * nominal 8 TB/s. */
int sr_hbm_copy_gbps(size_t bytes, int iters, double *gbps_out);

/* Fixed-shape serving session: n_windows windows of window_samples int16 samples per tick.  Two
 * slots of pinned + device buffers and a second HIP stream: the H2D copy of tick i+1 overlaps the
 * kernels of tick i.  submit() returns after queueing (at most two ticks in flight); collect()
 * waits for the oldest tick and hands back sums[n_windows][S], argmax[n_windows] and the
 * device-side time from the start of its H2D to its last kernel. */
typedef struct SRStream SRStream;
#define SR_STREAM_GRAPH 0x100   /* flags: replay each tick's kernels + result copies as one hipGraph */
SRStream *sr_stream_create(SRMfcc *m, SRModelSet *set, int n_windows, int64_t window_samples, int nd,
                           int flags);
int sr_stream_submit(SRStream *s, const int16_t *pcm /* [n_windows][window_samples] */);
int sr_stream_collect(SRStream *s, double *sums_out, int *argmax_out, double *device_ms);
void sr_stream_free(SRStream *s);

/* Long-term spectral divergence of half-overlapped Hann windows -- the measure behind the reference's
 * voice-activity front end (src/filters/ltsd.py:32-64, which calls third-party pyssp.vad.ltsd; the
 * algorithm is restated from its published form, parity unpinned).  winsize = int(0.04644 * fs)
 * (ltsd.py:17,66-69), hop winsize/2, windows per signal = len/(winsize/2) - 1.
 * sr_ltsd_noise_spectrum: mean amplitude spectrum (bins 0..winsize/2) over all windows of `noise`.
 * sr_ltsd_compute: LTSD in dB of every window of every utterance of `pcm` against that spectrum;
 * ltsd_out holds win_offsets_out[U] floats, utterance u's windows at [win_offsets_out[u], win_offsets_out[u+1]). */
int64_t sr_ltsd_num_windows(int64_t n_samples, int winsize);
int sr_ltsd_noise_spectrum(SRBatch *noise, int winsize, float *avg_amp_out /*[winsize/2+1]*/);
int sr_ltsd_compute(SRBatch *pcm, int winsize, int order, const float *noise_amp /*[winsize/2+1]*/,
                    float *ltsd_out, int64_t *win_offsets_out /*[U+1]*/);

/* GPU EM / MAP on contiguous fp32 frames (the engine behind train_model*). Returns the number
 * of iterations run, negative on error. seed < 0 -> time-based. */
int sr_train_f32(GMM *gmm, GMM *ubm_or_null, const float *X, long n, int dim,
                 const struct Parameter *param, long seed);

/* Kernel timing by HIP events on the library's own stream. */
#define SR_T_SCORE 0
#define SR_T_MFCC 1
#define SR_T_CMVN 2
#define SR_T_FINALIZE 3
#define SR_T_ESTEP 4
#define SR_T_SCORE_REF 5   /* reference-offset pre-pass of the split-fp16 shared-sigma engine */
#define SR_T_COUNT 6
int sr_profile_enable(int on);
int sr_profile_reset(void);
int sr_profile_get(int kind, double *total_ms, long *launches);

/* Options (process-wide; value 0 = automatic unless stated).  The complete table -- key, values, default, the test that pins it --
 * is DESIGN.md section 8.  The ones a caller may want:
 *   "mfcc_precision" 2 (default: float64 spectrum, ln and DCT for every frame, the reference's arithmetic) | 0 (fp32 throughout,
 *                    ~1.5x faster feature stage; cepstra up to 2e-2 off on voices whose mel bands lie > 60 dB apart),
 *   "score_engine"   1 vector ALU | 3 split-bf16 | 4 split-bf16 shared-sigma | 5 split-fp16 | 6 split-fp16 shared-sigma,
 *   "flush_order"    2 | 1 (see SR_CLAMP_COMPAT),
 *   "reference_side_effects" 1: train_model / train_model_from_ubm print the parameter block (pygmm.cc:31-41) and the trainer
 *                    writes ./gmm-training-intermediate-dump.model after every second iteration (gmm.cc:622-630), as the reference
 *                    does unconditionally; 0 (default): neither,
 *   "multi_numa_bind" 0: sr_multi slot threads leave their CPU affinity alone (default 1: bound to the cores of their GPU's NUMA
 *                    node, intersected with the mask the thread already has),
 *   "multi_merge_same_device" 0: slots that share a device get a host thread each (default 1: one queue per device).
 * The rest select kernel variants for A/B runs and tests. */
int sr_set_option(const char *key, long value);
/* Counters of the partial-product path since the library was loaded: resolve calls, (frame tile, model) pairs
 * noted by the engines, frames re-evaluated.  Any pointer may be NULL. */
void sr_flush_stats(long *calls, long *pairs, long *frames);
/* Counters of the k-means initialiser's nearest-centre search since the library was loaded: full searches taken the fast
 * way (one fused multiply-add per point, centre and dimension; sr_set_option("kmeans_assign_engine", 1) turns it off) and
 * the points those left to the exact pass (near-ties; same results bit for bit).  Either pointer may be NULL. */
void sr_kmeans_fast_stats(long *passes, long *rechecked);
/* Diagnostic for the roofline record: runs v_mfma_f32_32x32x16_f16 chains, nothing else, on every SIMD of the current
 * device for about ms_target milliseconds and reports the executed TFLOP/s and the shader clock they ran at -- the matrix
 * throughput this device sustains under its power cap (MI355X: ~1.6 PFLOP/s at ~1.55 GHz against the 2.5 PFLOP/s that
 * 2.4 GHz would give).  Either pointer may be NULL.  0 on success. */
int sr_mfma_peak_probe(double ms_target, double *tflops, double *mhz);
/* The same chains fed the way the scoring kernels feed them: a fresh A fragment read from LDS for every MFMA (8 KiB parameter
 * images, random finite fp16 bit patterns), 8 random B fragments resident in registers, chains of 8 links -- the ceiling of a
 * kernel of that shape under the power cap (the probe above holds both operands in registers for the whole launch, a load no
 * real kernel presents).  Either pointer may be NULL.  0 on success. */
int sr_mfma_streamed_probe(double ms_target, double *tflops, double *mhz);
/* Name of the scoring kernel variant the last scoring call launched (for bench / logs). */
const char *sr_last_score_kernel(void);
/* Which statistics kernel the last EM / MAP iteration of this process ran: 0 none yet, 1 vector ALU, 2 fp64 matrix cores,
 * 3 fp64 matrix cores with the responsibilities on the 16-bit matrix cores (round 4; csrc/em.hip), 4 the whole fit in one launch
 * (speaker-sized models: <= 32 mixtures x <= 40 dims, <= 8192 frames; csrc/em_small.hip), 5 float64 iterations on the device (short
 * data, <= 8192 frames x <= 64 dims, a model of any size: MAP enrolment from a large UBM; csrc/em_f64.hip).  Tests and benches. */
int sr_last_em_stats_engine(void);
/* The first `count` values of the random stream train_model / load draw from when no seed is given: glibc's rand()
 * from its default seed, restated inside the library (the reference draws its initialisation from libc rand():
 * src/gmm/src/random.hh:22-25, gmm.hh:44, kmeansII.cc:94,133; csrc/kmeans_init.hip).  For checks. */
int sr_reference_rand_sample(int *out, int count);

#ifdef __cplusplus
}
#endif
#endif /* PYGMM_HIP_H */
