/* examples/predict_pcm.c -- the C ABI from plain C (C99, no HIP, no C++): what a non-Python host of
 * the reference's hot path would write against include/pygmm_hip.h.
 *
 *   gcc -std=c99 -Iinclude examples/predict_pcm.c -o predict_pcm \
 *       -Lspeaker-recognition_amd/lib -l:pygmm.so -Wl,-rpath,$PWD/speaker-recognition_amd/lib -lm
 *
 * Builds S random 39-dimensional speaker models, writes them through the reference's text model
 * format (dump/load), synthesises U utterances of 16 kHz noise + a per-speaker tone, and runs
 * PCM -> MFCC(+delta, delta-delta) -> all speakers -> per-utterance sums and arg max in one call.
 * Prints the decisions; exits non-zero on any library error. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pygmm_hip.h"

static double frand(unsigned *s) {
    *s = *s * 1664525u + 1013904223u;
    return ((*s >> 8) & 0xFFFFFF) / (double)0x1000000;
}

int main(void) {
    enum { S = 4, K = 32, D = 39, U = 3, FS = 16000, N = 2 * FS };
    unsigned seed = 7;
    if (sr_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", sr_last_error());
        return 2;
    }
    /* speaker models from arrays, then a round trip through the reference's text format */
    GMM *models[S];
    for (int s = 0; s < S; s++) {
        static double w[K], mu[K * D], sg[K * D];
        double tot = 0;
        for (int k = 0; k < K; k++) tot += (w[k] = 0.2 + frand(&seed));
        for (int k = 0; k < K; k++) w[k] /= tot;
        for (int i = 0; i < K * D; i++) {
            mu[i] = 2.0 * frand(&seed) - 1.0 + 0.3 * s;
            sg[i] = 0.5 + frand(&seed);
        }
        GMM *g = sr_gmm_from_arrays(K, D, w, mu, sg);
        if (!g) { fprintf(stderr, "sr_gmm_from_arrays: %s\n", sr_last_error()); return 1; }
        char path[64];
        snprintf(path, sizeof path, "/tmp/predict_pcm_model_%d.txt", s);
        dump(g, path);                              /* pygmm.hh:31 */
        sr_free_gmm(g);
        models[s] = load(path);                     /* pygmm.hh:29 */
        if (!models[s] || get_dim(models[s]) != D || get_nr_mixtures(models[s]) != K) {
            fprintf(stderr, "load: %s\n", sr_last_error());
            return 1;
        }
    }
    SRModelSet *set = sr_modelset_create(models, S);
    if (!set) { fprintf(stderr, "sr_modelset_create: %s\n", sr_last_error()); return 1; }

    /* U utterances of 2 s, concatenated int16 + offsets */
    int16_t *pcm = malloc(sizeof(int16_t) * U * N);
    int64_t off[U + 1];
    for (int u = 0; u <= U; u++) off[u] = (int64_t)u * N;
    for (int u = 0; u < U; u++)
        for (int i = 0; i < N; i++)
            pcm[u * N + i] = (int16_t)(3000.0 * sin(2 * M_PI * (200.0 + 90.0 * u) * i / FS) + 600.0 * (frand(&seed) - 0.5));
    SRBatch *batch = sr_batch_from_pcm(pcm, off, U);
    SRMfcc *mf = sr_mfcc_create(FS, 25, 10, 2048, 50, 13, 0.95);   /* MFCC.py:115-121 keywords */
    if (!batch || !mf) { fprintf(stderr, "setup: %s\n", sr_last_error()); return 1; }

    double sums[U * S];
    int best[U];
    if (sr_predict_pcm_batch(mf, set, batch, /*nd=*/2, sums, best, SR_CLAMP_COMPAT) != 0) {
        fprintf(stderr, "sr_predict_pcm_batch: %s\n", sr_last_error());
        return 1;
    }
    for (int u = 0; u < U; u++) {
        printf("utterance %d -> speaker %d   sums:", u, best[u]);
        int arg = 0;
        for (int s = 0; s < S; s++) {
            printf(" %.1f", sums[u * S + s]);
            if (sums[u * S + s] > sums[u * S + arg]) arg = s;
        }
        printf("\n");
        if (arg != best[u] || !isfinite(sums[u * S])) { fprintf(stderr, "inconsistent result\n"); return 1; }
    }
    printf("kernel: %s\n", sr_last_score_kernel());
    sr_batch_free(batch);
    sr_mfcc_free(mf);
    sr_modelset_free(set);
    for (int s = 0; s < S; s++) sr_free_gmm(models[s]);
    free(pcm);
    return 0;
}
