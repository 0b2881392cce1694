// Microbenchmark: does a wave whose next MFMA waits for the (busy) matrix pipe block the vector-ALU issue of
// the other waves of its SIMD?  One workgroup of 12 waves per CU (three per SIMD).  NA of the three groups run
// dependent MFMA chains, the rest run exp/add loops; every wave reports its own cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NA, int NB, int NACC = 1, int SWAP = 0, int PRIO = 0>   // PRIO: s_setprio of the MFMA waves; SWAP: the exp/add groups come first (older waves); NACC accumulators per MFMA wave, used in turn; groups 0..NA-1: MFMA chains; groups NA..NA+NB-1: exp/add; others idle
__global__ __launch_bounds__(768, 3) void k(float *out, unsigned long long *ticks, int iters, const f16x8 *src) {
    const int wave = threadIdx.x >> 6;
    const int g0 = wave >> 2;
    const int grp = SWAP ? (g0 < NB ? NA + g0 : g0 < NA + NB ? g0 - NB : g0) : g0;     // SWAP: physical groups 0..NB-1 do the exp/add role
    float res = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (grp < NA) {
        f16x8 a[8], b[8];
        for (int j = 0; j < 8; j++) { a[j] = src[(threadIdx.x + 64 * j) & 1023]; b[j] = src[(threadIdx.x * 3 + 64 * j + 7) & 1023]; }
        __builtin_amdgcn_s_setprio(PRIO);
        f32x16 acc[NACC];
        for (int c = 0; c < NACC; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], acc[u % NACC], 0, 0, 0);
        }
        res = acc[0][0] + acc[NACC - 1][9];
    } else if (grp < NA + NB) {
        float x[16];
        for (int j = 0; j < 16; j++) x[j] = -0.001f * (threadIdx.x + j);
        float e0 = 0.f, e1 = 0.f;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) { e0 += __builtin_amdgcn_exp2f(x[r] + e1 * 1e-30f); e1 += __builtin_amdgcn_exp2f(x[r + 1]); }
            asm volatile("" : "+v"(e0), "+v"(e1));
        }
        res = e0 + e1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
    out[blockIdx.x * 768 + threadIdx.x] = res;
}

template <int NA, int NB, int NACC = 1, int SWAP = 0, int PRIO = 0> void run(float *out, unsigned long long *ticks, int iters, const f16x8 *src) {
    hipLaunchKernelGGL((k<NA, NB, NACC, SWAP, PRIO>), dim3(16), dim3(768), 0, 0, out, ticks, 16, src);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NA, NB, NACC, SWAP, PRIO>), dim3(16), dim3(768), 0, 0, out, ticks, iters, src);
    hipDeviceSynchronize();
    unsigned long long tk[12]; hipMemcpy(tk, ticks, 96, hipMemcpyDeviceToHost);
    printf("%d MFMA waves + %d exp/add waves per SIMD: cycles per iteration (8 MFMA | 16 exp + 16 add) of the waves on SIMD 0:", NA, NB);
    if (SWAP) printf(" (exp/add waves are the OLDER ones)");
    if (PRIO) printf(" (MFMA waves at s_setprio %d)", PRIO);
    for (int g = 0; g < NA + NB; g++) { const int pg = SWAP ? (g < NA ? NB + g : g - NA) : g; printf("  %s %.0f", g < NA ? "mfma" : "valu", (double)tk[4 * pg] / iters); }
    unsigned long long mx = 0; for (int g = 0; g < NA; g++) { const int pg = SWAP ? NB + g : g; if (tk[4 * pg] > mx) mx = tk[4 * pg]; }
    if (NA) printf("   [%d accumulator(s) per MFMA wave: %.1f cycles per MFMA on the pipe]", NACC, (double)mx / iters / 8 / NA);
    printf("\n");
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; hipMalloc(&out, 16 * 768 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 96);
    unsigned short h[8192]; unsigned x = 12345u;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x3800); }
    f16x8 *src; hipMalloc(&src, 16384); hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    run<1, 0>(out, ticks, iters, src);
    run<0, 1>(out, ticks, iters, src);
    run<0, 2>(out, ticks, iters, src);
    run<1, 1>(out, ticks, iters, src);
    run<1, 2>(out, ticks, iters, src);
    run<2, 0>(out, ticks, iters, src);
    run<2, 1>(out, ticks, iters, src);
    run<3, 0>(out, ticks, iters, src);
    run<1, 0, 2>(out, ticks, iters, src);
    run<2, 0, 2>(out, ticks, iters, src);
    run<3, 0, 2>(out, ticks, iters, src);
    run<2, 1, 2>(out, ticks, iters, src);
    run<1, 1, 1, 1>(out, ticks, iters, src);
    run<1, 2, 1, 1>(out, ticks, iters, src);
    run<2, 1, 1, 1>(out, ticks, iters, src);
    run<1, 2, 1, 1, 1>(out, ticks, iters, src);
    run<1, 2, 1, 1, 3>(out, ticks, iters, src);
    return 0;
}
