// Microbenchmark: do the matrix pipe and the vector ALU of ONE SIMD overlap across waves?
// One workgroup of 8 waves per CU (two per SIMD: waves w and w + 4).  Group A (waves 0-3) runs
// dependent chains of v_mfma_f32_32x32x16_f16; group B (waves 4-7) runs v_exp_f32 + v_add_f32 (the h2s
// epilogue's mix) or plain v_fma_f32.  Timed: A alone, B alone, both.  max(A, B) = overlap, A + B = none.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE_A, int MODE_B>   // MODE_A: 0 off, 1 one dependent chain, 2 two interleaved chains; MODE_B: 0 off, 1 exp+add, 2 fma, 3 exp only
__global__ __launch_bounds__(512, 2) void k(float *out, int iters) {
    const int wave = threadIdx.x >> 6;
    const int grp = wave >> 2;
    float res = 0.f;
    if (grp == 0) {
        if (MODE_A) {
            f16x8 a, b;
            for (int j = 0; j < 8; j++) { a[j] = (_Float16)(1.0f + 0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.5f + 0.002f * (threadIdx.x * 3 + j)); }
            f32x16 acc0 = {0}, acc1 = {0};
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                    if (MODE_A == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
                }
            }
            res = acc0[0] + acc0[7] + acc1[3];
        }
    } else {
        if (MODE_B) {
            float x[16];
            for (int j = 0; j < 16; j++) x[j] = -0.001f * (threadIdx.x + j);
            float e0 = 0.f, e1 = 0.f;
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    if (MODE_B == 1) { e0 += __builtin_amdgcn_exp2f(x[r] + e1 * 1e-30f); e1 += __builtin_amdgcn_exp2f(x[r + 1]); }
                    if (MODE_B == 2) { e0 = fmaf(e0, 0.999f, x[r]); e1 = fmaf(e1, 0.999f, x[r + 1]); e0 = fmaf(e0, 0.999f, x[r]); e1 = fmaf(e1, 0.999f, x[r + 1]); }
                    if (MODE_B == 3) { x[r] = __builtin_amdgcn_exp2f(x[r]); x[r + 1] = __builtin_amdgcn_exp2f(x[r + 1]); }
                }
                asm volatile("" : "+v"(e0), "+v"(e1));
            }
            res = e0 + e1 + x[0] + x[5];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int A, int B> float run(float *out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, out, 16);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; hipMalloc(&out, 256 * 512 * 4);
    printf("iters %d: per iteration group A = 8 (or 16) MFMA 32x32x16 f16, group B = 16 VALU ops (exp+add pairs: 16 exp + 16 add)\n", iters);
    float a1 = run<1, 0>(out, iters), a2 = run<2, 0>(out, iters);
    float b1 = run<0, 1>(out, iters), b2 = run<0, 2>(out, iters), b3 = run<0, 3>(out, iters);
    printf("A1 alone (1 chain, 8 MFMA/iter)          %.3f ms  = %.1f ns/iter\n", a1, a1 * 1e6 / iters);
    printf("A2 alone (2 chains, 16 MFMA/iter)        %.3f ms  = %.1f ns/iter\n", a2, a2 * 1e6 / iters);
    printf("B1 alone (16 exp + 16 add / iter)        %.3f ms  = %.1f ns/iter\n", b1, b1 * 1e6 / iters);
    printf("B2 alone (32 fma / iter)                 %.3f ms  = %.1f ns/iter\n", b2, b2 * 1e6 / iters);
    printf("B3 alone (16 exp / iter)                 %.3f ms  = %.1f ns/iter\n", b3, b3 * 1e6 / iters);
    float ab;
    ab = run<1, 1>(out, iters); printf("A1 + B1 together  %.3f ms   (max %.3f, sum %.3f)\n", ab, a1 > b1 ? a1 : b1, a1 + b1);
    ab = run<1, 2>(out, iters); printf("A1 + B2 together  %.3f ms   (max %.3f, sum %.3f)\n", ab, a1 > b2 ? a1 : b2, a1 + b2);
    ab = run<1, 3>(out, iters); printf("A1 + B3 together  %.3f ms   (max %.3f, sum %.3f)\n", ab, a1 > b3 ? a1 : b3, a1 + b3);
    ab = run<2, 1>(out, iters); printf("A2 + B1 together  %.3f ms   (max %.3f, sum %.3f)\n", ab, a2 > b1 ? a2 : b1, a2 + b1);
    ab = run<2, 2>(out, iters); printf("A2 + B2 together  %.3f ms   (max %.3f, sum %.3f)\n", ab, a2 > b2 ? a2 : b2, a2 + b2);
    return 0;
}
