// Microbenchmark: cadence of a dependent chain of v_mfma_f32_32x32x16_f16 on one SIMD, measured in
// s_memtime ticks and wall time: same operand registers for every link vs 8 distinct A/B register sets
// (as the scoring kernels have), with and without a co-resident vector-ALU wave, 1 or 2 workgroups' worth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int DISTINCT, int PARTNER>
__global__ __launch_bounds__(512, 2) void k(float *out, unsigned long long *ticks, int iters, const f16x8 *src) {
    const int wave = threadIdx.x >> 6;
    const int grp = wave >> 2;
    float res = 0.f;
    __shared__ uint4 lds[8 * 64];
    for (int i = threadIdx.x; i < 8 * 64; i += 512) lds[i] = make_uint4(i, i * 3, i * 5, i * 7);
    __syncthreads();
    if (grp == 0) {
        f16x8 a[8], b[8];
        for (int j = 0; j < 8; j++) { a[j] = src[(threadIdx.x + 64 * j) & 1023]; b[j] = src[(threadIdx.x * 3 + 64 * j + 7) & 1023]; }
        f32x16 acc = {0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(DISTINCT ? a[u] : a[0], DISTINCT ? b[u] : b[0], acc, 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
        res = acc[0] + acc[9];
    } else if (PARTNER) {
        float x[16];
        for (int j = 0; j < 16; j++) x[j] = -0.001f * (threadIdx.x + j);
        float e0 = 0.f, e1 = 0.f;
        uint4 fr[8];
        unsigned sink = 0;
        for (int i = 0; i < iters; i++) {
            if (PARTNER >= 2) {
#pragma unroll
                for (int u = 0; u < 8; u++) fr[u] = lds[u * 64 + (threadIdx.x & 63)];
            }
            if (PARTNER != 3) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) { e0 += __builtin_amdgcn_exp2f(x[r] + e1 * 1e-30f); e1 += __builtin_amdgcn_exp2f(x[r + 1]); }
            }
            asm volatile("" : "+v"(e0), "+v"(e1));
            if (PARTNER >= 2) {
#pragma unroll
                for (int u = 0; u < 8; u++) sink += fr[u].x ^ fr[u].w;
                asm volatile("" : "+v"(sink));
            }
        }
        res = e0 + e1 + (float)sink;
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int D, int P> void run(const char *name, float *out, unsigned long long *ticks, int iters, const f16x8 *src, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<D, P>), dim3(grid), dim3(512), 0, 0, out, ticks, 16, src);
    hipDeviceSynchronize();
    float best = 1e9; unsigned long long tk = 0;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<D, P>), dim3(grid), dim3(512), 0, 0, out, ticks, iters, src);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost); }
    }
    printf("%-52s grid %4d  %.3f ms  %.2f ns/MFMA  %.1f ticks/MFMA  (%.2f ticks/ns)\n", name, grid, best, best * 1e6 / iters / 8, (double)tk / iters / 8,
           (double)tk / (best * 1e6));
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; hipMalloc(&out, 512 * 512 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 8);
    unsigned short h[8192]; unsigned x = 12345u;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x3800); }   // fp16 in [0.5, 1) u [-1, -0.5), random mantissas
    f16x8 *src; hipMalloc(&src, 16384); hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    for (int grid : {256, 16}) {     // the whole chip (power-capped) and 16 CUs (not)
        run<0, 0>("same operands, alone", out, ticks, iters, src, grid);
        run<1, 0>("8 distinct operand sets, alone", out, ticks, iters, src, grid);
        run<0, 1>("same operands, + exp/add wave on the SIMD", out, ticks, iters, src, grid);
        run<1, 1>("8 distinct operand sets, + exp/add wave on the SIMD", out, ticks, iters, src, grid);
        run<1, 2>("8 distinct, + (8 ds_read_b128 + exp/add) wave", out, ticks, iters, src, grid);
        run<1, 3>("8 distinct, + (8 ds_read_b128 only) wave", out, ticks, iters, src, grid);
    }
    return 0;
}
