// Microbenchmark: ONE wave keeps the matrix pipe busy by itself -- two independent accumulator chains (two 32-frame
// column tiles against the same A fragments) alternate, and the epilogue of the PREVIOUS image pair (32 v_exp_f32 +
// 32 v_add_f32) is spread between the 16 MFMAs of the current one.  W waves per SIMD, cycles per image pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WPS, int VAR>   // VAR 0: chains, then both epilogues; 1: epilogue of the previous pair interleaved (sched_group_barrier); 2: same, compiler's order
__global__ __launch_bounds__(WPS * 256, WPS) void k(float *out, unsigned long long *ticks, int iters, const f16x8 *src) {
    f16x8 a[8], b0[8], b1[8];
    for (int j = 0; j < 8; j++) { a[j] = src[(threadIdx.x + 64 * j) & 1023]; b0[j] = src[(threadIdx.x * 3 + 64 * j + 7) & 1023]; b1[j] = src[(threadIdx.x * 5 + 64 * j + 3) & 1023]; }
    float s0 = 0.f, s1 = 0.f;
    const f32x16 zero = {0};
    f32x16 p0 = zero, p1 = zero;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        f32x16 c0, c1;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b0[u], u == 0 ? zero : c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b1[u], u == 0 ? zero : c1, 0, 0, 0);
        }
        if (VAR == 0) {
            __builtin_amdgcn_sched_barrier(0);
            float e0 = 0.f, e1 = 0.f, f0 = 0.f, f1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                e0 += __builtin_amdgcn_exp2f(c0[r]); e1 += __builtin_amdgcn_exp2f(c0[r + 1]);
                f0 += __builtin_amdgcn_exp2f(c1[r]); f1 += __builtin_amdgcn_exp2f(c1[r + 1]);
            }
            s0 += e0 + e1; s1 += f0 + f1;
            asm volatile("" : "+v"(s0), "+v"(s1));
            __builtin_amdgcn_sched_barrier(0);
        } else {
            float e0 = 0.f, e1 = 0.f, f0 = 0.f, f1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                e0 += __builtin_amdgcn_exp2f(p0[r]); e1 += __builtin_amdgcn_exp2f(p0[r + 1]);
                f0 += __builtin_amdgcn_exp2f(p1[r]); f1 += __builtin_amdgcn_exp2f(p1[r + 1]);
            }
            s0 += e0 + e1; s1 += f0 + f1;
            if (VAR == 1) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU (2 exp + 2 add)
                }
            }
            asm volatile("" : "+v"(s0), "+v"(s1));
            p0 = c0; p1 = c1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[threadIdx.x >> 6] = t1 - t0;
    out[blockIdx.x * WPS * 256 + threadIdx.x] = s0 + s1 + p0[3] + p1[5];
}

template <int WPS, int VAR> void run(const char *name, float *out, unsigned long long *ticks, int iters, const f16x8 *src, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WPS, VAR>), dim3(grid), dim3(WPS * 256), 0, 0, out, ticks, 16, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WPS, VAR>), dim3(grid), dim3(WPS * 256), 0, 0, out, ticks, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long tk[16]; hipMemcpy(tk, ticks, 8 * 4 * WPS, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < 4 * WPS; w++) if (tk[w] > mx) mx = tk[w];
    printf("%-44s waves/SIMD %d grid %3d  %.3f ms  %.0f cycles per 16-MFMA image pair per SIMD (slowest wave), %.1f ns;  MFMA pipe %.0f %%\n", name, WPS, grid, ms,
           (double)mx / iters / WPS, ms * 1e6 / iters / WPS, 100.0 * 512.0 * WPS * iters / (double)mx);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 8 * 64);
    unsigned short h[8192]; unsigned x = 12345u;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x2800); }   // small magnitudes: sums stay finite
    f16x8 *src; hipMalloc(&src, 16384); hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    for (int grid : {16, 256}) {
        run<1, 0>("two chains, then both epilogues", out, ticks, iters, src, grid);
        run<2, 0>("two chains, then both epilogues", out, ticks, iters, src, grid);
        run<1, 1>("previous epilogues inside the chains (1:4)", out, ticks, iters, src, grid);
        run<2, 1>("previous epilogues inside the chains (1:4)", out, ticks, iters, src, grid);
        run<1, 2>("previous epilogues, compiler's order", out, ticks, iters, src, grid);
        run<2, 2>("previous epilogues, compiler's order", out, ticks, iters, src, grid);
    }
    return 0;
}
