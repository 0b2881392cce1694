// Microbenchmark: W waves per SIMD, each looping { 8 dependent v_mfma_f32_32x32x16_f16 ; epilogue of 16 v_exp_f32 +
// adds on the accumulator } -- the h2s kernel's model-tile without LDS, DMA or barriers.  How well do the
// waves of a SIMD overlap one wave's chain with another's epilogue when left to the arbiter?
// Variants: PRIO (s_setprio 3 around the chain / around the epilogue), SPLIT (epilogue interleaved in the NEXT chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WPS, int VAR>
__global__ __launch_bounds__(WPS * 256, WPS) void k(float *out, unsigned long long *ticks, int iters, const f16x8 *src) {
    f16x8 a[8], b[8];
    for (int j = 0; j < 8; j++) { a[j] = src[(threadIdx.x + 64 * j) & 1023]; b[j] = src[(threadIdx.x * 3 + 64 * j + 7) & 1023]; }
    float ssum = 0.f;
    const f32x16 zero = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 prev = zero;
    unsigned long long tc = 0, te = 0;
    for (int i = 0; i < iters; i++) {
        f32x16 acc;
        unsigned long long s0 = 0, s1 = 0;
        if (VAR == 4) s0 = __builtin_amdgcn_s_memtime();
        if (VAR == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], u == 0 ? zero : acc, 0, 0, 0);
            if (VAR == 3) {       // the previous tile's epilogue inside this chain: 2 exps + 2 adds per link
                ssum += __builtin_amdgcn_exp2f(prev[2 * u] * 1e-3f - 3.f) + __builtin_amdgcn_exp2f(prev[2 * u + 1] * 1e-3f - 3.f);
            }
        }
        if (VAR == 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (VAR == 4) { s1 = __builtin_amdgcn_s_memtime(); tc += s1 - s0; }
        if (VAR == 2) __builtin_amdgcn_s_setprio(3);
        if (VAR != 3) {
            float e0 = 0.f, e1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) { e0 += __builtin_amdgcn_exp2f(acc[r] * 1e-3f - 3.f); e1 += __builtin_amdgcn_exp2f(acc[r + 1] * 1e-3f - 3.f); }
            ssum += e0 + e1;
        } else {
            prev = acc;
        }
        if (VAR == 2) __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(ssum));
        __builtin_amdgcn_sched_barrier(0);
        if (VAR == 4) te += __builtin_amdgcn_s_memtime() - s1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    if (VAR == 4 && (threadIdx.x & 63) == 0 && blockIdx.x == 0) { const int w = threadIdx.x >> 6; ticks[1 + 3 * w] = t1 - t0; ticks[2 + 3 * w] = tc; ticks[3 + 3 * w] = te; }
    out[blockIdx.x * WPS * 256 + threadIdx.x] = ssum;
}

template <int WPS, int VAR> void run(const char *name, float *out, unsigned long long *ticks, int iters, const f16x8 *src, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WPS, VAR>), dim3(grid), dim3(WPS * 256), 0, 0, out, ticks, 16, src);
    hipDeviceSynchronize();
    float best = 1e9; unsigned long long tk = 0;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<WPS, VAR>), dim3(grid), dim3(WPS * 256), 0, 0, out, ticks, iters, src);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost); }
    }
    // per SIMD: WPS tiles per `iters` loop trips
    printf("%-46s waves/SIMD %d grid %3d  %.3f ms  %.1f cycles per tile per SIMD  (%.2f GHz)  MFMA pipe %.0f %%\n", name, WPS, grid, best,
           (double)tk / iters / WPS, (double)tk / (best * 1e6), 100.0 * 256.0 * WPS * iters / (double)tk);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 8 * 64);
    unsigned short h[8192]; unsigned x = 12345u;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x3800); }
    f16x8 *src; hipMalloc(&src, 16384); hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    {   // per-wave breakdown on 16 CUs (2.4 GHz): waves w, w + 4, w + 8, ... share SIMD (w & 3)
        for (int wps = 2; wps <= 4; wps++) {
            if (wps == 2) hipLaunchKernelGGL((k<2, 4>), dim3(16), dim3(512), 0, 0, out, ticks, iters, src);
            if (wps == 3) hipLaunchKernelGGL((k<3, 4>), dim3(16), dim3(768), 0, 0, out, ticks, iters, src);
            if (wps == 4) hipLaunchKernelGGL((k<4, 4>), dim3(16), dim3(1024), 0, 0, out, ticks, iters, src);
            hipDeviceSynchronize();
            unsigned long long tk[64]; hipMemcpy(tk, ticks, 8 * (1 + 3 * 4 * wps), hipMemcpyDeviceToHost);
            printf("%d waves/SIMD, waves of SIMD 0: per tile [total | in chain | in epilogue]:", wps);
            for (int g = 0; g < wps; g++) printf("  wave %d [%.0f | %.0f | %.0f]", 4 * g, (double)tk[1 + 3 * 4 * g] / iters, (double)tk[2 + 3 * 4 * g] / iters, (double)tk[3 + 3 * 4 * g] / iters);
            printf("\n");
        }
    }
    for (int grid : {16}) {
        run<1, 0>("chain then epilogue", out, ticks, iters, src, grid);
        run<2, 0>("chain then epilogue", out, ticks, iters, src, grid);
        run<3, 0>("chain then epilogue", out, ticks, iters, src, grid);
        run<4, 0>("chain then epilogue", out, ticks, iters, src, grid);
        run<2, 1>("setprio 3 around the chain", out, ticks, iters, src, grid);
        run<3, 1>("setprio 3 around the chain", out, ticks, iters, src, grid);
        run<2, 2>("setprio 3 around the epilogue", out, ticks, iters, src, grid);
        run<3, 2>("setprio 3 around the epilogue", out, ticks, iters, src, grid);
        run<1, 3>("previous epilogue inside the chain", out, ticks, iters, src, grid);
        run<2, 3>("previous epilogue inside the chain", out, ticks, iters, src, grid);
        run<3, 3>("previous epilogue inside the chain", out, ticks, iters, src, grid);
    }
    return 0;
}
