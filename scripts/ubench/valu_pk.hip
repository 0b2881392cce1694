// Microbenchmark 4: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 (wave64, 8 independent
// chains per lane, 4 waves/SIMD): flops/cycle/CU for each -- does packing fp32 buy anything on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    v2 a[8];
    for (int i = 0; i < 8; i++) a[i] = (v2){threadIdx.x * 1e-3f + i, 1.0f + i};
    const v2 m = {s, s * 0.5f}, c = {1e-3f, 2e-3f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) {            // scalar: two v_fma_f32 per pair
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
                } else if (MODE == 1) {     // packed fma: one instruction per pair
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                } else {                    // packed add
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                }
            }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, double flops_per_pair) {
    const int grid = 256 * 4, iters = 4000;
    float *out; hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, 10, 0.999f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double pairs = (double)grid * 256 * iters * 64;      // 8 x 8 pair-updates per iteration per lane
    printf("%-14s %.3f ms  %.1f TFLOP/s  (%.2f G pair-updates/s)\n", name, best, pairs * flops_per_pair / (best * 1e-3) / 1e12, pairs / (best * 1e-3) / 1e9);
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32 x2", 4);
    run<1>("v_pk_fma_f32", 4);
    run<2>("v_pk_add_f32", 2);
    return 0;
}
