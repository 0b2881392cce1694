// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on this chip, (a) register operands only,
// (b) A fragments re-read from LDS every 4 MFMAs as the scoring kernel does, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS_A>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ float4 lds[640];
    for (int i = threadIdx.x; i < 640; i += 256) lds[i] = make_float4(i * 1e-3f, 1.f, 0.5f, 0.25f);
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    float b[8];
    for (int i = 0; i < 8; i++) b[i] = threadIdx.x * 1e-4f + i;
    float4 av = make_float4(1.f, 2.f, 3.f, 4.f);
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int kq = 0; kq < 10; kq++) {
            if (LDS_A) av = lds[kq * 64 + lane];
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[(a * 4 + 0) & 7], acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[(a * 4 + 1) & 7], acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[(a * 4 + 2) & 7], acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[(a * 4 + 3) & 7], acc[a], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS_A>
void run(const char *name, int blocks_per_cu) {
    int iters = 2000;
    int grid = 256 * blocks_per_cu;
    float *out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS_A>), dim3(grid), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, LDS_A>), dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double mfmas = (double)grid * 4 * iters * 10 * 4 * NACC;
    double tf = mfmas * 2.0 * 32 * 32 * 2 / (best * 1e-3) / 1e12;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, blocks_per_cu, best, tf, tf / 1.573);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; w++) {
        run<2, false>("reg operands, 2 acc", w);
        run<2, true>("A from LDS, 2 acc", w);
    }
    run<1, false>("reg operands, 1 acc", 2);
    run<4, false>("reg operands, 4 acc", 1);
    return 0;
}
