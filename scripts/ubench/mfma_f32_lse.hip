// Microbenchmark 2: MFMA f32 32x32x2 stream shaped like the scoring kernel: per "tile" 40*FT MFMAs
// with 40*FT distinct B registers, accumulators zeroed, then an LSE-like vector-ALU epilogue
// (16 rows: max, sub, exp2, add) that consumes the accumulators.  Reports TFLOP/s at 1..3 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FT, int MODE>   // MODE 0: MFMA only (acc kept live); 1: + LSE epilogue; 2: + LSE and B from 80 regs
__global__ __launch_bounds__(256) void k(float *out, const float *bin, int tiles) {
    __shared__ float4 lds[640];
    for (int i = threadIdx.x; i < 640; i += 256) lds[i] = make_float4(i * 1e-3f, 1.f, 0.5f, 0.25f);
    __syncthreads();
    constexpr int KQ = 10;
    float breg[FT][KQ * 4];
    for (int ft = 0; ft < FT; ft++)
        for (int i = 0; i < KQ * 4; i++) breg[ft][i] = bin[(threadIdx.x * 7 + i * 3 + ft) & 1023];
    const int lane = threadIdx.x & 63;
    float m[FT], ssum[FT];
    for (int ft = 0; ft < FT; ft++) { m[ft] = -1e30f; ssum[ft] = 0.f; }
    for (int t = 0; t < tiles; t++) {
        f32x16 acc[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ft++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ft][r] = 0.f;
#pragma unroll
        for (int kq = 0; kq < KQ; kq++) {
            const float4 a = lds[kq * 64 + lane];
#pragma unroll
            for (int ft = 0; ft < FT; ft++) {
                acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, breg[ft][4 * kq + 0], acc[ft], 0, 0, 0);
                acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, breg[ft][4 * kq + 1], acc[ft], 0, 0, 0);
                acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, breg[ft][4 * kq + 2], acc[ft], 0, 0, 0);
                acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, breg[ft][4 * kq + 3], acc[ft], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ft = 0; ft < FT; ft++) {
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) asm volatile("" ::"v"(acc[ft][r]));
                m[ft] = acc[ft][3];
            } else {
                float mx = acc[ft][0];
#pragma unroll
                for (int r = 1; r < 16; r++) mx = fmaxf(mx, acc[ft][r]);
                const float mn = fmaxf(m[ft], mx);
                float e = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) e += __builtin_amdgcn_exp2f(acc[ft][r] - mn);
                ssum[ft] = fmaf(ssum[ft], __builtin_amdgcn_exp2f(m[ft] - mn), e);
                m[ft] = mn;
            }
        }
    }
    float s = 0;
    for (int ft = 0; ft < FT; ft++) s += m[ft] + ssum[ft];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FT, int MODE>
void run(const char *name, int blocks_per_cu, const float *bin) {
    int tiles = 400, grid = 256 * blocks_per_cu;
    float *out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FT, MODE>), dim3(grid), dim3(256), 0, 0, out, bin, 4);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<FT, MODE>), dim3(grid), dim3(256), 0, 0, out, bin, tiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double mfmas = (double)grid * 4 * tiles * 40 * FT;
    double tf = mfmas * 2.0 * 32 * 32 * 2 / (best * 1e-3) / 1e12;
    printf("%-34s FT=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s (%.1f%%)\n", name, FT, blocks_per_cu, best, tf, tf / 1.573);
    hipFree(out);
}

int main() {
    float *bin; hipMalloc(&bin, 4096); hipMemset(bin, 0, 4096);
    for (int w = 1; w <= 3; w++) {
        run<2, 0>("MFMA only, acc zeroed per tile", w, bin);
        run<2, 1>("MFMA + LSE epilogue", w, bin);
        run<1, 1>("MFMA + LSE epilogue", w, bin);
    }
    return 0;
}
