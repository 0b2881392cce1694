// Microbenchmark: W waves per SIMD each take model-tiles { 8-MFMA chain ; 16 exp + 16 add epilogue } from a per-SIMD
// counter until N are done (no tail: the waves of a SIMD finish together), no LDS traffic, no DMA, no barriers.
// Reports cycles per tile per SIMD for several ways of writing the wave's program.
//   0  chain (one accumulator), then epilogue                      (the h2s kernel's order)
//   1  chain on two accumulators used in turn (k even / odd), epilogue adds them first
//   2  two tiles per trip: 16 MFMAs (two chains in turn), then both epilogues
//   3  as 0, epilogue = 16 exps first, then the adds
//   4  as 0 with s_sleep 1 between chain and epilogue (lets a waiting wave in)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float epilogue(const f32x16 &c) {
    float e0 = 0.f, e1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) { e0 += __builtin_amdgcn_exp2f(c[r]); e1 += __builtin_amdgcn_exp2f(c[r + 1]); }
    return e0 + e1;
}

template <int WPS, int VAR>
__global__ __launch_bounds__(WPS * 256, WPS) void k(float *out, unsigned long long *ticks, int n_tiles, const f16x8 *src) {
    __shared__ int counter[4];
    if (threadIdx.x < 4) counter[threadIdx.x] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, simd = wave & 3;
    f16x8 a[8], b[8], b2[8];
    for (int j = 0; j < 8; j++) { a[j] = src[(threadIdx.x + 64 * j) & 1023]; b[j] = src[(threadIdx.x * 3 + 64 * j + 7) & 1023]; b2[j] = src[(threadIdx.x * 5 + 64 * j + 1) & 1023]; }
    float ssum = 0.f;
    const f32x16 zero = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int done = 0;
    while (true) {
        int t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(&counter[simd], VAR == 2 ? 2 : 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= n_tiles) break;
        done++;
        if (VAR == 0 || VAR == 3 || VAR == 4) {
            f32x16 c;
#pragma unroll
            for (int u = 0; u < 8; u++) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], u == 0 ? zero : c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (VAR == 4) __builtin_amdgcn_s_sleep(1);
            if (VAR == 3) {
                float e[16];
#pragma unroll
                for (int r = 0; r < 16; r++) e[r] = __builtin_amdgcn_exp2f(c[r]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; r += 4) ssum += (e[r] + e[r + 1]) + (e[r + 2] + e[r + 3]);
            } else {
                ssum += epilogue(c);
            }
        } else if (VAR == 1) {
            f32x16 c0, c1;
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], u == 0 ? zero : c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u + 1], b[u + 1], u == 0 ? zero : c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            ssum += epilogue(c0 + c1);
        } else if (VAR == 2) {
            f32x16 c0, c1;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], u == 0 ? zero : c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b2[u], u == 0 ? zero : c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            ssum += epilogue(c0);
            ssum += epilogue(c1);
        }
        asm volatile("" : "+v"(ssum));
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { ticks[2 * wave] = t1 - t0; ticks[2 * wave + 1] = done; }
    out[blockIdx.x * WPS * 256 + threadIdx.x] = ssum;
}

template <int WPS, int VAR> void run(const char *name, float *out, unsigned long long *ticks, int n_tiles, const f16x8 *src) {
    hipLaunchKernelGGL((k<WPS, VAR>), dim3(16), dim3(WPS * 256), 0, 0, out, ticks, 64, src);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<WPS, VAR>), dim3(16), dim3(WPS * 256), 0, 0, out, ticks, n_tiles, src);
    hipDeviceSynchronize();
    unsigned long long tk[32]; hipMemcpy(tk, ticks, 8 * 2 * 4 * WPS, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < 4 * WPS; w++) if (tk[2 * w] > mx) mx = tk[2 * w];
    printf("%-58s waves/SIMD %d: %5.0f cycles per tile per SIMD, MFMA pipe %3.0f %%;  tiles done by the waves of SIMD 0:", name, WPS, (double)mx / n_tiles, 100.0 * 256.0 * n_tiles / (double)mx);
    for (int g = 0; g < WPS; g++) printf(" %llu", tk[2 * (4 * g) + 1] * (VAR == 2 ? 2 : 1));
    printf("\n");
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 40000;
    float *out; hipMalloc(&out, 16 * 1024 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 8 * 64);
    unsigned short h[8192]; unsigned x = 12345u;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x2800); }
    f16x8 *src; hipMalloc(&src, 16384); hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
#define ROW(V, NAME) run<1, V>(NAME, out, ticks, n, src); run<2, V>(NAME, out, ticks, n, src); run<3, V>(NAME, out, ticks, n, src); run<4, V>(NAME, out, ticks, n, src);
    ROW(0, "chain, then epilogue")
    ROW(1, "chain on two accumulators in turn")
    ROW(2, "two tiles per trip (two chains in turn, two epilogues)")
    ROW(3, "chain, 16 exps, then the adds")
    ROW(4, "chain, s_sleep 1, epilogue")
    return 0;
}
