// Microbenchmark: what do LDS instructions cost a wave that is busy issuing a dependent MFMA chain?  One wave per SIMD.
// Per tile: 8 x v_mfma_f32_32x32x16_f16 + NR x ds_read_b128 (fragments for the next tile) + NW x ds_write_b128
// (the previous tile's accumulators), LDS ops after the chain (AFTER) or spread between its links (BETWEEN).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NR, int NW, int BETWEEN>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *ticks, int n_tiles, const f16x8 *src) {
    __shared__ uint4 afrag[8 * 64];
    __shared__ float4 slot[4][4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 64; i += 256) afrag[i] = make_uint4(0x3c003c00u + i, 0x38003900u, 0x3a003b00u, 0x36003700u);
    __syncthreads();
    f16x8 a[8], an[8], b[8];
    for (int j = 0; j < 8; j++) { a[j] = src[(threadIdx.x + 64 * j) & 1023]; an[j] = a[j]; b[j] = src[(threadIdx.x * 3 + 64 * j + 7) & 1023]; }
    const f32x16 zero = {0};
    f32x16 prev = zero;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < n_tiles; t++) {
        f32x16 acc;
        float4 *dst = &slot[wave][0][lane];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], u == 0 ? zero : acc, 0, 0, 0);
            if (BETWEEN) {
                if (u < NR) an[u] = __builtin_bit_cast(f16x8, afrag[u * 64 + lane]);
                if (u < NW) dst[u * 64] = make_float4(prev[4 * u], prev[4 * u + 1], prev[4 * u + 2], prev[4 * u + 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!BETWEEN) {
#pragma unroll
            for (int u = 0; u < NR; u++) an[u] = __builtin_bit_cast(f16x8, afrag[u * 64 + lane]);
#pragma unroll
            for (int u = 0; u < NW; u++) dst[u * 64] = make_float4(prev[4 * u], prev[4 * u + 1], prev[4 * u + 2], prev[4 * u + 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = an[u];
        prev = acc;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = prev[0] + prev[7];
}

template <int NR, int NW, int BETWEEN> void run(float *out, unsigned long long *ticks, int n, const f16x8 *src) {
    hipLaunchKernelGGL((k<NR, NW, BETWEEN>), dim3(16), dim3(256), 0, 0, out, ticks, 64, src);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NR, NW, BETWEEN>), dim3(16), dim3(256), 0, 0, out, ticks, n, src);
    hipDeviceSynchronize();
    unsigned long long tk[4]; hipMemcpy(tk, ticks, 32, hipMemcpyDeviceToHost);
    printf("8 MFMA + %d ds_read_b128 + %d ds_write_b128, LDS ops %s: %.0f cycles per tile\n", NR, NW, BETWEEN ? "between the links" : "after the chain  ", (double)tk[0] / n);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; hipMalloc(&out, 16 * 256 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 64);
    unsigned short h[8192]; unsigned x = 12345u;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x2800); }
    f16x8 *src; hipMalloc(&src, 16384); hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    run<0, 0, 0>(out, ticks, n, src);
    run<8, 0, 0>(out, ticks, n, src);
    run<8, 0, 1>(out, ticks, n, src);
    run<0, 4, 0>(out, ticks, n, src);
    run<0, 4, 1>(out, ticks, n, src);
    run<8, 4, 0>(out, ticks, n, src);
    run<8, 4, 1>(out, ticks, n, src);
    return 0;
}
