// Microbenchmark: wave specialisation on a SIMD -- the OLDER wave only issues MFMAs (8-link chains, accumulators of
// the previous tile handed over through LDS), the YOUNGER wave only runs the epilogues (16 v_exp_f32 + adds) -- so that
// the arbiter's oldest-first rule works for the matrix pipe instead of against it.  One workgroup of 8 waves per CU
// (waves w and w + 4 share a SIMD), a ring of RING accumulator slots per pair, sequence numbers in LDS, polling with a
// bail-out.  Cycles per tile per SIMD; FRAGS = 1: the producer also reads 8 x ds_read_b128 of "A fragments" per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int RING = 4;

template <int FRAGS>
__global__ __launch_bounds__(512, 2) void k(float *out, unsigned long long *ticks, int n_tiles, const f16x8 *src) {
    __shared__ float4 slot[4][RING][4][64];        // [pair][ring][quarter][lane]: 4 KiB per slot
    __shared__ uint4 afrag[8 * 64];
    __shared__ volatile int produced[4], consumed[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pair = wave & 3, role = wave >> 2;
    if (threadIdx.x < 4) { produced[threadIdx.x] = 0; consumed[threadIdx.x] = 0; }
    for (int i = threadIdx.x; i < 8 * 64; i += 512) afrag[i] = make_uint4(0x3c003c00u + i, 0x38003900u, 0x3a003b00u, 0x36003700u);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    int bailed = 0;
    if (role == 0) {            // ---- producer: MFMA only ----
        f16x8 a[8], b[8];
        for (int j = 0; j < 8; j++) { a[j] = src[(threadIdx.x + 64 * j) & 1023]; b[j] = src[(threadIdx.x * 3 + 64 * j + 7) & 1023]; }
        const f32x16 zero = {0};
        f32x16 prev = zero;
        for (int t = 0; t <= n_tiles; t++) {
            f32x16 acc = zero;
            if (t < n_tiles) {
                if (FRAGS) {
#pragma unroll
                    for (int u = 0; u < 8; u++) a[u] = __builtin_bit_cast(f16x8, afrag[u * 64 + lane]);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], u == 0 ? zero : acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) {        // hand over the PREVIOUS tile (its chain finished long ago)
                const int seq = t - 1;
                int spins = 0;
                while (seq - consumed[pair] >= RING) { __builtin_amdgcn_s_sleep(1); if (++spins > 2000000) { bailed = 1; break; } }
                float4 *dst = &slot[pair][seq % RING][0][lane];
                dst[0 * 64] = make_float4(prev[0], prev[1], prev[2], prev[3]);
                dst[1 * 64] = make_float4(prev[4], prev[5], prev[6], prev[7]);
                dst[2 * 64] = make_float4(prev[8], prev[9], prev[10], prev[11]);
                dst[3 * 64] = make_float4(prev[12], prev[13], prev[14], prev[15]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) produced[pair] = seq + 1;
            }
            prev = acc;
            __builtin_amdgcn_sched_barrier(0);
            if (bailed) break;
        }
        res = prev[0];
    } else {                    // ---- consumer: epilogues only ----
        float ssum = 0.f;
        for (int seq = 0; seq < n_tiles; seq++) {
            int spins = 0;
            while (produced[pair] <= seq) { __builtin_amdgcn_s_sleep(1); if (++spins > 2000000) { bailed = 1; break; } }
            if (bailed) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float4 *srcp = &slot[pair][seq % RING][0][lane];
            const float4 q0 = srcp[0], q1 = srcp[64], q2 = srcp[128], q3 = srcp[192];
            float e0 = __builtin_amdgcn_exp2f(q0.x) + __builtin_amdgcn_exp2f(q0.z), e1 = __builtin_amdgcn_exp2f(q0.y) + __builtin_amdgcn_exp2f(q0.w);
            e0 += __builtin_amdgcn_exp2f(q1.x) + __builtin_amdgcn_exp2f(q1.z); e1 += __builtin_amdgcn_exp2f(q1.y) + __builtin_amdgcn_exp2f(q1.w);
            e0 += __builtin_amdgcn_exp2f(q2.x) + __builtin_amdgcn_exp2f(q2.z); e1 += __builtin_amdgcn_exp2f(q2.y) + __builtin_amdgcn_exp2f(q2.w);
            e0 += __builtin_amdgcn_exp2f(q3.x) + __builtin_amdgcn_exp2f(q3.z); e1 += __builtin_amdgcn_exp2f(q3.y) + __builtin_amdgcn_exp2f(q3.w);
            ssum += e0 + e1;
            asm volatile("" : "+v"(ssum));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) consumed[pair] = seq + 1;
        }
        res = ssum;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) { ticks[2 * wave] = t1 - t0; ticks[2 * wave + 1] = bailed; }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int FRAGS> void run(const char *name, float *out, unsigned long long *ticks, int n, const f16x8 *src, int grid) {
    hipLaunchKernelGGL((k<FRAGS>), dim3(grid), dim3(512), 0, 0, out, ticks, 64, src);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FRAGS>), dim3(grid), dim3(512), 0, 0, out, ticks, n, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long tk[16]; hipMemcpy(tk, ticks, 8 * 16, hipMemcpyDeviceToHost);
    printf("%-40s grid %3d: %.3f ms; SIMD 0: producer %.0f cycles per tile, consumer %.0f; bailed %llu %llu; MFMA pipe %.0f %%\n", name, grid, ms,
           (double)tk[0] / n, (double)tk[8] / n, tk[1], tk[9], 100.0 * 256.0 * n / (double)tk[8]);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; hipMalloc(&out, 256 * 512 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 8 * 16);
    unsigned short h[8192]; unsigned x = 12345u;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x2800); }
    f16x8 *src; hipMalloc(&src, 16384); hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    for (int grid : {16, 256}) {
        run<0>("producer: 8 MFMA + 4 ds_write_b128", out, ticks, n, src, grid);
        run<1>("producer: + 8 ds_read_b128 of fragments", out, ticks, n, src, grid);
    }
    return 0;
}
