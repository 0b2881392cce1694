// Microbenchmark (round 3): does ONE wave overlap its own matrix work with its own vector work when the order of the
// instructions is pinned?  mfma_lse_inwave.hip left the interleave to sched_group_barrier hints, which the compiler did not
// honour (the ISA has 5 MFMAs back to back and 7 exps in a clump): its "MFMA time and vector time ADD" reading was a
// statement about that schedule, not about the hardware.  Here every slot is fenced with sched_barrier(0):
//     slot u (u = 0..7):  v_mfma(cur chain)  |  NV vector instructions of the PREVIOUS tile's epilogue
// 8 MFMAs (256 cycles) against 16 v_exp_f32 + 16 v_add_f32 (or 8 v_pk_add_f32) per tile -- the configs[2] kernel's ratio.
//   VAR 0  chain, then the whole epilogue (what a straightforward loop does)
//   VAR 1  epilogue of the previous tile spread over slots 1..7 (slot 0 empty: its operands were written by the MFMA just before)
//   VAR 2  same with packed adds
//   VAR 3  chain only        VAR 4  epilogue only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define FENCE() __builtin_amdgcn_sched_barrier(0)

// One slot = one asm block: the MFMA, then the vector instructions, in exactly this order (the IR-level vectoriser and the
// post-RA scheduler both moved builtin-based code across sched_barrier(0)).  The adds of a slot consume the exps of the slot
// before (no trans-result-use hazard inside a block); e0..e2 carry them from block to block.
template <int VAR> __device__ __forceinline__ void tile(const f16x8 (&a)[8], const f16x8 (&b)[8], f32x16 &cur, const f32x16 &prev, float &s0, float &s1,
                                                         f32x2 &ps) {
    const f32x16 zero = {0};
    if (VAR == 1 || VAR == 2) {
        // exps land in v160..v162 (named registers: a packed add needs an aligned pair, and an asm operand cannot name the halves
        // of a tuple); the kernel is small enough that the compiler never goes near them
#define CLOB "v160", "v161", "v162"
#define MF "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n"
#define ADD2 (VAR == 1 ? "v_add_f32 %1, %1, v160\n v_add_f32 %1, %1, v161\n" : "v_pk_add_f32 %2, %2, v[160:161]\n")
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(cur) : "v"(a[0]), "v"(b[0]));   // slot 0: the previous chain's last MFMA is still writing prev
        asm volatile(MF "v_exp_f32 v160, %5\n v_exp_f32 v161, %6" : "+v"(cur), "+v"(s0), "+v"(ps) : "v"(a[1]), "v"(b[1]), "v"(prev[0]), "v"(prev[1]) : CLOB);
#define SLOT2(U, I, J) \
        if (VAR == 1) asm volatile(MF "v_add_f32 %1, %1, v160\n v_add_f32 %1, %1, v161\n v_exp_f32 v160, %5\n v_exp_f32 v161, %6" : "+v"(cur), "+v"(s0), "+v"(ps) : "v"(a[U]), "v"(b[U]), "v"(prev[I]), "v"(prev[J]) : CLOB); \
        else asm volatile(MF "v_pk_add_f32 %2, %2, v[160:161]\n v_exp_f32 v160, %5\n v_exp_f32 v161, %6" : "+v"(cur), "+v"(s0), "+v"(ps) : "v"(a[U]), "v"(b[U]), "v"(prev[I]), "v"(prev[J]) : CLOB)
#define SLOT3(U, I, J, K, PRE) \
        if (VAR == 1) asm volatile(MF "v_add_f32 %1, %1, v160\n v_add_f32 %1, %1, v161\n" PRE "v_exp_f32 v160, %5\n v_exp_f32 v161, %6\n v_exp_f32 v162, %7" : "+v"(cur), "+v"(s0), "+v"(ps) : "v"(a[U]), "v"(b[U]), "v"(prev[I]), "v"(prev[J]), "v"(prev[K]) : CLOB); \
        else asm volatile(MF "v_pk_add_f32 %2, %2, v[160:161]\n" PRE "v_exp_f32 v160, %5\n v_exp_f32 v161, %6\n v_exp_f32 v162, %7" : "+v"(cur), "+v"(s0), "+v"(ps) : "v"(a[U]), "v"(b[U]), "v"(prev[I]), "v"(prev[J]), "v"(prev[K]) : CLOB)
        SLOT2(2, 2, 3); SLOT2(3, 4, 5);
        SLOT3(4, 6, 7, 8, "");
        SLOT3(5, 9, 10, 11, "v_add_f32 %1, %1, v162\n");
        if (VAR == 1) asm volatile(MF "v_add_f32 %1, %1, v160\n v_add_f32 %1, %1, v161\n v_add_f32 %1, %1, v162\n v_exp_f32 v160, %5\n v_exp_f32 v161, %6" : "+v"(cur), "+v"(s0), "+v"(ps) : "v"(a[6]), "v"(b[6]), "v"(prev[12]), "v"(prev[13]) : CLOB);
        else asm volatile(MF "v_pk_add_f32 %2, %2, v[160:161]\n v_add_f32 %1, %1, v162\n v_exp_f32 v160, %5\n v_exp_f32 v161, %6" : "+v"(cur), "+v"(s0), "+v"(ps) : "v"(a[6]), "v"(b[6]), "v"(prev[12]), "v"(prev[13]) : CLOB);
        SLOT2(7, 14, 15);
        if (VAR == 1) asm volatile("v_add_f32 %0, %0, v160\n v_add_f32 %0, %0, v161" : "+v"(s0), "+v"(ps) : : CLOB);
        else asm volatile("v_pk_add_f32 %1, %1, v[160:161]" : "+v"(s0), "+v"(ps) : : CLOB);
        return;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (VAR != 4) cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], u == 0 ? zero : cur, 0, 0, 0);
        FENCE();
    }
    if (VAR == 0 || VAR == 4) {
        const f32x16 &src = VAR == 0 ? cur : prev;
#pragma unroll
        for (int r = 0; r < 16; r += 2) { s0 += __builtin_amdgcn_exp2f(src[r]); s1 += __builtin_amdgcn_exp2f(src[r + 1]); }
        FENCE();
    }
}

template <int WPS, int VAR>
__global__ __launch_bounds__(WPS * 256) __attribute__((amdgpu_waves_per_eu(3, 8))) void k(float *out, unsigned long long *ticks, int iters, const f16x8 *src) {
    f16x8 a[8], b0[8], b1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { a[j] = src[(threadIdx.x + 64 * j) & 1023]; b0[j] = src[(threadIdx.x * 3 + 64 * j + 7) & 1023]; b1[j] = src[(threadIdx.x * 5 + 64 * j + 3) & 1023]; }
    float s0 = 0.f, s1 = 0.f;
    f32x2 ps = {0.f, 0.f};
    f32x16 x = {0}, y = {0};
    if (VAR == 4) { for (int r = 0; r < 16; r++) { x[r] = -(float)(threadIdx.x & 7) - r; y[r] = -1.5f * r; } }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i += 2) {
        tile<VAR>(a, b0, x, y, s0, s1, ps);     // chain into x, epilogue of y
        tile<VAR>(a, b1, y, x, s0, s1, ps);     // chain into y, epilogue of x
        if (VAR == 4) asm volatile("" : "+v"(x), "+v"(y));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[threadIdx.x >> 6] = t1 - t0;
    out[blockIdx.x * WPS * 256 + threadIdx.x] = s0 + s1 + ps[0] + ps[1] + x[3] + y[5];
}

template <int WPS, int VAR> void run(const char *name, float *out, unsigned long long *ticks, int iters, const f16x8 *src, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-52s waves/SIMD %d grid %3d ", name, WPS, grid); fflush(stdout);
    hipLaunchKernelGGL((k<WPS, VAR>), dim3(grid), dim3(WPS * 256), 0, 0, out, ticks, 16, src);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WPS, VAR>), dim3(grid), dim3(WPS * 256), 0, 0, out, ticks, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long tk[16]; hipMemcpy(tk, ticks, 8 * 4 * WPS, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < 4 * WPS; w++) if (tk[w] > mx) mx = tk[w];
    printf(" %.3f ms  %6.0f cycles per tile per SIMD (slowest wave)  MFMA pipe %3.0f %%  clock %.2f GHz\n", ms,
           (double)mx / iters / WPS, VAR == 4 ? 0.0 : 100.0 * 256.0 * WPS * iters / (double)mx, (double)mx / (ms * 1e6));
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    unsigned long long *ticks; hipMalloc(&ticks, 8 * 64);
    unsigned short h[8192]; unsigned x = 12345u;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x2800); }
    f16x8 *src; hipMalloc(&src, 16384); hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    for (int grid : {16, 256}) {
        run<1, 3>("chain only", out, ticks, iters, src, grid);
        run<1, 4>("epilogue only", out, ticks, iters, src, grid);
        run<1, 0>("chain, then epilogue", out, ticks, iters, src, grid);
        run<2, 0>("chain, then epilogue", out, ticks, iters, src, grid);
        run<3, 0>("chain, then epilogue", out, ticks, iters, src, grid);
        run<1, 1>("previous epilogue pinned into slots 1..7", out, ticks, iters, src, grid);
        run<2, 1>("previous epilogue pinned into slots 1..7", out, ticks, iters, src, grid);
        run<3, 1>("previous epilogue pinned into slots 1..7", out, ticks, iters, src, grid);
        run<1, 2>("same, packed adds", out, ticks, iters, src, grid);
        run<2, 2>("same, packed adds", out, ticks, iters, src, grid);
        run<3, 2>("same, packed adds", out, ticks, iters, src, grid);
    }
    return 0;
}
