// Microbenchmark (round 4): what a packed fp32 instruction costs against two plain ones on gfx950 -- the question behind "packed
// butterflies for the MFCC kernel" (DESIGN.md 5b).  Every wave runs ITER x 32 independent chains:
//   VAR 0  64 v_fma_f32      (32 chains x 2)          VAR 1  32 v_pk_fma_f32  (the same 64 FMAs per lane)
//   VAR 2  64 v_add_f32                                VAR 3  32 v_pk_add_f32
//   VAR 4  32 v_pk_fma_f32 with op_sel / neg modifiers (a complex multiply's second half: swapped, negated operand)
// with 1..4 waves per SIMD on every CU.  Prints cycles per wave-instruction per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_pk_rate valu_pk_rate.hip ; run: ./valu_pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x2 a[16];
    const float t = (float)threadIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = f32x2{t + i, t - i};
    f32x2 m = {1.0001f + t * 1e-6f, 0.9999f}, c = {1e-3f, -1e-3f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (VAR == 0) asm volatile("v_fma_f32 %0, %0, %2, %4\n v_fma_f32 %1, %1, %3, %5" : "+v"(a[i].x), "+v"(a[i].y) : "v"(m.x), "v"(m.y), "v"(c.x), "v"(c.y));
                if (VAR == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (VAR == 2) asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a[i].x), "+v"(a[i].y) : "v"(c.x), "v"(c.y));
                if (VAR == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (VAR == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(m), "v"(c));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int VAR> static void run(const char *name, int waves_per_simd) {
    float *d;
    hipMalloc(&d, 4);
    const int iters = 20000, n_cu = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // one 256-thread workgroup = one wave per SIMD of a CU; waves_per_simd of them per CU
    hipLaunchKernelGGL(k<VAR>, dim3(n_cu * waves_per_simd), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(n_cu * waves_per_simd), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)iters * ((VAR == 0 || VAR == 2) ? 64 : 32) * waves_per_simd;   // per SIMD
    printf("%-46s waves/SIMD %d  %.3f ms  %.2f ns per wave-instruction per SIMD  (%.1f TFLOP/s)\n", name, waves_per_simd, ms, ms * 1e6 / wave_instr,
           (double)iters * 64 * 64 * ((VAR == 2 || VAR == 3) ? 1 : 2) * waves_per_simd * 1024 / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main() {
    for (int w = 1; w <= 4; w++) {
        run<0>("2 x v_fma_f32 per pair", w);
        run<1>("v_pk_fma_f32", w);
        run<4>("v_pk_fma_f32 op_sel / neg_lo", w);
        run<2>("2 x v_add_f32 per pair", w);
        run<3>("v_pk_add_f32", w);
    }
    return 0;
}
