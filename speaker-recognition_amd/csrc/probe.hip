// probe.hip -- what the matrix pipe sustains on THIS device under its power cap: a diagnostic for the roofline record.
//
// The guide's dense fp16 MFMA peak (2.5 PFLOP/s) is 256 CUs x 4 SIMDs x one v_mfma_f32_32x32x16_f16 per 32 cycles x 2.4 GHz.
// A kernel that issues nothing but that instruction on every SIMD runs the pipe at ~98 % -- and the socket (1.3-1.4 kW) answers
// by lowering the clock to ~1.55 GHz (profiles/r03_ubench_pinned.txt, r03_h2p_parts.txt): ~1.6 PFLOP/s is what the hardware
// delivers to ANY matrix kernel for longer than a few milliseconds.  bench.py reports the scoring kernel against both.
#include "common.hpp"

#include <algorithm>

namespace sr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 12 waves per CU (3 per SIMD, the scoring kernel's occupancy), each a dependent chain: the pipe goes to one wave at a time
// and stays busy (32.0 cycles per link, scripts/ubench/mfma_cadence.hip).
__global__ __launch_bounds__(768, 3)
void mfma_probe_kernel(float *out, unsigned long long *cycles, int links) {
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; j++) {       // small finite values with busy mantissas (an all-zero operand would flatter the power draw)
        a[j] = (_Float16)(0.001f * (float)((threadIdx.x * 7 + j * 13) % 61 + 1));
        b[j] = (_Float16)(0.002f * (float)((threadIdx.x * 5 + j * 11) % 53 + 1));
    }
    f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < links; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // (the pipe goes to a SIMD's oldest wave first: only its youngest wave is there from the first MFMA to the last)
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(cycles, t1 - t0);
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) s += c[r];
    out[blockIdx.x * 768 + threadIdx.x] = s;
}

// Runs the probe for about `ms_target` milliseconds; returns executed TFLOP/s (all CUs) and the shader clock it ran at.
void mfma_peak_probe(double ms_target, double *tflops, double *mhz) {
    ensure_device();
    const int n_cu = ctx().n_cu;
    DevBuf<float> out;
    out.alloc((size_t)n_cu * 768);
    DevBuf<unsigned long long> cyc;
    cyc.alloc(1);
    hipEvent_t e0, e1;
    SR_HIP(hipEventCreate(&e0));
    SR_HIP(hipEventCreate(&e1));
    auto run = [&](int links) {
        SR_HIP(hipEventRecord(e0, ctx().stream));
        hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)n_cu), dim3(768), 0, ctx().stream, out.p, cyc.p, links);
        SR_HIP(hipEventRecord(e1, ctx().stream));
        SR_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        SR_HIP(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms;
    };
    run(8 * 1024);                                             // warm-up (code object load, clocks up)
    SR_HIP(hipMemsetAsync(cyc.p, 0, sizeof(unsigned long long), ctx().stream));
    int links = 8 * 4096;
    double ms = run(links);
    // scale to the target: the clock settles under the cap within the first milliseconds, so one long launch is the measurement
    links = (int)std::min(2.0e8, std::max(8.0 * 1024, links * ms_target / std::max(ms, 1e-3))) / 8 * 8;
    SR_HIP(hipMemsetAsync(cyc.p, 0, sizeof(unsigned long long), ctx().stream));
    ms = run(links);
    unsigned long long c = 0;
    SR_HIP(hipMemcpy(&c, cyc.p, sizeof(c), hipMemcpyDeviceToHost));
    const double flop = (double)n_cu * 12.0 * (double)links * 2.0 * 32 * 32 * 16;
    if (tflops) *tflops = flop / (ms * 1e-3) / 1e12;
    if (mhz) *mhz = (double)c / (ms * 1e-3) / 1e6;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

}  // namespace sr
