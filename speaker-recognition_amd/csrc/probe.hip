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

// Mode 1 (round 6): the same chains FED the way the scoring kernel feeds them.  The probe above keeps both operands in registers
// for the whole launch and changes only C -- a load no real kernel presents: gmm_score_h2p_kernel reads a fresh A fragment from
// LDS for every MFMA (ds_read_b128 per lane, 8 KiB images of mixture parameters) and cycles through 8 resident B fragments (the
// wave's frames).  Here: 64 KiB of LDS per workgroup filled with random fp16 bit patterns (finite values: exponent field 8..22,
// all mantissa and sign bits random), every link's A fragment re-read from it (8 images of 8 fragments, round and round; the
// read of the fragment after next is in flight under the current MFMA, as in the kernel), 8 random B fragments in registers,
// chains of 8 links that start from a fresh C.  No global traffic, no exponentials, no barriers: the matrix pipe + its LDS feed
// + operands whose bits toggle -- the ceiling a kernel of that shape has under the power cap.
__global__ __launch_bounds__(768, 3)
void mfma_probe_streamed_kernel(float *out, unsigned long long *cycles, int links, unsigned seed) {
    __shared__ uint4 lds[4096];                                  // 64 KiB: 8 images x 8 fragments x 64 lanes x 16 B
    auto rnd16 = [](unsigned &st) {                              // xorshift32 -> one finite fp16 bit pattern
        st ^= st << 13; st ^= st >> 17; st ^= st << 5;
        const unsigned e = 8u + (st >> 7) % 15u;
        return (st & 0x83ffu) | (e << 10);
    };
    unsigned st = seed ^ (blockIdx.x * 7919u + threadIdx.x * 104729u + 1u);
    for (int i = threadIdx.x; i < 4096; i += 768) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = rnd16(st) | (rnd16(st) << 16);
        lds[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    f16x8 b[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = rnd16(st) | (rnd16(st) << 16);
        b[u] = __builtin_bit_cast(f16x8, make_uint4(w[0], w[1], w[2], w[3]));
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 c = zero, keep = zero;
    uint4 fr[8];
#pragma unroll
    for (int u = 0; u < 8; u++) fr[u] = lds[u * 64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int img = 1;
    for (int i = 0; i < links; i += 8) {
        const uint4 *next = lds + (img & 7) * 512 + lane;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[u]), b[u], u == 0 ? zero : c, 0, 0, 0);
            fr[u] = next[u * 64];                                // the register the MFMA just read: next image's fragment u
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {                            // issue order: MFMA, its register's refill, MFMA, ... (as the kernel's slots)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        keep[img & 15] += c[img & 15];                           // (the chain's result is used: one add per 8 MFMAs)
        img++;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) atomicMax(cycles, t1 - t0);
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) s += keep[r];
    out[blockIdx.x * 768 + threadIdx.x] = s;
}

// Runs the probe for about `ms_target` milliseconds; returns executed TFLOP/s (all CUs) and the shader clock it ran at.
// mode 0: operands resident in registers; 1: A fragments streamed from LDS, random operand bits (see above).
void mfma_peak_probe(double ms_target, double *tflops, double *mhz, int mode) {
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
        if (mode == 1)
            hipLaunchKernelGGL(mfma_probe_streamed_kernel, dim3((unsigned)n_cu), dim3(768), 0, ctx().stream, out.p, cyc.p, links, 0x9e3779b9u);
        else
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
