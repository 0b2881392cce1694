// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs (gradual underflow), or flush them?
// The split-fp16 scoring engine (csrc/gmm_score_split.hip) relies on the low parts of small values
// keeping their bits.  A = 2^-20 (fp16 subnormal) in every slot, B = 2^10: each output must be
// 16 * 2^-10 = 2^-6 if subnormals are honoured, 0 if flushed.  Also checks v_cvt_f16_f32 produces
// the subnormal in the first place.
// build: hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip -o mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float *out, unsigned *bits) {
    const _Float16 a = (_Float16)a_val, b = (_Float16)b_val;
    f16x8 av, bv;
    for (int i = 0; i < 8; i++) { av[i] = a; bv[i] = b; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; bits[0] = __builtin_bit_cast(unsigned short, a); }
}
int main() {
    float *out; unsigned *bits;
    hipMalloc(&out, 4); hipMalloc(&bits, 4);
    const float cases[][2] = {{9.5367431640625e-07f, 1024.f}, {5.9604644775390625e-08f, 16384.f}, {3.0517578125e-05f, 1.f}};
    int ok = 1;
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], out, bits);
        float h; unsigned hb;
        hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, bits, 4, hipMemcpyDeviceToHost);
        const float want = 16.f * c[0] * c[1];
        printf("a=%g (fp16 bits 0x%04x) b=%g -> %g (want %g) %s\n", c[0], hb, c[1], h, want, h == want ? "HONOURED" : "FLUSHED/WRONG");
        ok &= (h == want);
    }
    printf("fp16 subnormal MFMA inputs: %s\n", ok ? "honoured" : "NOT honoured");
    return ok ? 0 : 1;
}
