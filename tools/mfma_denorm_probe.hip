// tools/mfma_denorm_probe.hip -- does v_mfma_f32_32x32x16_bf16 honour bf16 DENORMAL operands on gfx950?  (developer probe)
// A masked nibble pair `w & 0x000F000F` is two bf16 denormals q * 2^-133; the decode GEMV feeds them to v_dot2c_f32_bf16, which honours them
// exactly.  If the matrix cores did too, an int4 GEMM could multiply raw codes (no in-loop conversion) and apply scale / bias per group on the
// accumulator side.  One wave: A = codes 0..15 as denormals, B = 2^64, C must be sum_k q_k * 2^-69.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_denorm_probe.hip -o tools/mfma_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

__global__ void k(float *out, float *out_dot) {
    const int lane = threadIdx.x;
    union { unsigned short u[8]; bf16x8_t v; } a, b;
    // A operand of lane l: row l & 31, k = 8 * (l >> 5) .. + 8: codes (k & 15) as bf16 denormals; B: 2^64 everywhere
    for (int i = 0; i < 8; ++i) a.u[i] = (unsigned short)((8 * (lane >> 5) + i) & 15), b.u[i] = 0x5F80;
    f32x16_t c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
    // the same sum through v_dot2_f32_bf16 pairs (the GEMV's instruction) for reference
    typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2_t;
    float d = 0.0f;
    for (int kk = 0; kk < 16; kk += 2) {
        union { unsigned short u[2]; bf16x2_t v; } p, q;
        p.u[0] = (unsigned short)(kk & 15), p.u[1] = (unsigned short)((kk + 1) & 15), q.u[0] = q.u[1] = 0x5F80;
        d = __builtin_amdgcn_fdot2_f32_bf16(p.v, q.v, d, false);
    }
    if (lane == 0) out_dot[0] = d;
}
int main() {
    float *o, *od, h = -1.0f, hd = -1.0f;
    hipMalloc(&o, 4), hipMalloc(&od, 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, od);
    hipMemcpy(&h, o, 4, hipMemcpyDeviceToHost), hipMemcpy(&hd, od, 4, hipMemcpyDeviceToHost);
    const float want = 120.0f * ldexpf(1.0f, -69);  // sum of 0..15 = 120
    printf("mfma  C[0][0] = %g   (sum_k q_k * 2^-69 = %g)   -> denormal operands %s\n", h, want, h == want ? "HONOURED" : (h == 0.0f ? "FLUSHED to zero" : "something else"));
    printf("dot2  sum      = %g\n", hd);
    return 0;
}
