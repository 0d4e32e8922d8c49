// tools/inst_rate.hip -- VALU instruction throughput probe on gfx950 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096, CHAINS = 8;

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, unsigned seed) {
    float a[CHAINS];
    unsigned u[CHAINS];
    for (int i = 0; i < CHAINS; ++i) a[i] = threadIdx.x * 0.001f + i, u[i] = seed + threadIdx.x * 7 + i;
    const unsigned xb = 0x3f803f80u + seed;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, u[i]), __builtin_bit_cast(bf16x2_t, xb), a[i], false);
            if (OP == 1) a[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, u[i]), __builtin_bit_cast(f16x2_t, xb), a[i], false);
            if (OP == 2) a[i] = fmaf(a[i], 1.0001f, 0.5f);
            if (OP == 3) u[i] = (u[i] & 0x000F000Fu) | (0x43004300u + it);
            if (OP == 4) u[i] = (u[i] >> 4) ^ it;
            if (OP == 5) { f16x2_t v = __builtin_bit_cast(f16x2_t, u[i]); v = v * (f16x2_t){(_Float16)1.001f, (_Float16)0.999f} + (f16x2_t){(_Float16)0.5f, (_Float16)0.25f}; u[i] = __builtin_bit_cast(unsigned, v); }
            if (OP == 6) a[i] = (float)(u[i] & 0xffu) + a[i];   // cvt_f32_ubyte0 + add
        }
    }
    float s = 0; unsigned t = 0;
    for (int i = 0; i < CHAINS; ++i) s += a[i], t ^= u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + t;
}
template <int OP> int run(const char *name, float *out, int ops_per_iter) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 4;  // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD = 4 waves x ITERS x CHAINS x ops ; cycles at 2.4 GHz
    double winst = 4.0 * ITERS * CHAINS * ops_per_iter, cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s %.3f ms  %.2f cycles per wave-instruction per SIMD (4 waves/SIMD, assumes 2.4 GHz)\n", name, ms, cyc / winst);
    return 0;
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4 * 4));
    run<0>("v_dot2c_f32_bf16", out, 1); run<1>("v_dot2c_f32_f16", out, 1); run<2>("v_fma_f32", out, 1); run<3>("v_and_or_b32", out, 1);
    run<4>("v_lshrrev+v_xor", out, 2); run<5>("v_pk_fma_f16", out, 1); run<6>("v_cvt_f32_ubyte0+v_add", out, 2);
    return 0;
}
