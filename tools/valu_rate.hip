// valu_rate.hip -- developer probe: issue cost (shader cycles per wave-instruction) of the VALU candidates for the GEMV inner product
// on gfx950, one and two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP16(S) S S S S S S S S S S S S S S S S
template <int WHICH>
__global__ void __launch_bounds__(1024) k_rate(unsigned long long *out, float *sink, int iters) {
    float a0 = threadIdx.x, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f;
    unsigned x0 = 0x3f803f80u + threadIdx.x, x1 = 0x40004000u, w0 = 0x00030005u, w1 = 0x00070001u;
    float f0 = 1.5f, f1 = 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, q0 = {f0, f1}, q1 = {f1, f0};
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) {  // v_dot2c_f32_bf16, 4 chains
            REP16(asm volatile("v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(x0), "v"(w1), "v"(x1));)
        } else if (WHICH == 1) {  // v_dot2_f32_bf16 (VOP3P)
            REP16(asm volatile("v_dot2_f32_bf16 %0, %4, %5, %0\n\tv_dot2_f32_bf16 %1, %6, %7, %1\n\tv_dot2_f32_bf16 %2, %4, %7, %2\n\tv_dot2_f32_bf16 %3, %6, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(x0), "v"(w1), "v"(x1));)
        } else if (WHICH == 2) {  // v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %5, %4, %1\n\tv_fma_f32 %2, %4, %4, %2\n\tv_fma_f32 %3, %5, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1));)
        } else if (WHICH == 3) {  // v_pk_fma_f32
            REP16(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %3, %2, %1\n\tv_pk_fma_f32 %0, %3, %3, %0\n\tv_pk_fma_f32 %1, %2, %2, %1"
                               : "+v"(p0), "+v"(p1) : "v"(q0), "v"(q1));)
        } else if (WHICH == 4) {  // v_dot2_f32_f16
            REP16(asm volatile("v_dot2_f32_f16 %0, %4, %5, %0\n\tv_dot2_f32_f16 %1, %6, %7, %1\n\tv_dot2_f32_f16 %2, %4, %7, %2\n\tv_dot2_f32_f16 %3, %6, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(x0), "v"(w1), "v"(x1));)
        } else if (WHICH == 5) {  // v_fma_mix_f32 with f16 halves (lo of src0 x lo of src1 + f32)
            REP16(asm volatile("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,1,0]\n\tv_fma_mix_f32 %1, %6, %7, %1 op_sel_hi:[1,1,0]\n\tv_fma_mix_f32 %2, %4, %7, %2 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n\t"
                               "v_fma_mix_f32 %3, %6, %5, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(x0), "v"(w1), "v"(x1));)
        } else if (WHICH == 6) {  // v_dot2c_f32_f16
            REP16(asm volatile("v_dot2c_f32_f16 %0, %4, %5\n\tv_dot2c_f32_f16 %1, %6, %7\n\tv_dot2c_f32_f16 %2, %4, %7\n\tv_dot2c_f32_f16 %3, %6, %5"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(x0), "v"(w1), "v"(x1));)
        } else if (WHICH == 7) {  // v_and_b32 + v_lshrrev (integer ALU reference)
            REP16(asm volatile("v_and_b32 %0, %4, %0\n\tv_lshrrev_b32 %1, 4, %1\n\tv_and_b32 %2, %5, %2\n\tv_lshrrev_b32 %3, 1, %3"
                               : "+v"(x0), "+v"(x1), "+v"(w0), "+v"(w1) : "v"(w0), "v"(w1));)
        } else if (WHICH == 9) {  // the same v_dot2c chains as 0, but 4096 instructions of straight-line code per iteration (16 KB: instruction fetch, not the loop buffer)
#define BIG4 asm volatile("v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(x0), "v"(w1), "v"(x1));
            REP16(REP16(BIG4 BIG4 BIG4 BIG4))
        } else if (WHICH == 11) {  // the rows kernel's per-row pattern: 4 x v_mov 0, 32 dot2c on 4 chains, then the short dependent tail
            asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3));
            asm volatile("v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5\n\t"
                         "v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5\n\t"
                         "v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5\n\t"
                         "v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5\n\t"
                         "v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5\n\t"
                         "v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5\n\t"
                         "v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5\n\t"
                         "v_dot2c_f32_bf16 %0, %4, %5\n\tv_dot2c_f32_bf16 %1, %6, %7\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %6, %5\n\t"
                         "s_nop 2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\ts_nop 0\n\tv_mul_f32 %1, 0x3d800000, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, 0x62000000, %0\n\t"
                         "v_sub_f32 %0, %0, %8\n\tv_mul_f32 %1, %9, %8\n\tv_fmac_f32 %1, %9, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(x0), "v"(w1), "v"(x1), "v"(f0), "v"(f1));
            f0 += a1;
        } else if (WHICH == 8) {  // v_dot4_i32_i8 / iu8
            REP16(asm volatile("v_dot4_i32_i8 %0, %4, %5, %0\n\tv_dot4_i32_i8 %1, %6, %7, %1\n\tv_dot4_i32_i8 %2, %4, %7, %2\n\tv_dot4_i32_i8 %3, %6, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0), "v"(x0), "v"(w1), "v"(x1));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 7) out[256 * 16] = r1 - r0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + __builtin_bit_cast(float, x0 ^ x1 ^ w0 ^ w1);
}

template <int WHICH>
static int run(const char *name, unsigned long long *d_out, float *d_sink, int per_iter = 64) {
    const int iters = 20000 * 64 / per_iter;
    for (int waves : {4, 8, 12, 16}) {  // per workgroup = per CU: 1 or 2 per SIMD
        k_rate<WHICH><<<256, waves * 64>>>(d_out, d_sink, iters);
        k_rate<WHICH><<<256, waves * 64>>>(d_out, d_sink, iters);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(256 * 16 + 1);
        CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += (double)h[b * 16 + w];
        s /= 256.0 * waves;
        printf("%-18s %d waves/SIMD: %6.2f cycles per wave-instruction (per wave), %6.2f per SIMD; counter %.2f GHz (cycle counter / 100 MHz real-time counter), %.2f ns per instruction and SIMD\n", name,
               waves / 4, s / (iters * (double)per_iter), s / (iters * (double)per_iter) / (waves / 4), (double)h[7 * 16] / ((double)h[256 * 16] * 10.0), (double)h[256 * 16] * 10.0 / (iters * (double)per_iter) / (waves / 4));
    }
    return 0;
}
int main() {
    unsigned long long *d_out;
    float *d_sink;
    CK(hipMalloc(&d_out, 256 * 16 * 8 + 8));
    CK(hipMalloc(&d_sink, 256 * 1024 * 4));
    run<0>("v_dot2c_f32_bf16", d_out, d_sink);
    run<9>("v_dot2c 16KB body", d_out, d_sink, 4096);
    run<11>("rows-kernel row", d_out, d_sink, 46);
    run<1>("v_dot2_f32_bf16", d_out, d_sink);
    run<6>("v_dot2c_f32_f16", d_out, d_sink);
    run<4>("v_dot2_f32_f16", d_out, d_sink);
    run<2>("v_fma_f32", d_out, d_sink);
    run<3>("v_pk_fma_f32", d_out, d_sink);
    run<5>("v_fma_mix_f32", d_out, d_sink);
    run<7>("v_and/v_lshrrev", d_out, d_sink);
    run<8>("v_dot4_i32_i8", d_out, d_sink);
    return 0;
}
