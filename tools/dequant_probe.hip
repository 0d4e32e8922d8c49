// tools/dequant_probe.hip -- which instruction sequence turns one W4M code word (8 int4 codes) into 8 weights in T, bit for bit
// mx.dequantize's T(fp32(s * q) + b), at the lowest VALU issue cost?  (developer probe, round 5)
//
// Variants (one code word -> one MFMA A fragment of 4 dwords):
//   0  w4m_dequant of rounds 2-4: 2 masks + shift, 8 v_cvt_f32_ubyteN, 8 fma (hipcc packs them into 4 v_pk_fma_f32), 4 v_cvt_pk_bf16_f32
//   1  the same with the fmas forced scalar (8 v_fma_f32)
//   2  v_fma_mix_f32: a masked nibble in a 16-bit half IS the f16 denormal q * 2^-24 (bits 0-3) or q * 2^-20 (bits 4-7); the instruction
//      converts an f16 source to fp32 on the fly, so fp32 weight = fma(f16(nibble), s * 2^24, b) in ONE instruction per code
//      (s * q is exact in fp32, hence fma == fadd(fmul)); 4 masks + 1 shift, 8 v_fma_mix_f32, 4 v_cvt_pk_bf16_f32 = 17 instructions
// Part A checks variant 2 (and 1) against variant-0 arithmetic on every (scale, bias) pair of a sample x all 16 codes.
// Part B times the three on all CUs, 1 and 2 waves per SIMD (ns per word and wave).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dequant_probe.hip -o tools/dequant_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned short u16;

__device__ __forceinline__ float bf16_f32(u16 b) { return __builtin_bit_cast(float, (u32)b << 16); }
__device__ __forceinline__ u32 pack_bf16(float lo, float hi) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(u32, __builtin_convertvector((f2_t){lo, hi}, b2_t));
}

// variant 0 / 1
template <bool SCALAR>
__device__ __forceinline__ uint4 dq_cvt(u32 word, float s, float b) {
    u32 e = word & 0x0F0F0F0Fu, o = (word >> 4) & 0x0F0F0F0Fu;
    asm volatile("" : "+v"(e), "+v"(o));
    float c[8] = {(float)(e & 0xFFu), (float)((e >> 16) & 0xFFu), (float)(o & 0xFFu), (float)((o >> 16) & 0xFFu),
                  (float)((e >> 8) & 0xFFu), (float)(e >> 24), (float)((o >> 8) & 0xFFu), (float)(o >> 24)};
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r[i] = __builtin_fmaf(s, c[i], b);
        if (SCALAR) asm volatile("" : "+v"(r[i]));  // keeps the SLP vectoriser from pairing them
    }
    return make_uint4(pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3]), pack_bf16(r[4], r[5]), pack_bf16(r[6], r[7]));
}

// variant 2: s24 = s * 2^24, s20 = s * 2^20
__device__ __forceinline__ float mix_lo(u32 h, float s, float b) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(s), "v"(b));
    return d;
}
__device__ __forceinline__ float mix_hi(u32 h, float s, float b) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(s), "v"(b));
    return d;
}
__device__ __forceinline__ uint4 dq_mix(u32 w, float s24, float s20, float b) {
    const u32 w8 = w >> 8;
    const u32 t0 = w & 0x000F000Fu, t1 = w & 0x00F000F0u, t2 = w8 & 0x000F000Fu, t3 = w8 & 0x00F000F0u;
    return make_uint4(pack_bf16(mix_lo(t0, s24, b), mix_hi(t0, s24, b)), pack_bf16(mix_lo(t1, s20, b), mix_hi(t1, s20, b)),
                      pack_bf16(mix_lo(t2, s24, b), mix_hi(t2, s24, b)), pack_bf16(mix_lo(t3, s20, b), mix_hi(t3, s20, b)));
}

// Part A: thread i takes (scale, bias) = sb[i]; word pattern p places code (p + j) & 15 at code position j
__global__ void k_check(const u32 *sb, int n, unsigned long long *bad, u32 *first_bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = bf16_f32((u16)(sb[i] & 0xFFFF)), b = bf16_f32((u16)(sb[i] >> 16));
    unsigned long long nb = 0;
    for (int p = 0; p < 16; ++p) {
        u32 w = 0;  // W4S word format: codes (2 i, 2 i + 1) at nibble i of the low / high half
        for (int j = 0; j < 8; ++j) w |= (u32)((p + j) & 15) << (4 * (j >> 1) + 16 * (j & 1));
        const uint4 r0 = dq_cvt<false>(w, s, b), r1 = dq_cvt<true>(w, s, b), r2 = dq_mix(w, s * 0x1p24f, s * 0x1p20f, b);
        // independent reference: separate multiply and add, RNE to bf16
        u32 ref[4];
        for (int k = 0; k < 4; ++k) {
            const float lo = __fadd_rn(__fmul_rn(s, (float)((p + 2 * k) & 15)), b), hi = __fadd_rn(__fmul_rn(s, (float)((p + 2 * k + 1) & 15)), b);
            ref[k] = pack_bf16(lo, hi);
        }
        const u32 a0[4] = {r0.x, r0.y, r0.z, r0.w}, a1[4] = {r1.x, r1.y, r1.z, r1.w}, a2[4] = {r2.x, r2.y, r2.z, r2.w};
        for (int k = 0; k < 4; ++k) {
            if (a0[k] != ref[k]) nb += 1;
            if (a1[k] != ref[k]) nb += 1ull << 20;
            if (a2[k] != ref[k]) {
                nb += 1ull << 40;
                if (atomicCAS(first_bad, 0u, sb[i] | 1u) == 0u) first_bad[1] = a2[k], first_bad[2] = ref[k], first_bad[3] = (u32)p * 16 + k;
            }
        }
    }
    if (nb) atomicAdd(bad, nb);
}

// Part B: every wave converts `iters` x 8 words held in registers, xor-folds the fragments (kept live), stores once
template <int V>
__global__ void __launch_bounds__(512) k_time(const u32 *words, const u32 *sb, int iters, u32 *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = words[(t * 8 + j) & 4095];
    const u32 sbv = sb[t & 1023];
    const float s = bf16_f32((u16)(sbv & 0xFFFF)), b = bf16_f32((u16)(sbv >> 16));
    const float s24 = s * 0x1p24f, s20 = s * 0x1p20f;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint4 r;
            if (V == 0) r = dq_cvt<false>(w[j], s, b);
            else if (V == 1) r = dq_cvt<true>(w[j], s, b);
            else r = dq_mix(w[j], s24, s20, b);
            acc.x ^= r.x, acc.y ^= r.y, acc.z ^= r.z, acc.w ^= r.w;
            w[j] += 0x11111111u * (u32)(it & 1) + acc.x * 0u;  // defeats loop-invariant hoisting without changing the mix much
            asm volatile("" : "+v"(w[j]));
        }
    }
    out[t] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main() {
    // ---- Part A
    const int n = 1 << 20;
    std::vector<u32> sb(n);
    srand(5);
    for (int i = 0; i < n; ++i) {
        // scale: random bf16 in a realistic + extreme range (both signs, tiny 1e-7 clamps, large), bias: random bf16 incl. zero and huge ratios
        u32 sbits, bbits;
        if (i < 65536) sbits = (u32)i, bbits = (u32)((i * 40503u) >> 8) & 0xFFFF;  // every bf16 scale pattern once (NaN / inf included: compared bitwise too)
        else {
            const int es = 100 + rand() % 40, eb = 90 + rand() % 50;  // exponents 2^-27 .. 2^12 / 2^-37 .. 2^12
            sbits = ((u32)(rand() & 1) << 15) | ((u32)es << 7) | (u32)(rand() & 127);
            bbits = (rand() % 16 == 0) ? 0u : (((u32)(rand() & 1) << 15) | ((u32)eb << 7) | (u32)(rand() & 127));
        }
        sb[i] = sbits | (bbits << 16);
    }
    u32 *d_sb, *d_first;
    unsigned long long *d_bad, bad = 0;
    hipMalloc(&d_sb, n * 4), hipMalloc(&d_bad, 8), hipMalloc(&d_first, 16);
    hipMemcpy(d_sb, sb.data(), n * 4, hipMemcpyHostToDevice), hipMemset(d_bad, 0, 8), hipMemset(d_first, 0, 16);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, d_sb, n, d_bad, d_first);
    u32 first[4];
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost), hipMemcpy(first, d_first, 16, hipMemcpyDeviceToHost);
    printf("part A: %d (scale, bias) pairs x 16 patterns x 4 pairs: mismatches vs fadd(fmul): cvt+fma %llu, scalar fma %llu, fma_mix %llu\n", n, bad & 0xFFFFF,
           (bad >> 20) & 0xFFFFF, bad >> 40);
    if (bad >> 40) printf("   first fma_mix mismatch: sb %08x got %08x want %08x (pattern/pair %u)\n", first[0], first[1], first[2], first[3]);
    // NaN payloads may differ legitimately; report finite-only mismatches separately would need another pass -- the realistic range (i >= 65536) has no NaN.

    // ---- Part B
    u32 *d_w, *d_out;
    hipMalloc(&d_w, 4096 * 4), hipMalloc(&d_out, 256 * 1024 * 4);
    std::vector<u32> hw(4096);
    for (auto &x : hw) x = (u32)rand() * 2654435761u;
    hipMemcpy(d_w, hw.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 2000;
    for (int waves = 4; waves <= 16; waves *= 2) {
        for (int v = 0; v < 3; ++v) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0, 0);
                const dim3 grid(256 * (waves > 8 ? 2 : 1)), block(64 * (waves > 8 ? 8 : waves));
                if (v == 0) hipLaunchKernelGGL(k_time<0>, grid, block, 0, 0, d_w, d_sb, iters, d_out);
                else if (v == 1) hipLaunchKernelGGL(k_time<1>, grid, block, 0, 0, d_w, d_sb, iters, d_out);
                else hipLaunchKernelGGL(k_time<2>, grid, block, 0, 0, d_w, d_sb, iters, d_out);
                hipEventRecord(e1, 0), hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("part B: %2d waves/CU variant %d: %.3f ms -> %.2f ns per word and wave (%.1f cycles at 2.4 GHz)\n", waves, v, best, best * 1e6 / (iters * 8.0),
                   best * 1e6 / (iters * 8.0) * 2.4);
        }
    }
    return 0;
}
