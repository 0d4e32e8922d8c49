// common.hpp -- shared host/device helpers for libpie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/pie_hip.h"

typedef unsigned int u32;
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

// ---------------------------------------------------------------- host side: error plumbing
namespace pie {
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
}  // namespace pie

#define PIE_HIP_TRY(expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return pie::fail(PIE_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

#define PIE_LAUNCH_CHECK() PIE_HIP_TRY(hipGetLastError())

#define PIE_REQUIRE(cond, code, msg)                 \
    do {                                             \
        if (!(cond)) return pie::fail((code), (msg)); \
    } while (0)

// Test / tuning switches: ONE process-wide table set through the C ABI (pie_set_knob, include/pie_hip.h); nothing under csrc/ reads the environment.
int pie_knob(int knob);  // current value; PIE_KNOB_DEFAULT (-1) = the built-in default

static inline bool pie_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// ---------------------------------------------------------------- device side: 16-bit float traits
// T-typed values travel as raw u16 bits; math is fp32.  codes2() turns two 4-bit codes (one per 16-bit
// half) into two T values OFFSET+q by OR-ing a magic exponent.
struct BF16 {
    // code pair -> dot2 operand: the bare nibble 0x000q IS a bf16 (denormal, q * 2^-133), and v_dot2c_f32_bf16
    // honours denormal inputs exactly (probed on MI355X), so no exponent has to be OR-ed in: the activations are
    // published pre-scaled by XSCALE = 2^64 (exact) and the group sum is rescaled by DSCALE = 2^69 (exact).
    // Products stay normal fp32 numbers for |q*x| > 2^-57; nothing overflows below |x| = 2^63.
    static constexpr float OFFSET = 0.0f, XSCALE = 0x1p64f, DSCALE = 0x1p69f;
    static __device__ __forceinline__ u32 codes2(u32 masked) { return masked; }
    // One code word (8 codes) against 4 packed activation pairs.  Gradual underflow makes 0x00q0 exactly 16*q*2^-133
    // as well (q >= 8 spills into the exponent LSB, which is precisely the denormal->normal transition), so code
    // positions 1 and 3 are masked in place and their chains carry a factor 16 that the caller divides out: one shift
    // per word instead of three.
    static constexpr float ODD_SCALE = 0.0625f;
    static __device__ __forceinline__ void dot_word(u32 w, u32 x0, u32 x1, u32 x2, u32 x3, float (&d)[4]);
    static constexpr float W2_S1 = 0.25f, W2_S2 = 0.0625f, W2_S3 = 0.015625f;  // W2S: what chains 1, 2, 3 are divided by (dot_word2 below)
    static __device__ __forceinline__ void dot_word2(u32 w, const u32 *x, float (&d)[4]);
    // the same in two steps, for kernels that multiply one code word with several activation rows: the four dot2 operands of a word
    // (the masking, 5 of the 9 instructions per word) are formed once and reused per row
    static __device__ __forceinline__ void word_ops(u32 w, u32 (&e)[4]) {
        const u32 w8 = w >> 8;
        e[0] = w & 0x000F000Fu, e[1] = w & 0x00F000F0u, e[2] = w8 & 0x000F000Fu, e[3] = w8 & 0x00F000F0u;
    }
    static __device__ __forceinline__ u32 bytes2(u32 masked) { return masked; }  // two 8-bit codes in the 16-bit halves, as operands
    static __device__ __forceinline__ float to_f32(u16 b) { return __builtin_bit_cast(float, (u32)b << 16); }
    static __device__ __forceinline__ u16 from_f32(float f) { return __builtin_bit_cast(u16, (__bf16)f); }  // v_cvt_pk_bf16_f32, RNE
    static __device__ __forceinline__ float dot2(u32 a, u32 b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
    }
};
// W2S: sixteen 2-bit codes of one word against eight activation pairs.  The masks stay in place (no shift): the pair at bit 2j of both halves is
// the bf16 number q * 4^j * 2^-133 (gradual underflow is continuous up to 0x00FF, common.hpp W8S), so chain j & 3 carries the factor 4^(j & 3) and the caller
// folds 1, 1/4, 1/16, 1/64 into the four chain sums: 8 v_and + 8 v_dot2c + 1 shift per 16 weights, the W4S loop's instruction count per weight.
__device__ __forceinline__ void BF16::dot_word2(u32 w, const u32 *x, float (&d)[4]) {
    const u32 w8 = w >> 8;
    d[0] = dot2(w & 0x00030003u, x[0], d[0]);
    d[1] = dot2(w & 0x000C000Cu, x[1], d[1]);
    d[2] = dot2(w & 0x00300030u, x[2], d[2]);
    d[3] = dot2(w & 0x00C000C0u, x[3], d[3]);
    d[0] = dot2(w8 & 0x00030003u, x[4], d[0]);
    d[1] = dot2(w8 & 0x000C000Cu, x[5], d[1]);
    d[2] = dot2(w8 & 0x00300030u, x[6], d[2]);
    d[3] = dot2(w8 & 0x00C000C0u, x[7], d[3]);
}
__device__ __forceinline__ void BF16::dot_word(u32 w, u32 x0, u32 x1, u32 x2, u32 x3, float (&d)[4]) {
    const u32 w8 = w >> 8;
    d[0] = dot2(w & 0x000F000Fu, x0, d[0]);
    d[1] = dot2(w & 0x00F000F0u, x1, d[1]);
    d[2] = dot2(w8 & 0x000F000Fu, x2, d[2]);
    d[3] = dot2(w8 & 0x00F000F0u, x3, d[3]);
}
struct F16 {
    // 0x6400|q is exactly 1024+q in f16; f16 results have 11 significant bits, too fine for the offset to
    // ride through the fp32 dot product, so it is removed exactly with one v_pk_add_f16 per code pair.
    static constexpr float OFFSET = 0.0f, XSCALE = 1.0f, DSCALE = 1.0f;
    static __device__ __forceinline__ u32 codes2(u32 masked) {
        f16x2_t v = __builtin_bit_cast(f16x2_t, masked | 0x64006400u);
        v = v - (f16x2_t){(_Float16)1024.0f, (_Float16)1024.0f};
        return __builtin_bit_cast(u32, v);
    }
    static constexpr float ODD_SCALE = 1.0f;
    static __device__ __forceinline__ u32 bytes2(u32 masked) { return codes2(masked); }  // 0x6400|q = 1024+q is exact up to q = 1023
    static __device__ __forceinline__ void word_ops(u32 w, u32 (&e)[4]) {
        e[0] = codes2(w & 0x000F000Fu), e[1] = codes2((w >> 4) & 0x000F000Fu), e[2] = codes2((w >> 8) & 0x000F000Fu), e[3] = codes2((w >> 12) & 0x000F000Fu);
    }
    static __device__ __forceinline__ void dot_word(u32 w, u32 x0, u32 x1, u32 x2, u32 x3, float (&d)[4]) {
        d[0] = dot2(codes2(w & 0x000F000Fu), x0, d[0]);
        d[1] = dot2(codes2((w >> 4) & 0x000F000Fu), x1, d[1]);
        d[2] = dot2(codes2((w >> 8) & 0x000F000Fu), x2, d[2]);
        d[3] = dot2(codes2((w >> 12) & 0x000F000Fu), x3, d[3]);
    }
    static constexpr float W2_S1 = 1.0f, W2_S2 = 1.0f, W2_S3 = 1.0f;  // W2S chain factors: every pair is shifted down, all chains at scale 1
    static __device__ __forceinline__ void dot_word2(u32 w, const u32 *x, float (&d)[4]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j & 3] = dot2(codes2((w >> (2 * j)) & 0x00030003u), x[j], d[j & 3]);
    }
    static __device__ __forceinline__ float to_f32(u16 b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ u16 from_f32(float f) { return __builtin_bit_cast(u16, (_Float16)f); }
    static __device__ __forceinline__ float dot2(u32 a, u32 b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
    }
};

template <class T> __device__ __forceinline__ float lo_f32(u32 p) { return T::to_f32((u16)(p & 0xffffu)); }
template <class T> __device__ __forceinline__ float hi_f32(u32 p) { return T::to_f32((u16)(p >> 16)); }
template <class T> __device__ __forceinline__ u32 pack2(float lo, float hi) {
    return (u32)T::from_f32(lo) | ((u32)T::from_f32(hi) << 16);
}
template <class T> __device__ __forceinline__ float round_T(float v) { return T::to_f32(T::from_f32(v)); }

// ---------------------------------------------------------------- wave64 reductions (DPP, no LDS)
// Sum over each 32-lane half of the wave; the result is valid in lanes 16..31 (low half) and 48..63 (high half).
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0x140, 0xF, 0xF, true);  // row_mirror -> every lane of a 16-row holds the row sum
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0x142, 0xA, 0xF, false); // row_bcast15 into rows 1 and 3
    return v;
}
// Cross-row exchanges without LDS (gfx950 v_permlane{16,32}_swap): swapping a value with itself leaves one partner's
// copy in r[0] and the other's in r[1] in BOTH lanes of a pair, so symmetric combines need no select.
__device__ __forceinline__ float xor16_sum(float v) {  // v + value of the lane 16 away (row 0<->1, 2<->3)
    const int vi = __builtin_bit_cast(int, v);
    auto r = __builtin_amdgcn_permlane16_swap(vi, vi, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const int vi = __builtin_bit_cast(int, v);
    auto r = __builtin_amdgcn_permlane32_swap(vi, vi, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const int vi = __builtin_bit_cast(int, v);
    auto r = __builtin_amdgcn_permlane16_swap(vi, vi, false, false);
    // v_med3_f32(a, b, +inf) = max(a, b) in ONE instruction: fmaxf() on values that come out of a bit cast costs two
    // extra canonicalising v_max each (the compiler must quiet a possible signalling NaN)
    return __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]), __builtin_inff());
}
__device__ __forceinline__ float xor32_max(float v) {
    const int vi = __builtin_bit_cast(int, v);
    auto r = __builtin_amdgcn_permlane32_swap(vi, vi, false, false);
    return __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]), __builtin_inff());
}
// Sum over each aligned group of 8 / 16 lanes (DPP inside one 16-lane row), result in every lane of the group.
__device__ __forceinline__ float lanes8_sum(float v) {
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    return v;
}
__device__ __forceinline__ float lanes16_sum(float v) {
    v = lanes8_sum(v);
    v += __builtin_amdgcn_update_dpp(0.0f, v, 0x140, 0xF, 0xF, true);  // row_mirror
    return v;
}
__device__ __forceinline__ float ror8(float v) {  // value of the lane 8 away inside the 16-lane row
    return __builtin_amdgcn_update_dpp(0.0f, v, 0x128, 0xF, 0xF, true);
}

// Full-wave sum / max, broadcast to every lane.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// W4S geometry (see include/pie_hip.h): one unit = row pair x 2048-wide K slice.
constexpr int W4S_UNIT_BYTES = 2304;
constexpr int W4S_SLICE_K = 2048;
static inline __host__ __device__ int w4s_slices(int K) { return (K + W4S_SLICE_K - 1) / W4S_SLICE_K; }
// W16S (dense 16-bit weights, same streaming shape): one unit = row pair x 512-wide K slice = 2 x [64 lanes x 16 B];
// lane l: row l >> 5, elements 512 s + 16 (l & 31) + 8 j of piece j.
constexpr int W16S_UNIT_BYTES = 2048;
constexpr int W16S_SLICE_K = 512;
static inline __host__ __device__ int w16s_slices(int K) { return (K + W16S_SLICE_K - 1) / W16S_SLICE_K; }
// W8S (MLX int8 group-64 triplets): like W4S one unit = row pair x 2048-wide K slice, lane l owns group 32 s + (l & 31) of row
// l >> 5, but its 64 codes are FOUR 16-byte pieces: [4 x 64 lanes x 16 B] codes + [64 x 4 B] {scale | bias << 16} = 4352 B.
// The bytes of every word are reordered to (c0, c2, c1, c3) so that w & 0x00FF00FF = codes (0, 1) and (w >> 8) & 0x00FF00FF =
// codes (2, 3) in the two 16-bit halves: a bare byte 0x00qq IS the bf16 number q * 2^-133 for every q <= 255 (denormal below
// 128, exponent field 1 above: gradual underflow is continuous), so the int4 path's operand trick carries over unchanged.
constexpr int W8S_UNIT_BYTES = 4352;
// W4S32 (MLX int4 group-32 triplets): the W4S unit with TWO {scale | bias << 16} words per lane -- its first code piece (32 codes) is one
// 32-wide group, its second the next: [2 x 64 lanes x 16 B] codes + [64 x 8 B] = 2560 B.
constexpr int W4S32_UNIT_BYTES = 2560;
// W8S32 (MLX int8 group-32 triplets): likewise the W8S unit with two {scale | bias << 16} words per lane (code pieces 0-1 / 2-3) = 4608 B.
constexpr int W8S32_UNIT_BYTES = 4608;
enum { FMT_W4S = 0, FMT_W16S = 1, FMT_W8S = 2, FMT_W4S32 = 3, FMT_W8S32 = 4, FMT_W2S = 5, FMT_W6S = 6 };
static inline __host__ __device__ constexpr int fmt_unit_bytes(int fmt) {
    if (fmt == FMT_W2S) return 1280;
    if (fmt == FMT_W6S) return 3328;
    return fmt == FMT_W16S ? W16S_UNIT_BYTES : (fmt == FMT_W8S ? W8S_UNIT_BYTES : (fmt == FMT_W4S32 ? W4S32_UNIT_BYTES : (fmt == FMT_W8S32 ? W8S32_UNIT_BYTES : W4S_UNIT_BYTES)));
}
// W2S (MLX int2 group-64 triplets, round 5): the same unit shape with ONE 16-byte code piece per lane (its group's 64 two-bit codes) --
// [64 lanes x 16 B] codes + [64 x 4 B] {scale | bias << 16} = 1280 B per row pair x 2048-wide K slice: 0.3125 B per weight, the checkpoint's own figure.
// Word t of a lane's piece holds codes 16 t .. 16 t + 15 of the group: the even ones in the low 16-bit half, the odd ones in the high half, pair j
// (codes 16 t + 2 j, + 1) at bits 2 j of both halves, so that (w >> 2 j) & 0x00030003 is one activation pair's two codes.
constexpr int W2S_UNIT_BYTES = 1280;
// W6S (MLX int6 group-64 triplets, round 5): a lane's 64 six-bit codes as TWO PLANES -- the low nibbles as the W4S unit's two code pieces, the high
// two bits as the W2S unit's one piece: [2 x 64 x 16 B] + [64 x 16 B] + [64 x 4 B] {scale | bias << 16} = 3328 B per row pair x 2048-wide K slice =
// 0.8125 B per weight, the checkpoint's own figure (MLX stores the codes as a byte-straddling bit stream, four to three bytes).  q = lo + 16 hi, so a
// group's dot product is the W4S dot of the low plane plus 16 times the W2S dot of the high plane: both loops unchanged, no straddling read in the stream.
constexpr int W6S_UNIT_BYTES = 3328;
constexpr int PIE_EMBED_W4G32 = 36, PIE_EMBED_W8G32 = 40;  // embedding_launch's `bits` for 4- / 8-bit codes in 32-wide groups (4 and 8 = the 64-wide group forms)
