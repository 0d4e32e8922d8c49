// w4m_gemm.hip -- int4 g=64 x T GEMMs on the MFMA units, weights read once in 4-bit form (MLX's qmm contract: weights dequantised to T,
// T x T products, fp32 accumulation, one rounding: mx.quantized_matmul for many rows, models/llama/language.py:83,108,127).
//
// Three kernels, one arithmetic, by row count:
//   6 .. 256 rows   k_w4r_gemm  (w4r_gemm.hpp, round 5): every CU streams its slab of tiles once and multiplies it into all the rows
//   > 256 rows      k_w4l2_gemm (one wave per SIMD, 256 accumulators in AGPRs, x by LDS-DMA) / k_w4l_gemm (8 waves, shapes the former does not take)
//   <= 32 rows      k_w4m_gemm  (round 2's first form: one workgroup per 32-column strip, x fragments straight from L2): what remains for
//                   K < 256 columns and as the tests' comparator (knob PIE_KNOB_W4R = 0); its persistent, LDS-staged and multi-strip
//                   siblings were replaced by k_w4r_gemm (19 us against 26 us on gate|up at 32 rows: EXPERIMENTS.md)
// Epilogues: store, SwiGLU on the interleaved gate|up rows, RoPE + cache append on the packed q|k|v rows, fp32 slabs of a K split.
//
// W4M layout (derived on the device from the W4S stream, cached next to it; same 0.5625 B/weight): tiles of 32 output rows x
// 64 columns (one quantisation group per row), 1152 B each, tile (nt, g) at ((nt * K/64) + g) * 1152:
//   bytes [0, 1024): lane l = 32 * kh + n holds 4 code words (16 B): word s = the 8 codes of row 32 nt + n, columns
//                    64 g + 16 s + 8 kh .. + 8, in the W4S word format (codes 2i, 2i+1 at nibble i of the low / high half)
//                    -- after dequantisation exactly the A fragment of v_mfma_f32_32x32x16 k-step s (row n, k-half kh);
//   bytes [1024, 1152): row n's (scale, bias) as one 32-bit word.
#include <cstdlib>
#include <map>
#include <mutex>

#include "prefill_attn.hpp"  // MfmaT, f32x16_t, DecState

#ifndef W4M_ABL
#define W4M_ABL 0  // developer ablation mask (tools/w4m_bench): 1 no dequantisation, 2 no x loads, 4 no MFMA; 0 in the product
#endif

constexpr int W4M_TILE_BYTES = 1152;
constexpr int W4M_WAVES = 8;
constexpr int W4M_WDEPTH = 8;  // weight tiles (16 B codes + 4 B scale/bias per lane) in flight per wave
constexpr int W4M_XDEPTH = 2;  //                x fragment sets (4 x 16 B per lane) in flight per wave

// W4S unit stream -> W4M tiles.  One thread per (tile, lane): pure word shuffle, no nibble work.
__global__ void __launch_bounds__(256) k_w4s_to_w4m(const u32 *w4s, int N, int K, int ns, u32 *w4m, int *wide) {
    const int groups = K >> 6;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)(N >> 5) * groups * 64) return;
    const int lane = (int)(i & 63), g = (int)((i >> 6) % groups), nt = (int)((i >> 6) / groups);
    const int n = lane & 31, kh = lane >> 5, r = 32 * nt + n;
    const u32 *unit = w4s + ((size_t)(r >> 1) * ns + (g >> 5)) * (W4S_UNIT_BYTES / 4);
    const int src_lane = (r & 1) * 32 + (g & 31);
    u32 *tile = w4m + ((size_t)nt * groups + g) * (W4M_TILE_BYTES / 4);
    u32 o[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int wk = 2 * s + kh;  // word (8 columns) inside the 64-column group
        o[s] = unit[(wk >> 2) * 256 + src_lane * 4 + (wk & 3)];
    }
    *reinterpret_cast<uint4 *>(tile + lane * 4) = make_uint4(o[0], o[1], o[2], o[3]);
    if (kh == 0) {
        const u32 sbw = unit[512 + src_lane];
        tile[256 + n] = sbw;
        // |scale| >= 2^100 read as bf16 (an f16 scale can only trip this falsely): outside w4r_dequant's domain (w4r_gemm.hpp)
        if (wide && ((sbw >> 7) & 0xFFu) >= 127u + 100u) atomicOr(wide, 1);
    }
}

// two fp32 -> one packed pair of T in a single v_cvt_pk_* (pack2<T> converts each half separately and ORs them: 3 instructions)
template <class T>
__device__ __forceinline__ u32 w4m_pack(float lo, float hi);
template <>
__device__ __forceinline__ u32 w4m_pack<BF16>(float lo, float hi) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(u32, __builtin_convertvector((f2_t){lo, hi}, b2_t));
}
template <>
__device__ __forceinline__ u32 w4m_pack<F16>(float lo, float hi) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(u32, __builtin_convertvector((f2_t){lo, hi}, h2_t));
}

#include "mfma_store.hpp"  // the LDS-transposed output epilogue of the one-wave-per-SIMD GEMMs

// 8 codes of one word -> 8 weights in T, the mx.dequantize arithmetic of k_dequant_w4s: T(fp32(s * q) + b), as the MFMA fragment
template <class T>
__device__ __forceinline__ uint4 w4m_dequant(u32 word, float s, float b) {
    u32 e = word & 0x0F0F0F0Fu, o = (word >> 4) & 0x0F0F0F0Fu;  // bytes: codes (0, 4, 1, 5) and (2, 6, 3, 7)
    // keep the two masked words as they are: folded back into per-code shift + and + v_cvt_f32_ubyte0 the conversion cost 4.7
    // VALU instructions per weight instead of one v_cvt_f32_ubyteN each
    asm volatile("" : "+v"(e), "+v"(o));
    const float c0 = (float)(e & 0xFFu), c4 = (float)((e >> 8) & 0xFFu), c1 = (float)((e >> 16) & 0xFFu), c5 = (float)(e >> 24);
    const float c2 = (float)(o & 0xFFu), c6 = (float)((o >> 8) & 0xFFu), c3 = (float)((o >> 16) & 0xFFu), c7 = (float)(o >> 24);
    // fmaf == fadd(fmul): s * q is exact in fp32 (a 16-bit float's <= 11 significant bits times a 4-bit code), so the separate
    // rounding of the product that mx.dequantize's fp32(s * q) + b implies never rounds -- one (packed) FMA per weight instead of two ops
    auto dq = [&](float q) { return __builtin_fmaf(s, q, b); };
    return make_uint4(w4m_pack<T>(dq(c0), dq(c1)), w4m_pack<T>(dq(c2), dq(c3)), w4m_pack<T>(dq(c4), dq(c5)), w4m_pack<T>(dq(c6), dq(c7)));
}

// The same conversion with the scale and the bias held as explicit register PAIRS {s, s}, {b, b} (opaque to the optimiser): with
// scalars the compiler forms v_pk_fma_f32 with an op_sel broadcast from whatever pair the scalar happens to sit in, and when the
// neighbouring register is the destination of a load still in flight the instruction waits for that load (seen in k_w4l2_gemm).
typedef float w4m_f32x2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ uint4 w4m_dequant_pk(u32 word, w4m_f32x2 s2, w4m_f32x2 b2) {
    // pin the word to THIS point of the step sequence: the masks below are plain VALU ops that the scheduler otherwise hoists to
    // right behind the load of the word, three steps earlier, where they wait for it (volatile statements keep their order)
    asm volatile("" : "+v"(word));
    u32 e = word & 0x0F0F0F0Fu, o = (word >> 4) & 0x0F0F0F0Fu;  // bytes: codes (0, 4, 1, 5) and (2, 6, 3, 7)
    asm volatile("" : "+v"(e), "+v"(o));
    const w4m_f32x2 q01 = {(float)(e & 0xFFu), (float)((e >> 16) & 0xFFu)}, q45 = {(float)((e >> 8) & 0xFFu), (float)(e >> 24)};
    const w4m_f32x2 q23 = {(float)(o & 0xFFu), (float)((o >> 16) & 0xFFu)}, q67 = {(float)((o >> 8) & 0xFFu), (float)(o >> 24)};
    const w4m_f32x2 r01 = __builtin_elementwise_fma(s2, q01, b2), r23 = __builtin_elementwise_fma(s2, q23, b2);
    const w4m_f32x2 r45 = __builtin_elementwise_fma(s2, q45, b2), r67 = __builtin_elementwise_fma(s2, q67, b2);
    return make_uint4(w4m_pack<T>(r01.x, r01.y), w4m_pack<T>(r23.x, r23.y), w4m_pack<T>(r45.x, r45.y), w4m_pack<T>(r67.x, r67.y));
}

// Sum of the 8 waves' partial tiles (fixed order), rounded once, stored for the live rows.
// Arguments of the q|k|v epilogue (rope = 2): RoPE on the q and k pairs and the cache append, exactly k_rope_append_rows (prefill.hip).
struct W4mRope {
    const float *rope_cs;               // [M, HD / 2, 2] (cos, sin) of every row's position (k_rope_cs_rows)
    const DecState *state;              // single sequence: row m sits at state->pos + m, cache capacity state->cap ...
    const int *ctx_len;                 // ... or a batch of sequences: row m at ctx_len[m] - 1 (< 0: idle slot)
    const unsigned long long *kv_table; // per-layer K / V buffer (or slab) bases ...
    u16 *slab;                          // ... or this layer's slab directly (batch)
    const int *block_table;             // paged KV (nullable): table row m * bt_stride
    int bt_stride, n_pages, layer, n_layers, n_heads, n_kv_heads, HD, traditional;
    u16 *q_out;                         // [M, n_heads, HD]
    const u16 *bias;                    // the Linear's bias (packed order), nullable
    size_t i8_page_bytes;               // != 0 (with slab): the pages are int8 with per-head fp16 scales (paged_i8.hip): K / V are quantised on the way in
};

// one packed column pair (R, R + 1) of x-row m: the fp32 sums a, b -> T, (+ bias), then RoPE + store / cache append
template <class T>
__device__ __forceinline__ void w4m_rope_pair(float a, float b, int m, int R, const W4mRope &r) {
    a = round_T<T>(a), b = round_T<T>(b);
    if (r.bias) a = round_T<T>(a + T::to_f32(r.bias[R])), b = round_T<T>(b + T::to_f32(r.bias[R + 1]));
    const int pos = r.ctx_len ? r.ctx_len[m] - 1 : r.state->pos + m, HD = r.HD, half = HD >> 1;
    if (pos < 0) return;
    int cap = r.ctx_len ? 64 : r.state->cap, kvrow = pos;
    u16 *kdst = r.slab ? r.slab : reinterpret_cast<u16 *>(r.kv_table[r.layer]);
    u16 *vdst = r.slab ? r.slab + (size_t)r.n_kv_heads * 64 * HD : reinterpret_cast<u16 *>(r.kv_table[r.n_layers + r.layer]);
    char *page8 = nullptr;  // int8 pages: this row's page (k_rope_append_rows' arithmetic: clamp(rint(x / s), -127, 127) of the T-rounded element)
    if (r.block_table) {
        const int *bt = r.block_table + (size_t)m * r.bt_stride;
        const unsigned pg = min((unsigned)bt[pos >> 6], (unsigned)r.n_pages - 1u);
        const size_t pg_off = (size_t)pg * 2 * 64 * r.n_kv_heads * HD;
        kdst += pg_off, vdst += pg_off, cap = 64, kvrow = pos & 63;
        if (r.i8_page_bytes && r.slab) page8 = reinterpret_cast<char *>(r.slab) + (size_t)pg * r.i8_page_bytes;
    }
    const size_t blk8 = (size_t)r.n_kv_heads * 64 * HD;
    auto q8 = [](float x, float sc) {
        float q = rintf(x / sc);
        q = q < -127.0f ? -127.0f : (q > 127.0f ? 127.0f : q);
        return (signed char)(q == q ? (int)q : 0);
    };
    auto f16f = [](u16 h) { return (float)__builtin_bit_cast(_Float16, h); };
    const int q_cols = r.n_heads * HD, k_cols = r.n_kv_heads * HD;
    if (R < q_cols + k_cols) {
        const int rr = R < q_cols ? R : R - q_cols;
        const int head = rr / HD, ii = (rr % HD) >> 1;
        const float2 csn = *reinterpret_cast<const float2 *>(r.rope_cs + ((size_t)m * half + ii) * 2);
        u16 *dst = R < q_cols ? r.q_out + ((size_t)m * r.n_heads + head) * HD : kdst + ((size_t)head * cap + kvrow) * HD;
        const int i0 = r.traditional ? 2 * ii : ii, i1 = r.traditional ? 2 * ii + 1 : ii + half;
        const u16 o0 = T::from_f32(__fsub_rn(__fmul_rn(a, csn.x), __fmul_rn(b, csn.y))), o1 = T::from_f32(__fadd_rn(__fmul_rn(a, csn.y), __fmul_rn(b, csn.x)));
        if (page8 && R >= q_cols) {
            const float sk = f16f(reinterpret_cast<const u16 *>(page8 + 2 * blk8)[head]);
            signed char *kb = reinterpret_cast<signed char *>(page8) + ((size_t)head * 64 + kvrow) * HD;
            kb[i0] = q8(T::to_f32(o0), sk), kb[i1] = q8(T::to_f32(o1), sk);
        } else dst[i0] = o0, dst[i1] = o1;
    } else {
        const int rr = R - q_cols - k_cols;
        if (page8) {
            const int head = rr / HD;
            const float sv = f16f(reinterpret_cast<const u16 *>(page8 + 2 * blk8)[r.n_kv_heads + head]);
            signed char *vb = reinterpret_cast<signed char *>(page8) + blk8 + ((size_t)head * 64 + kvrow) * HD + rr % HD;
            vb[0] = q8(a, sv), vb[1] = q8(b, sv);
        } else *reinterpret_cast<u32 *>(vdst + ((size_t)(rr / HD) * cap + kvrow) * HD + rr % HD) = pack2<T>(a, b);
    }
}

template <class T>
__device__ __forceinline__ void w4m_epilogue_rope(float (*s_red)[16][64], int nt, int M, const W4mRope &r) {
    const int i = 2 * (threadIdx.x >> 6), l = threadIdx.x & 63;
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int w = 0; w < W4M_WAVES; ++w) a += s_red[w][i][l], b += s_red[w][i + 1][l];
    const int m = l & 31, R = 32 * nt + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);  // packed columns (R, R + 1)
    if (m >= M) return;
    w4m_rope_pair<T>(a, b, m, R, r);
}

// ... or (swiglu) the MLP activation: the packed gate|up matrix interleaves its rows (2i, 2i + 1) = (gate_i, up_i), so a strip holds
// 16 complete pairs and act[m][16 nt + j] = T(T(silu(g)) * u) with g, u the T-rounded (and biased) Linear outputs -- exactly what
// the GEMM followed by the bias and SwiGLU row kernels produce, without writing and re-reading the [M, 2I] block.
template <class T>
__device__ __forceinline__ void w4m_epilogue(float (*s_red)[16][64], int nt, int M, int N, u16 *y, bool swiglu, const u16 *bias) {
    if (swiglu) {  // one (gate, up) pair per thread: accumulator registers (2p, 2p + 1) of lane l
        const int i = 2 * (threadIdx.x >> 6), l = threadIdx.x & 63;
        float g = 0.0f, u = 0.0f;
#pragma unroll
        for (int w = 0; w < W4M_WAVES; ++w) g += s_red[w][i][l], u += s_red[w][i + 1][l];
        const int mm = l & 31, nn = 32 * nt + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);  // even: the gate row; nn + 1: its up row
        g = round_T<T>(g), u = round_T<T>(u);
        if (bias) g = round_T<T>(g + T::to_f32(bias[nn])), u = round_T<T>(u + T::to_f32(bias[nn + 1]));
        if (mm < M) y[(size_t)mm * (N >> 1) + (nn >> 1)] = T::from_f32(round_T<T>(g / (1.0f + expf(-g))) * u);
        return;
    }
    for (int o = threadIdx.x; o < 16 * 64; o += W4M_WAVES * 64) {
        const int i = o >> 6, l = o & 63;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < W4M_WAVES; ++w) v += s_red[w][i][l];
        const int mm = l & 31, nn = 32 * nt + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);  // accumulator register i <-> A row (i & 3) + 8 (i >> 2) + 4 kh
        if (mm < M) y[(size_t)mm * N + nn] = T::from_f32(v);
    }
}

// y[M, N] = x[M, K] . dequant(W)[N, K]^T, M <= 32.  grid = N / 32 workgroups of 8 waves.
// Two rings per wave: the weight tiles come from HBM (~2 us away) and cost 5 registers per slot -> W4M_WDEPTH = 8 slots in flight;
// the x fragments come from L2 and cost 16 registers per slot -> 2 slots.
template <class T>
__global__ void __launch_bounds__(W4M_WAVES * 64) k_w4m_gemm(const char *w4m, const u16 *x, int M, int N, int K, u16 *y, int swiglu, const u16 *bias, const W4mRope rope) {
    __shared__ float s_red[W4M_WAVES][16][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kh = lane >> 5, all_groups = K >> 6;
    const int nt = blockIdx.x;
    const int g_lo = 0, groups = all_groups;
    const char *strip = w4m + ((size_t)nt * all_groups + g_lo) * W4M_TILE_BYTES;
    const int m = n < M ? n : M - 1;                          // B-operand column = x row (columns >= M are never stored)
    const u16 *xrow = x + (size_t)m * K + 8 * kh + (size_t)g_lo * 64;
    const int my_groups = groups > wave ? (groups - wave + W4M_WAVES - 1) / W4M_WAVES : 0;

    typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
    uint4 cw[W4M_WDEPTH], xf[W4M_XDEPTH][4];
    u32 sb[W4M_WDEPTH];
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // iteration `it` of this wave -> tile group (clamped to a valid one: never branch around a load)
    auto group_of = [&](int it) {
        it = it < my_groups ? it : my_groups - 1;
        int g = wave + (it < 0 ? 0 : it) * W4M_WAVES;
        return g < groups ? g : groups - 1;
    };
#define W4M_WISSUE(d, it)                                                                                          \
    {                                                                                                              \
        const char *tile_ = strip + (size_t)group_of(it) * W4M_TILE_BYTES;                                          \
        const nt_u32x4 c_ = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(tile_) + lane);           \
        cw[d] = make_uint4(c_.x, c_.y, c_.z, c_.w);                                                                \
        sb[d] = __builtin_nontemporal_load(reinterpret_cast<const u32 *>(tile_ + 1024) + n);                        \
    }
#define W4M_XISSUE(d, it)                                                                                          \
    {                                                                                                              \
        const u16 *xp_ = xrow + (size_t)group_of(it) * 64;                                                          \
        xf[d][0] = *reinterpret_cast<const uint4 *>(xp_), xf[d][1] = *reinterpret_cast<const uint4 *>(xp_ + 16);    \
        xf[d][2] = *reinterpret_cast<const uint4 *>(xp_ + 32), xf[d][3] = *reinterpret_cast<const uint4 *>(xp_ + 48); \
    }
#pragma unroll
    for (int d = 0; d < W4M_WDEPTH; ++d) W4M_WISSUE(d, d)
#pragma unroll
    for (int d = 0; d < W4M_XDEPTH; ++d) W4M_XISSUE(d, d)
    for (int base = 0; base < my_groups; base += W4M_WDEPTH) {
#pragma unroll
        for (int d = 0; d < W4M_WDEPTH; ++d) {
            const int it = base + d;
            constexpr int XD = W4M_XDEPTH;
            if (it < my_groups) {  // wave-uniform
                const float s = lo_f32<T>(sb[d]), b = hi_f32<T>(sb[d]);
                const u32 words[4] = {cw[d].x, cw[d].y, cw[d].z, cw[d].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 af = (W4M_ABL & 1) ? make_uint4(words[k], words[k] ^ sb[d], words[k] + 1, sb[d]) : w4m_dequant<T>(words[k], s, b);
                    if (W4M_ABL & 4) acc[k] += __builtin_bit_cast(float, af.x ^ af.y ^ af.z ^ af.w ^ xf[d % XD][k].x);
                    else acc = MfmaT<T>::run(af, xf[d % XD][k], acc);
                }
            }
            W4M_WISSUE(d, it + W4M_WDEPTH)
            if (!(W4M_ABL & 2)) W4M_XISSUE(d % XD, it + XD)
        }
    }
#undef W4M_WISSUE
#undef W4M_XISSUE
    // partial tiles -> LDS, summed in wave order (deterministic), rounded once, stored for the M live columns
#pragma unroll
    for (int i = 0; i < 16; ++i) s_red[wave][i][lane] = acc[i];
    __syncthreads();
    if (swiglu == 2) w4m_epilogue_rope<T>(s_red, nt, M, rope);
    else w4m_epilogue<T>(s_red, nt, M, N, y, swiglu != 0, bias);
}

// ---------------------------------------------------------------- MANY rows (33 .. thousands): the prompt GEMM
// y[M, N] = x[M, K] . dequant(W)[N, K]^T on the same W4M tiles, MFMA-bound (round 1 expanded the weights to a 16-bit copy --
// 15 GB on the 8B model -- and called hipBLASLt).  One workgroup = 256 rows of x by 256 output columns:
//   * wave w owns the 32-column strip nb * 8 + w: its W4M tiles arrive straight from HBM / L2 in registers (1 KiB coalesced per
//     tile, ring of W4L_WDEPTH) and are dequantised IN REGISTERS into the A fragments of v_mfma_f32_32x32x16 (w4m_dequant: the
//     T(fp32(s q) + b) of mx.dequantize) -- the weights never see LDS and move at 0.5625 B each;
//   * the x tile [256 rows, 64 columns] of a K step is staged ONCE per workgroup in LDS (double-buffered, 144-byte rows:
//     16-byte fragment reads of 32 consecutive rows hit distinct bank quads) and read by all 8 waves: per K step and wave
//     32 MFMAs (8 row blocks x 4 k-steps) against 32 ds_read_b128 and ~110 VALU instructions of dequantisation per lane, i.e. the
//     matrix pipe is the busiest unit and the conversion rides in its shadow on the partner wave of the SIMD;
//   * 128 accumulator registers per lane (8 row blocks x 16), one 8-wave workgroup per CU.
// Prompts of up to 64 / 128 rows use 64- / 128-row tiles (MB = 2 / 4) and split K over blockIdx.z so that the grid still fills the chip.
#ifndef W4L_ABL
#define W4L_ABL 0  // developer ablation mask (scripts/bench_w4l.py): 1 no dequantisation, 2 no LDS fragment reads, 4 no x staging; 0 in the product
#endif
constexpr int W4L_XROW = 128 + 16;  // LDS bytes per staged x row: 64 columns + pad
constexpr int W4L_WDEPTH = 4;       // weight tiles in flight per wave

// MB: 32-row blocks of x per workgroup (tile = 32 MB rows x 256 columns).  gridDim.z > 1: the K groups are split over blockIdx.z
// and the fp32 partial tiles go to `part` [z][M][N] (summed in z order by k_w4l_reduce: deterministic) -- prompts of a few dozen to
// a few hundred rows leave too few 256-column workgroups to fill 256 CUs otherwise.
template <class T, int MB>
__global__ void __launch_bounds__(W4M_WAVES * 64) k_w4l_gemm(const char *w4m, const u16 *x, int M, int N, int K, u16 *y, float *part) {
    constexpr int MT = 32 * MB;
    constexpr int PPT = MT * 8 / (W4M_WAVES * 64);  // 16-byte x pieces per thread and K step (MB = 2: 1, 4: 2, 8: 4)
    __shared__ __attribute__((aligned(16))) char s_x[2][MT * W4L_XROW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kh = lane >> 5, all_groups = K >> 6;
    const int per_z = (all_groups + (int)gridDim.z - 1) / (int)gridDim.z;
    const int g_lo = blockIdx.z * per_z, g_hi = min(all_groups, g_lo + per_z);
    const int groups = g_hi - g_lo;  // >= 1 (launcher)
    const int m0 = blockIdx.y * MT;
    const int rows = M - m0 < MT ? M - m0 : MT;  // live rows of this tile
    const int nt = blockIdx.x * W4M_WAVES + wave;
    const bool has_strip = nt * 32 < N;  // wave-uniform; an idle wave still stages x and joins the barriers
    const char *strip = w4m + ((size_t)(has_strip ? nt : 0) * all_groups + g_lo) * W4M_TILE_BYTES;
    x += (size_t)g_lo * 64;

    typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
    uint4 cw[W4L_WDEPTH];
    u32 sb[W4L_WDEPTH];
#define W4L_WISSUE(d, g)                                                                                   \
    {                                                                                                      \
        const char *tile_ = strip + (size_t)((g) < groups ? (g) : groups - 1) * W4M_TILE_BYTES;             \
        const nt_u32x4 c_ = *(reinterpret_cast<const nt_u32x4 *>(tile_) + lane);                            \
        cw[d] = make_uint4(c_.x, c_.y, c_.z, c_.w);                                                        \
        sb[d] = *(reinterpret_cast<const u32 *>(tile_ + 1024) + n);                                         \
    }
    // x staging: MT rows x 8 pieces of 16 bytes per K step; piece p = tid + 512 j -> row p >> 3, piece p & 7
    uint4 xs[PPT];
    auto x_fetch = [&](int g) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int p = threadIdx.x + 512 * j, r = p >> 3, c = p & 7;
            xs[j] = r < rows ? *reinterpret_cast<const uint4 *>(x + (size_t)(m0 + r) * K + g * 64 + c * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto x_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int p = threadIdx.x + 512 * j;
            *reinterpret_cast<uint4 *>(s_x[buf] + (p >> 3) * W4L_XROW + (p & 7) * 16) = xs[j];
        }
    };
#pragma unroll
    for (int d = 0; d < W4L_WDEPTH; ++d) W4L_WISSUE(d, d)
    x_fetch(0);
    x_store(0);
    if (groups > 1) x_fetch(1);

    f32x16_t acc[MB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mi][i] = 0.0f;
    // A fragments of the CURRENT K step; the next step's are dequantised between this step's MFMAs (the barrier puts the two waves
    // of a SIMD in phase, so conversion ahead of the MFMAs would idle the matrix pipe on both)
    uint4 af[4];
    {
        const float s0 = lo_f32<T>(sb[0]), b0 = hi_f32<T>(sb[0]);
        const u32 w0[4] = {cw[0].x, cw[0].y, cw[0].z, cw[0].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) af[k] = w4m_dequant<T>(w0[k], s0, b0);
    }
    for (int base = 0; base < groups; base += W4L_WDEPTH) {
#pragma unroll
        for (int d = 0; d < W4L_WDEPTH; ++d) {
            const int g = base + d;
            if (g < groups) {  // uniform for the workgroup
                __syncthreads();  // image g is complete; nobody reads image g + 1's buffer (= g - 1's) any more
                if (!(W4L_ABL & 4)) {
                    if (g + 1 < groups) x_store((g + 1) & 1);
                    if (g + 2 < groups) x_fetch(g + 2);
                }
                const int dn = (d + 1) % W4L_WDEPTH;  // ring slot of tile g + 1 (a constant after unrolling)
                const float sn = lo_f32<T>(sb[dn]), bn = hi_f32<T>(sb[dn]);
                const u32 wn[4] = {cw[dn].x, cw[dn].y, cw[dn].z, cw[dn].w};
                uint4 afn[4];
                const char *xr = s_x[g & 1] + n * W4L_XROW + kh * 16;
                // 4 MB steps t = (k, mi); B fragments ride a 4-deep register ring so an LDS read has 4 MFMAs to land
                constexpr int NS = 4 * MB;
                uint4 bq[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) bq[t] = *reinterpret_cast<const uint4 *>(xr + (t % MB) * 32 * W4L_XROW + 32 * (t / MB));
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    const int k = t / MB, mi = t % MB;
                    acc[mi] = MfmaT<T>::run(af[k], bq[t & 3], acc[mi]);
                    if (t + 4 < NS && !(W4L_ABL & 2)) bq[t & 3] = *reinterpret_cast<const uint4 *>(xr + ((t + 4) % MB) * 32 * W4L_XROW + 32 * ((t + 4) / MB));
                    if (mi == 0) afn[k] = (W4L_ABL & 1) ? make_uint4(wn[k], wn[k] ^ sb[dn], wn[k] + 1, sb[dn]) : w4m_dequant<T>(wn[k], sn, bn);  // one word of the next tile per k-step, in the MFMAs' shadow
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) af[k] = afn[k];
                W4L_WISSUE(d, g + W4L_WDEPTH)
            }
        }
    }
#undef W4L_WISSUE
    if (!has_strip) return;
    // accumulator register i of lane l <-> output column 32 nt + (i & 3) + 8 (i >> 2) + 4 kh, row m0 + 32 mi + (l & 31): four
    // consecutive columns per register quad -> one 8-byte (T) or 16-byte (fp32 partial) store
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
        const int m = 32 * mi + n;
        if (m < rows) {
            const size_t o = (size_t)(m0 + m) * N + 32 * nt + 4 * kh;
            if (part) {
                float *pr = part + (size_t)blockIdx.z * M * N + o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(pr + 8 * q) = make_float4(acc[mi][4 * q], acc[mi][4 * q + 1], acc[mi][4 * q + 2], acc[mi][4 * q + 3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint2 *>(y + o + 8 * q) = make_uint2(w4m_pack<T>(acc[mi][4 * q], acc[mi][4 * q + 1]), w4m_pack<T>(acc[mi][4 * q + 2], acc[mi][4 * q + 3]));
            }
        }
    }
}

// ---------------------------------------------------------------- the 256-row form: ONE wave per SIMD with the whole register file
// k_w4l_gemm above keeps 8 waves of 256 registers; each reads a B fragment (x) from LDS per MFMA and stages x through registers.
// Measured ablations (scripts/bench_w4l.py, M = 4096): that x path costs 18 % of the kernel.  This form halves it and takes it off
// the VALU / VGPR path:  4 waves x 512 registers (256 accumulators in AGPRs), wave tile = 256 rows x 64 columns (TWO W strips), so
// one B fragment feeds two MFMAs; the x tile [256 rows x 64 k] goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no
// ds_write), its 128-byte rows XOR-swizzled on the SOURCE side (chunk c of row r lands in slot c ^ ((r >> 1) & 7): the reads of a
// ds_read_b128 lane group then cover all 64 banks; the DMA's destination is lane-linear, so the permutation has to be applied to the
// per-lane global address).  Everything a step issues (the next x tile, the weight tiles two steps ahead) has the whole step --
// 64 MFMAs = 2048 matrix-pipe cycles -- to land before the step's closing barrier.
#ifndef W4L2_SPREAD
#define W4L2_SPREAD 4  // first MFMA iteration of a step after which its memory issue starts (0 = all of it in front of the MFMAs: 2-3 % slower)
#endif
#ifndef W4L2_ABL
#define W4L2_ABL 0  // developer ablation mask (0 in the product): 1 no conversion, 2 no barrier, 4 no x DMA, 8 no LDS fragment reads, 16 no weight loads
#endif
typedef __attribute__((address_space(3))) void w4l_lds_void;
typedef __attribute__((address_space(1))) const void w4l_glb_void;

// SWIGLU: the packed gate|up matrix (columns (2 i, 2 i + 1) = (gate_i, up_i)): y is the MLP activation [M, N / 2] = T(silu(T(gate)) * T(up))
// (language.py:127), which saves the [M, N] round trip and the row kernel's launch (60 us per layer at 4096 rows); needs the whole K here.
// MB: 32-row blocks of x per workgroup (row tile 64 / 128 / 256): medium prompts reach ~192 workgroups with smaller row tiles instead of
// deep K splits, whose fp32 partial tiles cost a reduce pass
template <class T, int MB, bool SWIGLU = false>
__global__ void __launch_bounds__(256) k_w4l2_gemm(const char *w4m, const u16 *x, int M, int N, int K, u16 *y, float *part) {
    constexpr int MT = 32 * MB, XJ = MB;  // rows per tile; DMA instructions per wave and tile (MT / 8 row groups over 4 waves)
    constexpr int OUT_BYTES = 4 * MT * (SWIGLU ? 64 : 128);  // the epilogue's output tile (mfma_store.hpp) reuses -- and outgrows -- the two x buffers
    __shared__ __attribute__((aligned(1024))) char s_x[OUT_BYTES / (MT * 128) > 2 ? OUT_BYTES / (MT * 128) : 2][MT * 128];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kh = lane >> 5, all_groups = K >> 6;
    const int per_z = (all_groups + (int)gridDim.z - 1) / (int)gridDim.z;
    const int g_lo = blockIdx.z * per_z, g_hi = min(all_groups, g_lo + per_z);
    const int groups = g_hi - g_lo;  // >= 1 (launcher)
    const int m0 = blockIdx.y * MT;
    const int rows = M - m0 < MT ? M - m0 : MT;
    const int nt0 = (blockIdx.x * 4 + wave) * 2;  // this wave's two strips: nt0, nt0 + 1
    const int n_strips = N >> 5;
    const bool has0 = nt0 < n_strips, has1 = nt0 + 1 < n_strips;  // wave-uniform; idle waves still stage x and join the barriers
    const char *strip0 = w4m + ((size_t)(has0 ? nt0 : 0) * all_groups + g_lo) * W4M_TILE_BYTES;
    const char *strip1 = w4m + ((size_t)(has1 ? nt0 + 1 : 0) * all_groups + g_lo) * W4M_TILE_BYTES;

    // x staging: wave w moves rows [8 MB w, 8 MB (w + 1)) of the tile with MB DMA instructions of 8 rows each
    const u16 *xsrc[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int r = 8 * MB * wave + 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int rr = r < rows ? r : rows - 1;  // ragged tile: rows past the end repeat the last one (never stored)
        xsrc[j] = x + (size_t)(m0 + rr) * K + (size_t)g_lo * 64 + c * 8;
    }
    // The DMA is issued through an asm statement on purpose: issued through __builtin_amdgcn_global_load_lds the compiler knows it
    // writes LDS, cannot tell the two tile buffers apart, and puts s_waitcnt vmcnt(0) in front of the step's first ds_read -- the
    // transfer it was meant to overlap.  (M0 = LDS destination base of the wave; the compiler reserves M0, so it is saved and
    // restored inside the statement.)  The step's closing barrier is preceded by an explicit vmcnt(0).
    auto x_issue1 = [&](int g, int buf, int j) {
        const u16 *src = xsrc[j] + (size_t)g * 64;
        const unsigned dst = (unsigned)(size_t)(w4l_lds_void *)(s_x[0]) + (unsigned)(buf * (MT * 128) + (8 * MB * wave + 8 * j) * 128);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(dst)
                     : "memory");
    };
    auto x_issue = [&](int g, int buf) {
#pragma unroll
        for (int j = 0; j < XJ; ++j) x_issue1(g, buf, j);
    };
    typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
    // raw weight tiles: tile j lives in ring slot j % 4 from its issue (step j - 4) to its conversion (during step j - 1)
    uint4 cw[4][2];  // [slot][strip]
    u32 sb[4][2];
    auto w_issue = [&](int slot, int g) {
        const size_t o = (size_t)(g < groups ? g : groups - 1) * W4M_TILE_BYTES;
        const nt_u32x4 c0 = *(reinterpret_cast<const nt_u32x4 *>(strip0 + o) + lane);
        const nt_u32x4 c1 = *(reinterpret_cast<const nt_u32x4 *>(strip1 + o) + lane);
        cw[slot][0] = make_uint4(c0.x, c0.y, c0.z, c0.w), cw[slot][1] = make_uint4(c1.x, c1.y, c1.z, c1.w);
        sb[slot][0] = *(reinterpret_cast<const u32 *>(strip0 + o + 1024) + n);
        sb[slot][1] = *(reinterpret_cast<const u32 *>(strip1 + o + 1024) + n);
    };
    x_issue(0, 0);
#pragma unroll
    for (int d = 0; d < 4; ++d) w_issue(d, d);
    // an (empty) statement with an AGPR operand: without any, hipcc marks the kernel "no AGPRs needed" and may select the VGPR form of
    // the MFMAs, using the AGPR half of the file as a spill area
    asm volatile("" : : "a"(0.0f));

    f32x16_t acc[2][MB];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s2][mi][i] = 0.0f;
    uint4 af[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const float s0 = lo_f32<T>(sb[0][s2]), b0 = hi_f32<T>(sb[0][s2]);
        const u32 w0[4] = {cw[0][s2].x, cw[0][s2].y, cw[0][s2].z, cw[0][s2].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) af[s2][k] = w4m_dequant<T>(w0[k], s0, b0);
    }
    // B fragment of (row block mi, k-step k): row 32 mi + n, 16-byte chunk 2 k + kh, swizzled by the row
    int xoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xoff[k] = n * 128 + (((2 * k + kh) ^ ((n >> 1) & 7)) << 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of tile 0 has landed
    __syncthreads();

    // groups % 4 == 0 (launcher): four copies of the step with static ring slots and tile buffers and NO control flow around the
    // accumulators (with a per-step `if (g < groups)` the 256 accumulators went through phi copies and 180 registers spilled).
    // Queue order per step: [x DMA of tile g + 1][raw weight tile g + 4]; the closing wait leaves only that weight tile in flight.
    for (int base = 0; base < groups; base += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int g = base + d;
#if !W4L2_SPREAD
            if (!(W4L2_ABL & 4)) x_issue(g + 1 < groups ? g + 1 : g, (d + 1) & 1);  // that buffer was last read in step g - 1, closed by its barrier (last step: a harmless re-read)
            if (!(W4L2_ABL & 16)) w_issue(d, g + 4);            // slot d held tile g, converted during step g - 1
#endif
            const int dn = (d + 1) & 3;                         // slot of tile g + 1: converted during this step
            w4m_f32x2 sc[2], bi[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float sv = lo_f32<T>(sb[dn][s2]), bv = hi_f32<T>(sb[dn][s2]);
                sc[s2] = (w4m_f32x2){sv, sv}, bi[s2] = (w4m_f32x2){bv, bv};
                asm volatile("" : "+v"(sc[s2]), "+v"(bi[s2]));
            }
            const char *xb = s_x[d & 1];
            uint4 afn[2][4];
            uint4 bq[4];
            constexpr int NS = 4 * MB, CV = MB / 2;  // (k-step, row block) pairs of a step; pairs per word conversion (8 words per step)
#pragma unroll
            for (int t = 0; t < 4; ++t) bq[t] = *reinterpret_cast<const uint4 *>(xb + (t % MB) * 4096 + xoff[t / MB]);
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int k = t / MB, mi = t % MB;
                acc[0][mi] = MfmaT<T>::run(af[0][k], bq[t & 3], acc[0][mi]);
                acc[1][mi] = MfmaT<T>::run(af[1][k], bq[t & 3], acc[1][mi]);
#if W4L2_SPREAD
                // the step's memory issue rides between the MFMAs instead of in front of them (each DMA is 5 scalar instructions + the
                // load: at the top of the step the matrix pipe idles while they issue): x tile g + 1 one piece per iteration, then
                // the raw weight tile g + 4 -- the queue order [DMA x MB][weights x 4] the closing vmcnt(4) relies on is unchanged
                constexpr int SP = W4L2_SPREAD + XJ < NS ? W4L2_SPREAD : 1;
                if (t >= SP && t < SP + XJ && !(W4L2_ABL & 4)) x_issue1(g + 1 < groups ? g + 1 : g, (d + 1) & 1, t - SP);
                if (t == SP + XJ && !(W4L2_ABL & 16)) w_issue(d, g + 4);
#endif
                if (t + 4 < NS && !(W4L2_ABL & 8)) bq[t & 3] = *reinterpret_cast<const uint4 *>(xb + ((t + 4) % MB) * 4096 + xoff[(t + 4) / MB]);
                if (t % CV == 0) {  // one word of the next tiles per CV pairs: 8 conversions in the shadow of the step's MFMAs
                    const int q = t / CV, s2 = q & 1, kk = q >> 1;
                    const u32 wn[4] = {cw[dn][s2].x, cw[dn][s2].y, cw[dn][s2].z, cw[dn][s2].w};
                    afn[s2][kk] = (W4L2_ABL & 1) ? af[s2][kk] : w4m_dequant_pk<T>(wn[kk], sc[s2], bi[s2]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int k = 0; k < 4; ++k) af[s2][k] = afn[s2][k];
            __builtin_amdgcn_s_waitcnt(0x0F70 | 4);  // vmcnt(4): everything but the weight tile just issued -- i.e. the DMA of tile g + 1 -- has landed
            if (!(W4L2_ABL & 2)) __syncthreads();     // ... everyone's has, and everyone is done reading buffer d & 1
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the clamped look-ahead loads of the last steps
    if (!SWIGLU && part) {  // K split: fp32 partial tiles straight from the accumulator layout (register i of lane l <-> column 32 nt + (i & 3) + 8 (i >> 2) + 4 kh, row 32 mi + (l & 31))
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (!(s2 ? has1 : has0)) continue;
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
                const int m = 32 * mi + n;
                if (m >= rows) continue;
                float *pr = part + (size_t)blockIdx.z * M * N + (size_t)(m0 + m) * N + 32 * (nt0 + s2) + 4 * kh;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(pr + 8 * q) = make_float4(acc[s2][mi][4 * q], acc[s2][mi][4 * q + 1], acc[s2][mi][4 * q + 2], acc[s2][mi][4 * q + 3]);
            }
        }
        return;
    }
    // everything else: rounded (SWIGLU: activated), through LDS, as whole lines (mfma_store.hpp; round 5: the 8-byte-per-row stores of the
    // accumulator layout cost 8-27 % of these kernels).  Idle strips (N not a multiple of 256) fall outside N there.
    __syncthreads();  // everyone has read its last B fragments
    mfma_tile_store<T, MB, 2, SWIGLU>(acc, &s_x[0][0] + wave * (MT * (SWIGLU ? 64 : 128)), lane, nt0, m0, rows, N, SWIGLU ? N >> 1 : N, y, nullptr);
}

// y = T(sum over z, in z order, of the fp32 partial tiles)
template <class T>
__global__ void __launch_bounds__(256) k_w4l_reduce(const float *part, int S, size_t MN, u16 *y) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    float4 a = *reinterpret_cast<const float4 *>(part + i);
    for (int z = 1; z < S; ++z) {
        const float4 b = *reinterpret_cast<const float4 *>(part + (size_t)z * MN + i);
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    *reinterpret_cast<uint2 *>(y + i) = make_uint2(w4m_pack<T>(a.x, a.y), w4m_pack<T>(a.z, a.w));
}


// K-split factor the launcher will use for [M, N, K]: enough workgroups for the chip, at least 8 groups (512 columns) per split.
// Decomposition: row tile (64 / 128 / 256 rows) x K split, and which kernel.  The one-wave-per-SIMD form (k_w4l2_gemm) needs a multiple
// of 4 K groups per split; where it applies, the plan takes the shallowest split that yields ~192 workgroups, and for that split the
// largest row tile -- a smaller row tile costs a few per cent of MFMA efficiency, a deeper split costs a whole fp32 reduce pass
// (at 1024 rows the K-split q|k|v / o_proj / down products spent 15 % of the prefill in it).  Other shapes keep the 8-wave form.
struct W4lPlan {
    int mt, S;
    bool v2;
};
static W4lPlan w4l_plan(int M, int N, int K) {
    const int groups = K >> 6, col_t = (N / 32 + 7) / 8;
    const int mt_max = M <= 64 ? 64 : (M <= 128 ? 128 : 256);
    if (groups % 4 == 0 && M > 32) {
        // a small cost model, calibrated on the 8B shapes at 512 / 1024 / 2048 rows (scripts/bench_w4l.py, swept with a developer override of the plan):
        // a K step costs ~1.30 / 0.85 / 0.75 us for 256 / 128 / 64-row tiles (below 256 rows the conversion of the weights, which does
        // not shrink with the tile, bounds the step), workgroups run in rounds of 256, a K split pays an fp32 write + read of S + 1
        // [M, N] slabs at ~1.3 TB/s.  The plan is the cheapest (row tile, split) under it.
        W4lPlan best = {mt_max, 1, true};
        double best_us = 1e30;
        for (int s = 1; s <= 16 && (s == 1 || 8 * s <= groups); ++s) {  // a split keeps at least 8 groups (512 columns)
            if (groups % (4 * s)) continue;
            for (int mt = mt_max; mt >= 64; mt >>= 1) {
                const double step_us = mt == 256 ? 1.30 : (mt == 128 ? 0.85 : 0.75);
                const int wgs = col_t * ((M + mt - 1) / mt) * s;
                double us = (double)((wgs + 255) / 256) * (groups / s) * step_us + 4.0;
                if (s > 1) us += (double)M * N * 4.0 * (s + 1) / 1.3e6 + 4.0;
                if (us < best_us) best_us = us, best = {mt, s, true};
            }
        }
        return best;
    }
    const int wgs = col_t * ((M + mt_max - 1) / mt_max);
    int s = wgs >= 192 ? 1 : (256 + wgs - 1) / wgs;
    const int max_s = groups / 8 > 0 ? groups / 8 : 1;
    s = s > max_s ? max_s : s;
    return {mt_max, s > 16 ? 16 : s, false};
}
int w4l_splits(int M, int N, int K) { return w4l_plan(M, N, K).S; }
size_t w4l_workspace_bytes(int M, int N, int K) {
    const int s = w4l_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

// workspace: w4l_workspace_bytes() of device scratch (may be null when that is 0)
// swiglu_act (nullable): gate|up matrix and the caller wants the MLP activation [M, N / 2] there instead of y; *fused says whether it got it
// slabs (nullable): the caller's consumer can sum the fp32 partial slabs of a K-split shape itself (in z order, then the Linear's one rounding:
//   what k_w4l_reduce does) -- *slabs = S and NO reduce launch then, workspace holds [S][M][N]; *slabs = 0 when y was written
int w4l_gemm_launch(int dtype, const void *w4m, const void *x, int M, int N, int K, void *y, void *workspace, hipStream_t st, void *swiglu_act, bool *fused,
                    int *slabs) {
    if (fused) *fused = false;
    if (slabs) *slabs = 0;
    PIE_REQUIRE(M >= 1 && N > 0 && K > 0 && N % 32 == 0 && K % 64 == 0, PIE_E_SHAPE, "W4 GEMM: N must be a multiple of 32 and K of 64");
    PIE_REQUIRE(pie_aligned(w4m, 16) && pie_aligned(x, 16) && pie_aligned(y, 8), PIE_E_ALIGN, "W4 GEMM: 16-byte alignment required");
    PIE_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16, PIE_E_ARG, "W4 GEMM: dtype must be PIE_BF16 or PIE_F16");
    const W4lPlan plan = w4l_plan(M, N, K);
    const int S = plan.S, mt = plan.mt;
    PIE_REQUIRE(S == 1 || workspace, PIE_E_ARG, "W4 GEMM: this shape splits K and needs its workspace");
    if (plan.v2) {
        float *part2 = S > 1 ? (float *)workspace : nullptr;
        const dim3 grid((unsigned)((N / 32 + 7) / 8), (unsigned)((M + mt - 1) / mt), (unsigned)S);
#define W4L2_GO(TT, MB_, SW_, Y_, P_) hipLaunchKernelGGL((k_w4l2_gemm<TT, MB_, SW_>), grid, dim3(256), 0, st, (const char *)w4m, (const u16 *)x, M, N, K, (u16 *)(Y_), P_)
#define W4L2_MB(TT, SW_, Y_, P_) \
    if (mt == 64) W4L2_GO(TT, 2, SW_, Y_, P_); \
    else if (mt == 128) W4L2_GO(TT, 4, SW_, Y_, P_); \
    else W4L2_GO(TT, 8, SW_, Y_, P_)
        if (S == 1 && swiglu_act && fused) {
            if (dtype == PIE_BF16) { W4L2_MB(BF16, true, swiglu_act, nullptr); }
            else { W4L2_MB(F16, true, swiglu_act, nullptr); }
            PIE_LAUNCH_CHECK();
            *fused = true;
            return PIE_OK;
        }
        if (dtype == PIE_BF16) { W4L2_MB(BF16, false, y, part2); }
        else { W4L2_MB(F16, false, y, part2); }
#undef W4L2_MB
#undef W4L2_GO
        PIE_LAUNCH_CHECK();
        if (S > 1 && slabs) {
            *slabs = S;
            return PIE_OK;
        }
        if (S > 1) {
            const size_t MN = (size_t)M * N;
            const dim3 rg((unsigned)((MN / 4 + 255) / 256));
            if (dtype == PIE_BF16) hipLaunchKernelGGL(k_w4l_reduce<BF16>, rg, dim3(256), 0, st, part2, S, MN, (u16 *)y);
            else hipLaunchKernelGGL(k_w4l_reduce<F16>, rg, dim3(256), 0, st, part2, S, MN, (u16 *)y);
            PIE_LAUNCH_CHECK();
        }
        return PIE_OK;
    }
    // blockIdx.x = column block (fastest): the workgroups that share an x tile are dispatched together and read it through every XCD's L2
    const dim3 grid((unsigned)((N / 32 + W4M_WAVES - 1) / W4M_WAVES), (unsigned)((M + mt - 1) / mt), (unsigned)S), block(W4M_WAVES * 64);
    float *part = S > 1 ? (float *)workspace : nullptr;
#define W4L_GO(TT, MB_) hipLaunchKernelGGL((k_w4l_gemm<TT, MB_>), grid, block, 0, st, (const char *)w4m, (const u16 *)x, M, N, K, (u16 *)y, part)
    if (dtype == PIE_BF16) {
        if (mt == 64) W4L_GO(BF16, 2);
        else if (mt == 128) W4L_GO(BF16, 4);
        else W4L_GO(BF16, 8);
    } else {
        if (mt == 64) W4L_GO(F16, 2);
        else if (mt == 128) W4L_GO(F16, 4);
        else W4L_GO(F16, 8);
    }
#undef W4L_GO
    PIE_LAUNCH_CHECK();
    if (S > 1 && slabs) {
        *slabs = S;
        return PIE_OK;
    }
    if (S > 1) {
        const size_t MN = (size_t)M * N;
        const dim3 rg((unsigned)((MN / 4 + 255) / 256));
        if (dtype == PIE_BF16) hipLaunchKernelGGL(k_w4l_reduce<BF16>, rg, dim3(256), 0, st, part, S, MN, (u16 *)y);
        else hipLaunchKernelGGL(k_w4l_reduce<F16>, rg, dim3(256), 0, st, part, S, MN, (u16 *)y);
        PIE_LAUNCH_CHECK();
    }
    return PIE_OK;
}

// ---------------------------------------------------------------- 16-bit weights, many rows (w16_gemm.hpp)
#include "w16_gemm.hpp"
int bias_any_launch(int dtype, void *y, const void *bias, int M, int N, hipStream_t st);  // vision.hip

static char *g_w16_zero = nullptr;  // 4 KiB of zeros: the weight tile of the ring's dummy steps
static std::mutex g_w16_mutex;
static int w16_zero_block(const char **out) {
    std::lock_guard<std::mutex> lock(g_w16_mutex);
    if (!g_w16_zero) {
        PIE_HIP_TRY(hipMalloc((void **)&g_w16_zero, W16M_TILE_BYTES));
        PIE_HIP_TRY(hipMemset(g_w16_zero, 0, W16M_TILE_BYTES));
    }
    *out = g_w16_zero;
    return PIE_OK;
}

size_t w16m_size(int N, int K) { return w16m_bytes(N, K); }
// row-major [N][K] (any N, K) -> W16M
int w16m_from_rows_launch(const void *w, int N, int K, void *w16m, hipStream_t st) {
    const char *zero;
    if (const int rc = w16_zero_block(&zero)) return rc;  // (first use of the format: never under stream capture)
    const int groups = (K + 63) / 64;
    const size_t pieces = w16m_bytes(N, K) / 16;
    hipLaunchKernelGGL(k_rows_to_w16m, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const u16 *)w, N, K, groups, pieces, (uint4 *)w16m);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}
// W16S units (the dense decode stream) -> W16M, packed row order
int w16m_from_w16s_launch(const void *w16s, int N, int K, void *w16m, hipStream_t st) {
    const char *zero;
    if (const int rc = w16_zero_block(&zero)) return rc;
    const int groups = (K + 63) / 64;
    const size_t pieces = w16m_bytes(N, K) / 16;
    hipLaunchKernelGGL(k_w16s_to_w16m, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const uint4 *)w16s, N, K, (K + 511) / 512, groups, pieces,
                       (uint4 *)w16m);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

// Decomposition: row tile (64 / 128 / 256 rows), strips per wave (workgroup = 128 or 256 columns) and K split, by a small cost model
// calibrated on tools/w16_bench sweep (8B and vision-tower shapes, 64 .. 4096 rows): a K step of a (32 mb) x (128 sw) tile costs
// t(mb, sw) us -- larger tiles are cheaper per MFMA (0.125 us per 32 x 32 block pair for 64 x 128, 0.09 for 256 x 256) but quantise worse --
// less when part of the chip idles (the large tiles are power-bound: the clock rises); workgroups run in rounds of 256; few rows
// are bound by the weight stream (~5.5 TB/s); a launch costs ~4 us of ramp, prologue and epilogue; a K split writes and re-reads its fp32
// slabs and launches the reduce.
struct W16Plan {
    int mb, sw, S;
};
static int g_w16_plan_override[3] = {0, 0, 0};  // developer override (tools/w16_bench)
static int g_w16_block[2] = {4, 8};              // XCD-local traversal block (row tiles x column tiles)
#ifdef W16L_PROF
static unsigned long long *g_w16_prof = nullptr;
#endif
static int w16_even_splits(int groups, int S) {  // the largest split count <= S whose ceil-sized runs are all non-empty
    for (;;) {
        const int per = (groups + S - 1) / S, s2 = (groups + per - 1) / per;
        if (s2 == S) return S;
        S = s2;
    }
}
static W16Plan w16_plan(int M, int N, int K, bool no_split = false) {
    const int groups = (K + 63) / 64;
    if (g_w16_plan_override[0]) return {g_w16_plan_override[0], g_w16_plan_override[1], w16_even_splits(groups, g_w16_plan_override[2])};
    const int mb_max = M <= 64 ? 2 : (M <= 128 ? 4 : 8);
    W16Plan best = {mb_max, 2, 1};
    double best_us = 1e30;
    // t(mb, sw) = light + (full - light) * load: one partial round of workgroups runs at a higher clock than sustained full rounds
    static const double t_light[4][3] = {{0, 0, 0}, {0, 0.25, 0.42}, {0, 0.31, 0.60}, {0, 0.50, 1.45}};  // [log2 mb][sw]
    static const double t_full[4][3] = {{0, 0, 0}, {0, 0.25, 0.45}, {0, 0.48, 0.80}, {0, 0.83, 1.45}};
    const double stream_us = 2.0 * N * (64.0 * groups) / 5.5e6;
    for (int S = 1; S <= (no_split ? 1 : 16) && (S == 1 || 8 * S <= groups); ++S)
        for (int mb = mb_max; mb >= 2; mb >>= 1)
            for (int sw = 2; sw >= (no_split ? 2 : 1); --sw) {
                const int wgs = ((N + 128 * sw - 1) / (128 * sw)) * ((M + 32 * mb - 1) / (32 * mb)) * S;
                const int steps = ((groups + S - 1) / S + 3) & ~3;
                const int rounds = (wgs + 255) / 256;
                const double load = rounds > 1 ? 1.0 : (double)wgs / 256.0;
                const int mi = mb == 8 ? 3 : (mb == 4 ? 2 : 1);
                double us = rounds * steps * (t_light[mi][sw] + (t_full[mi][sw] - t_light[mi][sw]) * load);
                us = (us > stream_us ? us : stream_us) + 4.0;
                if (S > 1) us += (double)M * N * 4.0 * (2 * S + 0.5) / 4.0e6 + 2.0;
                if (us < best_us) best_us = us, best = {mb, sw, S};
            }
    best.S = w16_even_splits(groups, best.S);
    return best;
}
int w16l_splits(int M, int N, int K) { return w16_plan(M, N, K).S; }
size_t w16l_workspace_bytes(int M, int N, int K) {
    if (N & 3) return 0;
    const int s = w16l_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

// y [M, N] = x . W^T (+ bias), W in W16M tiles of an [N, K] Linear; x: M rows of ldx elements (0 = packed) holding 64 ceil(K / 64) used
// columns, zeros past K.
// swiglu_act (nullable): the packed gate|up matrix and the caller wants the MLP activation [M, N / 2] there; *fused says whether it got it.
// workspace: w16l_workspace_bytes() of device scratch (may be null when that is 0).
// ldy: row stride of the output that is written (y, or the activation; 0 = packed), a multiple of 4.
int w16l_gemm_launch(int dtype, const void *w16m, const void *x, int ldx, int M, int N, int K, void *y, void *workspace, hipStream_t st, const void *bias,
                     void *swiglu_act, bool *fused, int ldy) {
    if (fused) *fused = false;
    const int Kx = 64 * ((K + 63) / 64);
    if (ldx == 0) ldx = Kx;
    PIE_REQUIRE(ldx >= Kx && ldx % 8 == 0 && (size_t)256 * ldx * 2 < (1ull << 32), PIE_E_SHAPE,
                "16-bit GEMM: x rows must hold 64 * ceil(K / 64) elements (zeros past K) at a stride that is a multiple of 8");
    PIE_REQUIRE(M >= 1 && N > 0 && K > 0, PIE_E_SHAPE, "16-bit GEMM: empty operand");
    PIE_REQUIRE(pie_aligned(w16m, 16) && pie_aligned(x, 16) && pie_aligned(y, 8) && (!bias || pie_aligned(bias, (N & 3) ? 2 : 8)), PIE_E_ALIGN, "16-bit GEMM: 16-byte alignment required");
    PIE_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16, PIE_E_ARG, "16-bit GEMM: dtype must be PIE_BF16 or PIE_F16");
    // swiglu_act without `fused`: the caller has no fallback (the C ABI's fused MLP): plan without a K split, two strips per wave
    W16Plan plan = w16_plan(M, N, K, (swiglu_act && !fused) || (N & 3));  // (the fp32 slabs of a K split are written 16 bytes at a time)
    const bool sg = swiglu_act && plan.S == 1;
    if (sg) plan.sw = 2;  // (the fused activation is built for two strips per wave)
    PIE_REQUIRE(plan.S == 1 || workspace, PIE_E_ARG, "16-bit GEMM: this shape splits K and needs its workspace");
    W16Args a = {};
    int rc = w16_zero_block(&a.zero);
    if (rc) return rc;
    PIE_REQUIRE(!sg || N % 8 == 0, PIE_E_SHAPE, "16-bit GEMM: a fused SwiGLU needs N % 8 == 0");
    const int out_cols = sg ? N / 2 : N;
    if (ldy == 0 || (swiglu_act && !sg)) ldy = out_cols;  // (a declined fusion writes the packed y)
    PIE_REQUIRE(ldy >= out_cols, PIE_E_SHAPE, "16-bit GEMM: output row stride must cover the row");
    PIE_REQUIRE(plan.S == 1 || ldy == N, PIE_E_SHAPE, "16-bit GEMM: a K-split shape writes packed rows");
    a.w16m = (const char *)w16m, a.x = (const u16 *)x, a.M = M, a.N = N, a.K = Kx, a.ldx = ldx, a.ldy = ldy;
    a.y = (u16 *)(sg ? swiglu_act : y), a.part = plan.S > 1 ? (float *)workspace : nullptr;
    a.bias = plan.S > 1 ? nullptr : (const u16 *)bias;
    a.tm = (M + 32 * plan.mb - 1) / (32 * plan.mb), a.tn = (N + 128 * plan.sw - 1) / (128 * plan.sw);
    a.bm = g_w16_block[0] < a.tm ? g_w16_block[0] : a.tm, a.bc = g_w16_block[1];
#ifdef W16L_PROF
    a.prof = g_w16_prof;
#endif
    const dim3 grid((unsigned)(8 * ((a.tn + 7) / 8) * a.tm), (unsigned)plan.S);
#define W16_GO(TT, MB_, SW_, SG_) hipLaunchKernelGGL((k_w16l_gemm<TT, MB_, SW_, SG_>), grid, dim3(256), 0, st, a)
#define W16_SW(TT, MB_) \
    if (sg) W16_GO(TT, MB_, 2, true); \
    else if (plan.sw == 2) W16_GO(TT, MB_, 2, false); \
    else W16_GO(TT, MB_, 1, false)
#define W16_MB(TT) \
    if (plan.mb == 2) { W16_SW(TT, 2); } \
    else if (plan.mb == 4) { W16_SW(TT, 4); } \
    else { W16_SW(TT, 8); }
    if (dtype == PIE_BF16) { W16_MB(BF16) } else { W16_MB(F16) }
#undef W16_MB
#undef W16_SW
#undef W16_GO
    PIE_LAUNCH_CHECK();
    if (sg && fused) *fused = true;
    if (plan.S > 1) {
        const size_t MN = (size_t)M * N;
        const dim3 rg((unsigned)((MN / 4 + 255) / 256));
        if (dtype == PIE_BF16) hipLaunchKernelGGL(k_w4l_reduce<BF16>, rg, dim3(256), 0, st, a.part, plan.S, MN, (u16 *)y);
        else hipLaunchKernelGGL(k_w4l_reduce<F16>, rg, dim3(256), 0, st, a.part, plan.S, MN, (u16 *)y);
        PIE_LAUNCH_CHECK();
        if (bias) return bias_any_launch(dtype, y, bias, M, N, st);
    }
    return PIE_OK;
}

// ---------------------------------------------------------------- 6 .. 256 rows: the weight-streaming form (w4r_gemm.hpp)
#include "w4r_gemm.hpp"

// Decomposition: 4 strips (128 columns) per workgroup, all rows; K split over blockIdx.y into runs of whole steps where the column
// workgroups alone cannot fill the chip (q|k|v: 48, o_proj / down: 32 on the 8B model) -- the splits' fp32 slabs are summed by the consumer.
// Geometry per row count (tools/w4r_bench on the 8B shapes; EXPERIMENTS.md): <= 64 rows two strips per wave, four K-phases, four x buffers
// of 16 / 32 KB; from 65 rows one strip per wave, two K-phases and as many x buffers of 24 .. 64 KB as LDS holds (4 / 3 / 2 up to 128 / 192 /
// 256 rows): the x chunk has to be in flight two steps ahead, and a 256-column chunk of 128 rows alone is 64 KB.
struct W4rPlan {
    int mb, kw, S, steps;  // row blocks, K-phases, K splits, steps per split; mb == 0: shape not served
};
static W4rPlan w4r_plan(int M, int N, int K, bool may_split) {
    W4rPlan pl = {0, 0, 1, 0};
    if (M < 1 || M > 256 || N < 32 || N % 32 || K < 64 || K % 64) return pl;
    int mb = (M + 31) / 32;
    if (mb == 7) mb = 8;
    const int kw = mb >= 3 ? 2 : 4;
    const int groups = K >> 6;
    if (groups < kw) return pl;
    const int total = (groups + kw - 1) / kw, col_wgs = (N / 32 + 3) / 4;
    int S = 1;
    if (may_split && col_wgs < 192) {
        S = 256 / col_wgs;
        // at least 1024 columns per split up to 64 rows (measured: o_proj at 32 rows 7.9 us with four splits, 9.3 with eight), 512 beyond (there the
        // matrix cores bound the launch and half a chip of workgroups costs more than short splits do)
        const int min_steps = mb <= 2 ? 16 / kw : 8 / kw;
        const int max_s = total / min_steps > 0 ? total / min_steps : 1;
        S = S > max_s ? max_s : S;
        S = S > 16 ? 16 : (S < 1 ? 1 : S);
        while (S > 1 && total % S) --S;  // equal splits: five uneven splits of q|k|v (7, 7, 7, 7, 4 steps) lose to four even ones (16.3 vs 15.0 us at 128 rows)
    }
    const int per = (total + S - 1) / S;
    S = (total + per - 1) / per;
    pl.mb = mb, pl.kw = kw, pl.S = S, pl.steps = per;
    return pl;
}
bool w4r_serves(int M, int N, int K) { return pie_knob(PIE_KNOB_W4R) != 0 && w4r_plan(M, N, K, false).mb != 0; }
int w4r_splits(int M, int N, int K) { return w4r_plan(M, N, K, true).S; }
size_t w4r_workspace_bytes(int M, int N, int K) {
    const int s = w4r_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

template <class T, bool PLAIN>
static void w4r_go(const W4rPlan &pl, int N, hipStream_t st, const W4rArgs &a, const W4mRope &r) {
    const dim3 grid((unsigned)((N / 32 + 3) / 4), (unsigned)pl.S), block(512);
    switch (pl.mb) {
    case 1: hipLaunchKernelGGL((k_w4r_gemm<T, 8, 1, 2, 4, 4, PLAIN>), grid, block, 0, st, a, r); break;
    case 2: hipLaunchKernelGGL((k_w4r_gemm<T, 8, 2, 2, 4, 4, PLAIN>), grid, block, 0, st, a, r); break;
    case 3: hipLaunchKernelGGL((k_w4r_gemm<T, 8, 3, 1, 2, 4, PLAIN>), grid, block, 0, st, a, r); break;
    case 4: hipLaunchKernelGGL((k_w4r_gemm<T, 8, 4, 1, 2, 4, PLAIN>), grid, block, 0, st, a, r); break;
    case 5: hipLaunchKernelGGL((k_w4r_gemm<T, 8, 5, 1, 2, 3, PLAIN>), grid, block, 0, st, a, r); break;
    case 6: hipLaunchKernelGGL((k_w4r_gemm<T, 8, 6, 1, 2, 3, PLAIN>), grid, block, 0, st, a, r); break;
    default: hipLaunchKernelGGL((k_w4r_gemm<T, 8, 8, 1, 2, 2, PLAIN>), grid, block, 0, st, a, r); break;
    }
}

// epi: W4R_STORE (y [M, N], + bias), W4R_SWIGLU (y = activation [M, N / 2], + bias), W4R_ROPE (rope != nullptr; q -> rope->q_out, k / v -> cache).
// slabs (nullable, STORE only): the caller's consumer sums the fp32 slabs of a K-split shape itself -> *slabs = S and workspace holds
// [S][M][N]; otherwise a K-split shape is reduced here (k_w4l_reduce) and the bias, if any, is left to the caller (returns *slabs = 0).
// wide_scales: some |scale| of the matrix is >= 2^100 (w4m_repack_launch reports it): outside w4r_dequant's domain, the plain conversion everywhere.
int w4r_gemm_launch(int dtype, const void *w4m, const void *x, int M, int N, int K, void *y, void *workspace, hipStream_t st, int epi, const void *bias,
                    const W4mRope *rope, int *slabs, bool *bias_done, bool wide_scales) {
    if (slabs) *slabs = 0;
    if (bias_done) *bias_done = false;
    PIE_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16, PIE_E_ARG, "W4R GEMM: dtype must be PIE_BF16 or PIE_F16");
    PIE_REQUIRE((epi == W4R_ROPE) == (rope != nullptr), PIE_E_ARG, "W4R GEMM: the q|k|v epilogue needs its arguments");
    const bool may_split = epi == W4R_STORE && workspace != nullptr;
    const W4rPlan pl = w4r_plan(M, N, K, may_split);
    PIE_REQUIRE(pl.mb != 0, PIE_E_SHAPE, "W4R GEMM: 1..256 rows, N a multiple of 32, K a multiple of 64 and at least 256 (128 beyond 64 rows)");
    PIE_REQUIRE(pie_aligned(w4m, 16) && pie_aligned(x, 16) && pie_aligned(y, 16) && pie_aligned(bias, 16), PIE_E_ALIGN, "W4R GEMM: 16-byte alignment required");
    PIE_REQUIRE(epi != W4R_SWIGLU || N % 64 == 0, PIE_E_SHAPE, "W4R GEMM: the SwiGLU epilogue needs whole column octets of pairs");
    W4rArgs a = {(const char *)w4m, (const u16 *)x, M, N, K, pl.steps, (u16 *)y, nullptr, (const u16 *)bias, epi};
#ifdef W4R_PROF
    extern unsigned long long *g_w4r_prof;
    a.prof = g_w4r_prof;
#endif
    const W4mRope r = rope ? *rope : W4mRope{};
    const int e = pl.S > 1 ? W4R_SLAB : epi;
    if (pl.S > 1) a.part = (float *)workspace, a.bias = nullptr;
    a.epi = e;
    if (dtype == PIE_F16) w4r_go<F16, false>(pl, N, st, a, r);  // f16 scales cannot leave w4r_dequant's domain
    else if (wide_scales) w4r_go<BF16, true>(pl, N, st, a, r);
    else w4r_go<BF16, false>(pl, N, st, a, r);
    PIE_LAUNCH_CHECK();
    if (pl.S > 1) {
        if (slabs) {
            *slabs = pl.S;
            return PIE_OK;
        }
        const size_t MN = (size_t)M * N;
        const dim3 rg((unsigned)((MN / 4 + 255) / 256));
        if (dtype == PIE_BF16) hipLaunchKernelGGL(k_w4l_reduce<BF16>, rg, dim3(256), 0, st, (const float *)workspace, pl.S, MN, (u16 *)y);
        else hipLaunchKernelGGL(k_w4l_reduce<F16>, rg, dim3(256), 0, st, (const float *)workspace, pl.S, MN, (u16 *)y);
        PIE_LAUNCH_CHECK();
        return PIE_OK;
    }
    if (bias_done) *bias_done = true;
    return PIE_OK;
}

size_t w4m_bytes(int N, int K) { return (size_t)(N >> 5) * (K >> 6) * W4M_TILE_BYTES; }

// Which W4M buffers hold a scale of magnitude >= 2^100 (k_w4r_gemm then converts with plain instructions): recorded per buffer when it is
// built -- one 4-byte read-back and a stream synchronisation per matrix, at load / first use -- and looked up by every launcher.
static std::mutex g_w4m_wide_mutex;
static std::map<const void *, bool> g_w4m_wide;
static int *g_w4m_wide_dev = nullptr;
bool w4m_wide_scales(const void *w4m) {
    std::lock_guard<std::mutex> lock(g_w4m_wide_mutex);
    auto it = g_w4m_wide.find(w4m);
    return it == g_w4m_wide.end() ? true : it->second;  // a buffer nobody checked: the conversion without a domain
}

int w4m_repack_launch(const void *w4s, int N, int K, void *w4m, hipStream_t st) {
    PIE_REQUIRE(N > 0 && K > 0 && N % 32 == 0 && K % 64 == 0, PIE_E_SHAPE, "W4M repack: N must be a multiple of 32 and K of 64");
    const size_t n = (size_t)(N >> 5) * (K >> 6) * 64;
    std::lock_guard<std::mutex> lock(g_w4m_wide_mutex);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    const bool check = cap == hipStreamCaptureStatusNone;  // no read-back inside a stream capture: the buffer then counts as wide
    if (check && !g_w4m_wide_dev) PIE_HIP_TRY(hipMalloc((void **)&g_w4m_wide_dev, sizeof(int)));
    if (check) PIE_HIP_TRY(hipMemsetAsync(g_w4m_wide_dev, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_w4s_to_w4m, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const u32 *)w4s, N, K, w4s_slices(K), (u32 *)w4m, check ? g_w4m_wide_dev : (int *)nullptr);
    PIE_LAUNCH_CHECK();
    int wide = 1;
    if (check) {
        PIE_HIP_TRY(hipMemcpyAsync(&wide, g_w4m_wide_dev, sizeof(int), hipMemcpyDeviceToHost, st));
        PIE_HIP_TRY(hipStreamSynchronize(st));
    }
    g_w4m_wide[w4m] = wide != 0;
    return PIE_OK;
}

// swiglu: N = 2 * inter interleaved gate|up rows -> y is the activation [M, N / 2] (bias: the Linear's, applied before it).
// swiglu == 2 (rope != nullptr): N = packed q|k|v rows; the epilogue rotates q / k and appends k / v to the cache (y unused).
int w4m_gemm_launch(int dtype, const void *w4m, const void *x, int M, int N, int K, void *y, hipStream_t st, int swiglu, const void *bias, const W4mRope *rope) {
    const W4mRope rope_args = rope ? *rope : W4mRope{};
    PIE_REQUIRE((swiglu == 2) == (rope != nullptr), PIE_E_ARG, "W4M GEMM: the q|k|v epilogue needs its arguments");
    PIE_REQUIRE(M >= 1 && M <= 32, PIE_E_SHAPE, "W4M GEMM: 1 to 32 rows");
    PIE_REQUIRE(N > 0 && K > 0 && N % 32 == 0 && K % 64 == 0, PIE_E_SHAPE, "W4M GEMM: N must be a multiple of 32 and K of 64");
    PIE_REQUIRE(pie_aligned(w4m, 16) && pie_aligned(x, 16), PIE_E_ALIGN, "W4M GEMM: 16-byte alignment required");
    PIE_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16, PIE_E_ARG, "W4M GEMM: dtype must be PIE_BF16 or PIE_F16");
    const dim3 grid(N >> 5), block(W4M_WAVES * 64);
    if (dtype == PIE_BF16) hipLaunchKernelGGL(k_w4m_gemm<BF16>, grid, block, 0, st, (const char *)w4m, (const u16 *)x, M, N, K, (u16 *)y, swiglu, (const u16 *)bias, rope_args);
    else hipLaunchKernelGGL(k_w4m_gemm<F16>, grid, block, 0, st, (const char *)w4m, (const u16 *)x, M, N, K, (u16 *)y, swiglu, (const u16 *)bias, rope_args);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}


extern "C" {

size_t pie_w4m_bytes(int N, int K) { return N > 0 && K > 0 && N % 32 == 0 && K % 64 == 0 ? w4m_bytes(N, K) : 0; }

int pie_repack_w4s_to_w4m(const void *w4s, int N, int K, void *w4m, void *stream) {
    PIE_REQUIRE(w4s && w4m, PIE_E_ARG, "pie_repack_w4s_to_w4m: null pointer");
    return w4m_repack_launch(w4s, N, K, w4m, (hipStream_t)stream);
}

int pie_qgemm_w4m(const void *x, const void *w4m, int M, int N, int K, int dtype, void *y, void *stream) {
    PIE_REQUIRE(x && w4m && y, PIE_E_ARG, "pie_qgemm_w4m: null pointer");
    // before any plan or workspace arithmetic: with N < 32 a plan has no column tiles and divides by zero on the host
    PIE_REQUIRE(M > 0 && N >= 32 && N % 32 == 0 && K >= 64 && K % 64 == 0, PIE_E_SHAPE, "pie_qgemm_w4m: M > 0, N a multiple of 32, K a multiple of 64");
    if (w4r_serves(M, N, K)) {  // 6 .. 256 rows (and fewer, when asked at this level): the weight-streaming form
        hipStream_t st = (hipStream_t)stream;
        void *ws = nullptr;
        const size_t wb = w4r_workspace_bytes(M, N, K);
        if (wb && hipMallocAsync(&ws, wb, st) != hipSuccess) return pie::fail(PIE_E_HIP, "pie_qgemm_w4m: hipMallocAsync failed");
        const int rc = w4r_gemm_launch(dtype, w4m, x, M, N, K, y, ws, st, W4R_STORE, nullptr, nullptr, nullptr, nullptr, w4m_wide_scales(w4m));
        if (ws) (void)hipFreeAsync(ws, st);
        return rc;
    }
    if (M > 32) {  // the prompt GEMM; a K-split shape takes stream-ordered scratch for its fp32 partial tiles
        hipStream_t st = (hipStream_t)stream;
        void *ws = nullptr;
        const size_t wb = w4l_workspace_bytes(M, N, K);
        if (wb && hipMallocAsync(&ws, wb, st) != hipSuccess) return pie::fail(PIE_E_HIP, "pie_qgemm_w4m: hipMallocAsync failed");
        const int rc = w4l_gemm_launch(dtype, w4m, x, M, N, K, y, ws, st, nullptr, nullptr, nullptr);
        if (ws) (void)hipFreeAsync(ws, st);
        return rc;
    }
    return w4m_gemm_launch(dtype, w4m, x, M, N, K, y, (hipStream_t)stream, 0, nullptr, nullptr);
}

}  // extern "C"
