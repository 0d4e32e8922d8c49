// prefill.hip -- batched prompt processing: Model.__call__(inputs[1, L], cache) for L > 1
// (models/llama/language.py:199-210 with the causal mask of models/base.py:18-53), SURVEY.md 8 row f1.
//
// At L > 1 the reference's nn.QuantizedLinear reaches MLX's matrix kernels (qmm), which dequantise the int4 block to T
// and feed a T x T -> fp32 MMA; the vector kernels (qmv, exact fp32 affine math) serve only the few-row case.  This file
// follows that split: prompts of prefill_min_rows() tokens or more run here, shorter ones as iterated decode steps.
//
// MI355X design, per weight format (no library GEMM anywhere: the hipBLASLt path of rounds 1-4 was removed in round 5):
//   * int4 g=64 modules multiply their 4-bit W4M tiles directly (w4m_gemm.hip: k_w4r_gemm up to 256 rows, k_w4l2_gemm beyond);
//   * dense modules, and int8 / group-32 modules, multiply a 16-bit copy in W16M tiles (w16_gemm.hpp: k_w16l_gemm, the weights
//     straight into MFMA fragment registers, x through LDS-DMA) -- dense ones tiled from their W16S units, quantised ones from
//     k_dequant_w4s / k_dequant_w8s (the mx.dequantize arithmetic: T(fp32(s*q) + b)); 288 GB of HBM keep the copies resident;
//   * hand-written HIP for everything around it: RMSNorm rows, RoPE + cache append from the packed q|k|v rows, causal
//     attention (MFMA flash kernel), the split merge, SwiGLU on the interleaved gate/up rows, residual adds.
// The chunk is bounded (PIE_KNOB_PREFILL_CHUNK, default 4096 rows: ~0.6 GB of activation scratch on the 8B model).
#include <cstdlib>
#include <map>
#include <mutex>

#include "decoder.hpp"
#include "prefill_attn.hpp"

int embedding_launch(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H, int dtype,
                     void *out, const float *freqs, const DecState *state, float *rope_cs, int half, hipStream_t st, int bits);

// ---------------------------------------------------------------- kernels
// W4S -> T row-major [N, K]; one thread per code word (8 weights, 16 B out).  Same arithmetic as k_dequantize_w4g64.
template <class T, bool G32 = false>  // G32: W4S32 units (two {scale | bias} words per lane: code piece j is its own 32-wide group)
__global__ void k_dequant_w4s(const u32 *packed, int N, int K, int ns, u16 *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wpr = K >> 3;  // words per row
    if (idx >= (size_t)N * wpr) return;
    const int r = (int)(idx / wpr), wk = (int)(idx % wpr);
    const int g = wk >> 3, lane = (r & 1) * 32 + (g & 31), j = (wk & 7) >> 2, t = wk & 3;
    const u32 *unit = packed + ((size_t)(r >> 1) * ns + (g >> 5)) * ((G32 ? W4S32_UNIT_BYTES : W4S_UNIT_BYTES) / 4);
    const u32 word = unit[j * 256 + lane * 4 + t], sb = G32 ? unit[512 + 2 * lane + j] : unit[512 + lane];
    const float s = lo_f32<T>(sb), b = hi_f32<T>(sb);
    u32 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // codes (2i, 2i+1) sit in the low / high half at nibble i (the W4S nibble order)
        const float lo = __fadd_rn(__fmul_rn(s, (float)((word >> (4 * i)) & 0xFu)), b);
        const float hi = __fadd_rn(__fmul_rn(s, (float)((word >> (16 + 4 * i)) & 0xFu)), b);
        o[i] = pack2<T>(lo, hi);
    }
    *reinterpret_cast<uint4 *>(out + (size_t)r * K + (size_t)wk * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

// W2S -> T row-major [N, K]; one thread per code word (16 weights, 32 B out).
template <class T>
__global__ void k_dequant_w2s(const u32 *packed, int N, int K, int ns, u16 *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wpr = K >> 4;  // words per row
    if (idx >= (size_t)N * wpr) return;
    const int r = (int)(idx / wpr), wk = (int)(idx % wpr);
    const int g = wk >> 2, lane = (r & 1) * 32 + (g & 31), t = wk & 3;
    const u32 *unit = packed + ((size_t)(r >> 1) * ns + (g >> 5)) * (W2S_UNIT_BYTES / 4);
    const u32 word = unit[lane * 4 + t], sb = unit[256 + lane];
    const float s = lo_f32<T>(sb), b = hi_f32<T>(sb);
    u32 o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // codes (2j, 2j+1) of the word sit at bits 2j of the low / high half (the W2S order)
        const float lo = __fadd_rn(__fmul_rn(s, (float)((word >> (2 * j)) & 0x3u)), b);
        const float hi = __fadd_rn(__fmul_rn(s, (float)((word >> (16 + 2 * j)) & 0x3u)), b);
        o[j] = pack2<T>(lo, hi);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)r * K + (size_t)wk * 16);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]), dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// W6S -> T row-major [N, K]; one thread per low-plane code word (8 weights, 16 B out): q = low nibble + 16 x the high plane's two bits.
template <class T>
__global__ void k_dequant_w6s(const u32 *packed, int N, int K, int ns, u16 *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wpr = K >> 3;
    if (idx >= (size_t)N * wpr) return;
    const int r = (int)(idx / wpr), wk = (int)(idx % wpr);
    const int g = wk >> 3, lane = (r & 1) * 32 + (g & 31), w4 = wk & 7, j = w4 >> 2, t = w4 & 3;
    const u32 *unit = packed + ((size_t)(r >> 1) * ns + (g >> 5)) * (W6S_UNIT_BYTES / 4);
    const u32 word = unit[j * 256 + lane * 4 + t], sb = unit[768 + lane];
    const u32 hiw = unit[512 + lane * 4 + (w4 >> 1)] >> (8 * (w4 & 1));  // codes 8 w4 .. + 7 = pairs 4 (w4 & 1) .. + 3 of high word w4 >> 1
    const float s = lo_f32<T>(sb), b = hi_f32<T>(sb);
    u32 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 qa = ((word >> (4 * i)) & 0xFu) | (((hiw >> (2 * i)) & 0x3u) << 4), qb = ((word >> (16 + 4 * i)) & 0xFu) | (((hiw >> (16 + 2 * i)) & 0x3u) << 4);
        o[i] = pack2<T>(__fadd_rn(__fmul_rn(s, (float)qa), b), __fadd_rn(__fmul_rn(s, (float)qb), b));
    }
    *reinterpret_cast<uint4 *>(out + (size_t)r * K + (size_t)wk * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

// W8S -> T row-major [N, K]; one thread per code word (4 weights, 8 B out).
template <class T, bool G32 = false>  // G32: W8S32 units (code pieces 0-1 / 2-3 are two 32-wide groups)
__global__ void k_dequant_w8s(const u32 *packed, int N, int K, int ns, u16 *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wpr = K >> 2;
    if (idx >= (size_t)N * wpr) return;
    const int r = (int)(idx / wpr), wk = (int)(idx % wpr);
    const int g = wk >> 4, lane = (r & 1) * 32 + (g & 31), j = (wk & 15) >> 2, t = wk & 3;
    const u32 *unit = packed + ((size_t)(r >> 1) * ns + (g >> 5)) * ((G32 ? W8S32_UNIT_BYTES : W8S_UNIT_BYTES) / 4);
    const u32 word = unit[j * 256 + lane * 4 + t], sb = G32 ? unit[1024 + 2 * lane + (j >> 1)] : unit[1024 + lane];
    const float s = lo_f32<T>(sb), b = hi_f32<T>(sb);
    // stored byte order (c0, c2, c1, c3)
    const float q0 = __fadd_rn(__fmul_rn(s, (float)(word & 0xFFu)), b), q2 = __fadd_rn(__fmul_rn(s, (float)((word >> 8) & 0xFFu)), b);
    const float q1 = __fadd_rn(__fmul_rn(s, (float)((word >> 16) & 0xFFu)), b), q3 = __fadd_rn(__fmul_rn(s, (float)(word >> 24)), b);
    *reinterpret_cast<uint2 *>(out + (size_t)r * K + (size_t)wk * 4) = make_uint2(pack2<T>(q0, q1), pack2<T>(q2, q3));
}

// (cos, sin) of every row's rotary angles, once per forward: the angle of pair ii at position p is the same for all heads and all
// layers, and sincosf on arguments up to ~1e5 rad (argument reduction) was most of k_rope_append_rows' 8 us per layer.
__global__ void k_rope_cs_rows(const float *freqs, const DecState *state, const int *ctx_len, int half, float *cs) {
    const int m = blockIdx.x;
    int pos = ctx_len ? ctx_len[m] - 1 : state->pos + m;
    pos = pos < 0 ? 0 : pos;
    for (int ii = threadIdx.x; ii < half; ii += blockDim.x) {
        float sn, c;
        sincosf((float)pos * (1.0f / freqs[ii]), &sn, &c);
        cs[((size_t)m * half + ii) * 2] = c, cs[((size_t)m * half + ii) * 2 + 1] = sn;
    }
}

// A K-split many-row GEMM (w4m_gemm.hip) leaves S fp32 partial slabs [S][M][N]; its consumer can form the Linear's output itself --
// the slabs summed in slab order, then the one rounding to T: exactly k_w4l_reduce's arithmetic -- which saves that launch and the
// round trip of y through memory (prompts of 33..~700 rows split K; a launch is ~5 us of a 150-300 us layer there).
struct W4lSlabs {
    const float *part = nullptr;
    int S = 0;
    size_t MN = 0;
};
#ifndef SLAB_BATCH
#define SLAB_BATCH 4
#endif
// V = float2 / float4: one thread's NV vectors at element offsets off[]; the slabs are fetched BATCH at a time (all BATCH x NV loads in flight
// before the first add -- a plain loop over the slabs waits out one memory latency per slab: 15 us instead of 5 for a 64-row o_proj
// consumer) and added strictly in slab order.
template <class V, int NV, int BATCH = 4>
__device__ __forceinline__ void slab_sum(const float *part, int S, size_t MN, const size_t (&off)[NV], V (&acc)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = *reinterpret_cast<const V *>(part + off[v]);
    for (int z = 1; z < S; z += BATCH) {
        V b[BATCH][NV];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const size_t zz = (size_t)(z + u < S ? z + u : S - 1) * MN;  // clamp, never branch around a load
#pragma unroll
            for (int v = 0; v < NV; ++v) b[u][v] = *reinterpret_cast<const V *>(part + zz + off[v]);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u)
            if (z + u < S) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    acc[v].x += b[u][v].x, acc[v].y += b[u][v].y;
                    if constexpr (sizeof(V) == 16) acc[v].z += b[u][v].z, acc[v].w += b[u][v].w;
                }
            }
    }
}

// RoPE (llama/utils.py:42-50, offset = cache.offset) + cache append (reusable.py:134-137) for M rows of the packed
// q|k|v projection: packed columns (2i, 2i+1) of a q/k head are its dims (i, i + D/2); v columns are natural.
// grid M, one thread per packed column pair.
// SLAB / I8: the two optional input / output forms as compile-time switches -- carried as run-time branches they cost the plain
// 4096-row launch 7.6 us (34.4 vs 26.8 us per layer)
template <class T, bool SLAB = false, bool I8 = false>
__global__ void __launch_bounds__(256) k_rope_append_rows(const u16 *qkv, int n_cols, const float *freqs, const DecState *state,
                                                         const unsigned long long *kv_table, int layer, int n_layers, int n_heads,
                                                         int n_kv_heads, int HD, int traditional, u16 *q_out, const int *block_table,
                                                         int n_pages, const float *rope_cs, const int *ctx_len = nullptr, int bt_stride = 0,
                                                         u16 *slab = nullptr, const int *row_seq = nullptr, u16 *kc = nullptr, u16 *vc = nullptr,
                                                         const float *part = nullptr, int S = 0, size_t MN = 0, size_t i8_page_bytes = 0) {
    // part != nullptr: the projection arrives as the fp32 slabs of a K-split GEMM (W4lSlabs) instead of qkv
    // i8_page_bytes != 0 (with `slab`): int8 pages with per-head fp16 scales (paged_i8.hip): the rotated K / the V element x, already
    // rounded to T, is stored as clamp(rint(x / s), -127, 127) with the scale of (its page, its kv-head)
    // row_seq / kc / vc (several prompts in one pass): row m belongs to sequence row_seq[m] (its block-table row), and the
    // rotated K and the V rows are also written to contiguous [n_kv_heads, rows, HD] buffers the attention of this pass reads
    // ctx_len != nullptr (multi-sequence decode step): row m is its own sequence at position ctx_len[m] - 1 with its own block
    // table row and the layer's slab given directly; otherwise row m continues the decoder's one sequence at state->pos + m
    const int m = blockIdx.x, pos = ctx_len ? ctx_len[m] - 1 : state->pos + m, half = HD >> 1;
    if (pos < 0) return;  // idle slot of the batch
    int cap = ctx_len ? 64 : state->cap, kvrow = pos;
    const int q_cols = n_heads * HD, k_cols = n_kv_heads * HD;
    u16 *kdst = slab ? slab : reinterpret_cast<u16 *>(kv_table[layer]);
    u16 *vdst = slab ? slab + (size_t)n_kv_heads * 64 * HD : reinterpret_cast<u16 *>(kv_table[n_layers + layer]);
    block_table = block_table ? block_table + (size_t)(row_seq ? row_seq[m] : m) * bt_stride : nullptr;
    const int n_rows = gridDim.x;
    char *page8 = nullptr;  // int8 pages: this row's page
    if (block_table) {  // paged KV: the row goes to slot pos % 64 of page block_table[pos / 64]
        const unsigned pg = min((unsigned)block_table[pos >> 6], (unsigned)n_pages - 1u);
        const size_t pg_off = (size_t)pg * 2 * 64 * n_kv_heads * HD;
        kdst += pg_off, vdst += pg_off, cap = 64, kvrow = pos & 63;
        if (I8 && i8_page_bytes && slab) page8 = reinterpret_cast<char *>(slab) + (size_t)pg * i8_page_bytes;
    }
    const size_t blk8 = (size_t)n_kv_heads * 64 * HD;
    auto q8 = [](float x, float sc) {
        float q = rintf(x / sc);
        q = q < -127.0f ? -127.0f : (q > 127.0f ? 127.0f : q);
        return (signed char)(q == q ? (int)q : 0);
    };
    auto f16f = [](u16 h) { return (float)__builtin_bit_cast(_Float16, h); };
    const u16 *row = qkv + (size_t)m * n_cols;
    // gridDim.y workgroups share a row (few rows summing fp32 slabs: one workgroup per row cannot keep enough loads in flight)
    for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < (n_cols >> 1); p += gridDim.y * blockDim.x) {
        const int R = 2 * p;
        u32 pr;
        if (SLAB && part) {
            const size_t i[1] = {(size_t)m * n_cols + R};
            float2 a[1];
            slab_sum<float2, 1>(part, S, MN, i, a);
            pr = pack2<T>(a[0].x, a[0].y);
        } else pr = *reinterpret_cast<const u32 *>(row + R);
        const float ra = lo_f32<T>(pr), rb = hi_f32<T>(pr);
        if (R < q_cols + k_cols) {
            const int rr = R < q_cols ? R : R - q_cols;
            const int head = rr / HD, ii = (rr % HD) >> 1;
            const float2 csn = *reinterpret_cast<const float2 *>(rope_cs + ((size_t)m * half + ii) * 2);  // k_rope_cs_rows
            const float cs = csn.x, sn = csn.y;
            u16 *dst = R < q_cols ? q_out + ((size_t)m * n_heads + head) * HD : kdst + ((size_t)head * cap + kvrow) * HD;
            const int i0 = traditional ? 2 * ii : ii, i1 = traditional ? 2 * ii + 1 : ii + half;
            const u16 o0 = T::from_f32(__fsub_rn(__fmul_rn(ra, cs), __fmul_rn(rb, sn))), o1 = T::from_f32(__fadd_rn(__fmul_rn(ra, sn), __fmul_rn(rb, cs)));
            if (I8 && page8 && R >= q_cols) {
                const float sk = f16f(reinterpret_cast<const u16 *>(page8 + 2 * blk8)[head]);
                signed char *kb = reinterpret_cast<signed char *>(page8) + ((size_t)head * 64 + kvrow) * HD;
                kb[i0] = q8(T::to_f32(o0), sk), kb[i1] = q8(T::to_f32(o1), sk);
            } else dst[i0] = o0, dst[i1] = o1;
            if (kc && R >= q_cols) {
                u16 *c = kc + ((size_t)head * n_rows + m) * HD;
                c[i0] = o0, c[i1] = o1;
            }
        } else {
            const int rr = R - q_cols - k_cols;
            if (I8 && page8) {
                const int head = rr / HD;
                const float sv = f16f(reinterpret_cast<const u16 *>(page8 + 2 * blk8)[n_kv_heads + head]);
                signed char *vb = reinterpret_cast<signed char *>(page8) + blk8 + ((size_t)head * 64 + kvrow) * HD + rr % HD;
                vb[0] = q8(ra, sv), vb[1] = q8(rb, sv);
            } else *reinterpret_cast<u32 *>(vdst + ((size_t)(rr / HD) * cap + kvrow) * HD + rr % HD) = pr;
            if (vc) *reinterpret_cast<u32 *>(vc + ((size_t)(rr / HD) * n_rows + m) * HD + rr % HD) = pr;
        }
    }
}

// nn.silu(gate) * up (language.py:127) on interleaved columns (2i, 2i+1) = (gate_i, up_i); 4 outputs per thread.
template <class T>
__global__ void k_swiglu_rows(const u16 *gu, size_t n_out, u16 *act) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n_out) return;  // n_out is a multiple of 4 (inter % 64 == 0)
    const uint4 v = *reinterpret_cast<const uint4 *>(gu + 2 * i);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    u16 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float g = lo_f32<T>(w[k]), u = hi_f32<T>(w[k]);
        o[k] = T::from_f32(round_T<T>(g / (1.0f + expf(-g))) * u);
    }
    *reinterpret_cast<uint2 *>(act + i) = make_uint2((u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16));
}

// y[m, :] = T(y[m, :] + b) for the Linear biases (y already T-rounded by the GEMM); 8 columns per thread, N % 8 == 0.
template <class T>
__global__ void k_bias_rows(u16 *y, const u16 *b, size_t n8, int N8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const uint4 v = reinterpret_cast<const uint4 *>(y)[i], bb = reinterpret_cast<const uint4 *>(b)[i % N8];
    const u32 a[4] = {v.x, v.y, v.z, v.w}, c[4] = {bb.x, bb.y, bb.z, bb.w};
    u32 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2<T>(lo_f32<T>(a[k]) + lo_f32<T>(c[k]), hi_f32<T>(a[k]) + hi_f32<T>(c[k]));
    reinterpret_cast<uint4 *>(y)[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

// h = x + r (language.py:151,153) fused with the RMSNorm that follows it (language.py:137-141): x <- T(x + r) in place,
// xn <- w * T(h * rsqrt(mean(h^2) + eps)).  One workgroup per row; the row stays in registers between the two passes.
template <class T>
__global__ void __launch_bounds__(256) k_add_rms_norm_rows(u16 *x, const u16 *r, const u16 *w, float eps, int H, u16 *xn, const u16 *bias = nullptr,
                                                          const float *part = nullptr, int S = 0, size_t MN = 0) {
    // part != nullptr: the Linear's output arrives as the fp32 slabs of a K-split many-row GEMM (W4lSlabs): r = T(T(sum of the slabs) + bias)
    __shared__ float red[4];
    u16 *xr = x + (size_t)blockIdx.x * H;
    const u16 *rr = r + (size_t)blockIdx.x * H;
    u16 *yr = xn + (size_t)blockIdx.x * H;
    constexpr int MAXP = 4;  // 8-element pieces per thread: H <= 8192
    uint4 hv[MAXP];
    float ssq = 0.0f;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int i = (threadIdx.x + p * 256) * 8;
        if (i < H) {
            const uint4 a = *reinterpret_cast<const uint4 *>(xr + i);
            uint4 b;
            if (part) {
                float4 f0, f1;
                {
                    const size_t o[2] = {(size_t)blockIdx.x * H + i, (size_t)blockIdx.x * H + i + 4};
                    float4 f[2];
                    slab_sum<float4, 2, SLAB_BATCH>(part, S, MN, o, f);
                    f0 = f[0], f1 = f[1];
                }
                float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                if (bias) {
                    const uint4 bb = *reinterpret_cast<const uint4 *>(bias + i);
                    const u32 bw[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[2 * j] = round_T<T>(v[2 * j]) + lo_f32<T>(bw[j]), v[2 * j + 1] = round_T<T>(v[2 * j + 1]) + hi_f32<T>(bw[j]);
                }
                b = make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
            } else {
                b = *reinterpret_cast<const uint4 *>(rr + i);
            }
            const u32 av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
            u32 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = pack2<T>(lo_f32<T>(av[j]) + lo_f32<T>(bv[j]), hi_f32<T>(av[j]) + hi_f32<T>(bv[j]));
                const float lo = lo_f32<T>(o[j]), hi = hi_f32<T>(o[j]);
                ssq = fmaf(lo, lo, ssq);
                ssq = fmaf(hi, hi, ssq);
            }
            hv[p] = make_uint4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<uint4 *>(xr + i) = hv[p];
        }
    }
    ssq = wave_sum(ssq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ssq;
    __syncthreads();
    const float inv = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int i = (threadIdx.x + p * 256) * 8;
        if (i < H) {
            const uint4 g = *reinterpret_cast<const uint4 *>(w + i);
            const u32 vv[4] = {hv[p].x, hv[p].y, hv[p].z, hv[p].w}, gg[4] = {g.x, g.y, g.z, g.w};
            u32 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack2<T>(round_T<T>(lo_f32<T>(vv[j]) * inv) * lo_f32<T>(gg[j]), round_T<T>(hi_f32<T>(vv[j]) * inv) * hi_f32<T>(gg[j]));
            *reinterpret_cast<uint4 *>(yr + i) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <class T>
static int add_rms_norm_rows(u16 *x, const u16 *r, const void *w, float eps, int M, int H, int dtype, u16 *xn, hipStream_t st,
                             const void *bias = nullptr, const W4lSlabs *sl = nullptr) {
    if (sl && sl->S > 1) {  // (linear_rows only hands out slabs for H <= 8192)
        hipLaunchKernelGGL(k_add_rms_norm_rows<T>, dim3(M), dim3(256), 0, st, x, r, (const u16 *)w, eps, H, xn, (const u16 *)bias, sl->part, sl->S,
                           sl->MN);
        PIE_LAUNCH_CHECK();
        return PIE_OK;
    }
    if (H > 8192) {  // wider than the register-resident row: the two separate kernels
        const int rc = pie_add(x, r, (size_t)M * H, dtype, x, st);
        return rc ? rc : pie_rms_norm(x, w, eps, M, H, dtype, xn, st);
    }
    hipLaunchKernelGGL(k_add_rms_norm_rows<T>, dim3(M), dim3(256), 0, st, x, r, (const u16 *)w, eps, H, xn, (const u16 *)bias);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

__global__ void k_add_pos(DecState *s, int delta) { s->pos += delta; }

// ---------------------------------------------------------------- scratch
struct PrefillScratch {
    int rows = 0;           // chunk capacity (rows)
    size_t w_elems = 0;     // capacity of the dequantised-weight buffer (elements)
    u16 *wT = nullptr, *wM = nullptr, *x = nullptr, *xn = nullptr, *qkv = nullptr, *q = nullptr, *attn = nullptr, *gu = nullptr, *act = nullptr, *r = nullptr;
    float *part_acc = nullptr, *part_ml = nullptr, *rope_cs = nullptr;
    u16 *kc = nullptr, *vc = nullptr;  // [n_kv_heads, rows, head_dim]: this pass's K / V rows, contiguous (several prompts in one pass)
    int part_splits = 0;
    // Resident T copies of the layer matrices, keyed by the packed-weight pointer: the per-chunk dequantisation moves
    // 4.6 B per parameter (7 ms of a 10.8 ms 128-token prefill on the 8B model) for 2 B per parameter of HBM; kept when
    // that is a small share of the free memory (PIE_KNOB_PREFILL_RESIDENT overrides), built on first use.
    std::map<const void *, u16 *> resident;
    // W4M tile copies (w4m_gemm.hip) of int4 layer matrices for prompts of at most small_rows() rows: 0.5625 B per weight each
    std::map<const void *, void *> resident_w4m;
    LogitStat *tail_stats = nullptr;  // multi-sequence step / several prompts per pass: per-row tail partials
    size_t tail_entries = 0;           // its capacity in ENTRIES (rows x partials per row: the two users need 256 and max(lm_head waves, 256) per row)
    // Captured multi-sequence step (pie_decoder_step_batch with PIE_STEP_GRAPH): valid while the caller's buffers (key) and every
    // device allocation baked into the launches (alloc_gen) stay what they were at capture time.
    unsigned alloc_gen = 0;
    hipGraphExec_t batch_graph = nullptr;
    std::vector<uintptr_t> batch_key, warm_key;
    unsigned batch_gen = 0, warm_gen = 0;
    void *w4l_ws = nullptr;   // fp32 partial tiles of the K-split prompt GEMM (w4m_gemm.hip)
    size_t w4l_ws_bytes = 0;
    int resident_mode = -1;   // -1: budget not fixed yet
    size_t resident_left = 0; // bytes still available for resident copies
};

// The tail partials are sized in entries, not rows: a prompt pass needs 256 per row, the fused few-sequence step one per lm_head GEMV
// wave (2048 at V = 128256) -- a buffer sized by the first and reused by the second was overrun 8 x.
static int tail_stats_reserve(PrefillScratch *s, size_t entries) {
    if (s->tail_entries >= entries) return PIE_OK;
    if (s->tail_stats) (void)hipFree(s->tail_stats);
    s->tail_stats = nullptr, s->tail_entries = 0;
    PIE_HIP_TRY(hipMalloc((void **)&s->tail_stats, sizeof(LogitStat) * entries));
    s->tail_entries = entries, ++s->alloc_gen;
    return PIE_OK;
}

static void scratch_release(PrefillScratch *s) {  // the chunk buffers; resident weight copies survive a re-size
    void *ptrs[] = {s->wT, s->wM, s->x, s->xn, s->qkv, s->q, s->attn, s->gu, s->act, s->r, s->part_acc, s->part_ml, s->rope_cs, s->kc, s->vc};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    auto keep = std::move(s->resident);
    auto keep_m = std::move(s->resident_w4m);
    LogitStat *keep_ts = s->tail_stats;
    const size_t keep_tr = s->tail_entries;
    const unsigned gen = s->alloc_gen + 1;  // the chunk buffers move: captured launches are stale
    if (s->batch_graph) (void)hipGraphExecDestroy(s->batch_graph);
    const int mode = s->resident_mode;
    const size_t left = s->resident_left;
    void *keep_ws = s->w4l_ws;
    const size_t keep_wsb = s->w4l_ws_bytes;
    *s = PrefillScratch();
    s->w4l_ws = keep_ws, s->w4l_ws_bytes = keep_wsb;
    s->resident = std::move(keep), s->resident_w4m = std::move(keep_m), s->resident_mode = mode, s->resident_left = left;
    s->tail_stats = keep_ts, s->tail_entries = keep_tr, s->alloc_gen = gen;
}

void prefill_free(pie_decoder *d) {
    if (!d->prefill) return;
    scratch_release(d->prefill);
    for (auto &kv : d->prefill->resident) (void)hipFree(kv.second);
    for (auto &kv : d->prefill->resident_w4m) (void)hipFree(kv.second);
    if (d->prefill->tail_stats) (void)hipFree(d->prefill->tail_stats);
    if (d->prefill->w4l_ws) (void)hipFree(d->prefill->w4l_ws);
    if (d->prefill->batch_graph) (void)hipGraphExecDestroy(d->prefill->batch_graph);
    delete d->prefill;
    d->prefill = nullptr;
}

int prefill_min_rows() {  // read per call (not cached): tests and tools switch regimes inside one process
    const int k = pie_knob(PIE_KNOB_PREFILL_MIN);  // prompts shorter than this run as iterated decode steps (MLX's qmv regime)
    const int n = k >= 0 ? k : 6;  // MLX's own qmv limit is 6..32 rows by device and shape; 5 ms batched vs 1.35 ms per iterated row
    return n < 2 ? 2 : n;
}

static int prefill_chunk_rows() {
    const int k = pie_knob(PIE_KNOB_PREFILL_CHUNK);  // measured on the 8B model, 4096-token prompt: 512 -> 119 ms, 1024 -> 83, 2048 -> 67, 4096 -> 65
    const int n = k >= 0 ? k : 4096;
    return n < 16 ? 16 : (n > 8192 ? 8192 : n);
}

static int scratch_reserve(pie_decoder *d, int rows, size_t w_elems, int splits) {
    if (!d->prefill) d->prefill = new PrefillScratch();
    PrefillScratch *s = d->prefill;
    const pie_decoder_config &c = d->cfg;
    const size_t QD = (size_t)c.n_heads * c.head_dim, NQKV = QD + 2 * (size_t)c.n_kv_heads * c.head_dim;
    if (rows > s->rows || splits > s->part_splits) {
        const size_t we = s->w_elems > w_elems ? s->w_elems : w_elems;
        scratch_release(s);
        const size_t R = (size_t)rows;
#define PF_ALLOC(ptr, bytes) PIE_HIP_TRY(hipMalloc((void **)&(ptr), (bytes)))
        PF_ALLOC(s->x, 2 * R * c.hidden);
        PF_ALLOC(s->xn, 2 * R * c.hidden);
        PF_ALLOC(s->r, 2 * R * c.hidden);
        PF_ALLOC(s->qkv, 2 * R * NQKV);
        PF_ALLOC(s->q, 2 * R * QD);
        PF_ALLOC(s->attn, 2 * R * QD);
        PF_ALLOC(s->gu, 2 * R * 2 * c.inter);
        PF_ALLOC(s->act, 2 * R * c.inter);
        PF_ALLOC(s->part_acc, 4 * R * c.n_heads * splits * c.head_dim);
        PF_ALLOC(s->part_ml, 4 * R * c.n_heads * splits * 2);
        PF_ALLOC(s->rope_cs, 4 * R * c.head_dim);
        PF_ALLOC(s->kc, 2 * R * c.n_kv_heads * c.head_dim);
        PF_ALLOC(s->vc, 2 * R * c.n_kv_heads * c.head_dim);
        PF_ALLOC(s->wT, 2 * we);
        PF_ALLOC(s->wM, 2 * we);
        s->rows = rows, s->part_splits = splits, s->w_elems = we;
    }
    if (w_elems > s->w_elems) {
        if (s->wT) (void)hipFree(s->wT);
        if (s->wM) (void)hipFree(s->wM);
        s->wT = nullptr, s->wM = nullptr;
        PF_ALLOC(s->wT, 2 * w_elems);
        PF_ALLOC(s->wM, 2 * w_elems);
        s->w_elems = w_elems, ++s->alloc_gen;
    }
#undef PF_ALLOC
    return PIE_OK;
}

// ---------------------------------------------------------------- the batched forward

// Byte budget for resident T copies of layer matrices, fixed at the first batched prefill: half of the free device memory
// (PIE_KNOB_PREFILL_RESIDENT: 0 disables, <GiB> sets it).  Matrices are admitted in first-use order while the budget lasts
// (8B model: all 15 GB; 70B: ~7/8 of its 140 GB); the rest keep going through the scratch every chunk.
static size_t resident_budget(pie_decoder *d) {
    PrefillScratch *s = d->prefill;
    if (s->resident_mode < 0) {
        size_t free_b = 0, total_b = 0;
        const int k = pie_knob(PIE_KNOB_PREFILL_RESIDENT);
        if (k >= 0) s->resident_left = (size_t)k << 30;
        else s->resident_left = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? free_b / 2 : 0;
        s->resident_mode = 1;
    }
    return s->resident_left;
}

// w16_gemm.hpp / w4m_gemm.hip: the 16-bit many-row MFMA GEMM on W16M tiles (weights in MFMA A-fragment order)
size_t w16m_size(int N, int K);
int w16m_from_rows_launch(const void *w, int N, int K, void *w16m, hipStream_t st);
int w16m_from_w16s_launch(const void *w16s, int N, int K, void *w16m, hipStream_t st);
size_t w16l_workspace_bytes(int M, int N, int K);
int w16l_gemm_launch(int dtype, const void *w16m, const void *x, int ldx, int M, int N, int K, void *y, void *workspace, hipStream_t st, const void *bias,
                     void *swiglu_act, bool *fused, int ldy);

// The 16-bit operand of a many-row product, as W16M tiles: dense modules straight from their W16S units, int8 / group-32 modules
// dequantised (mx.dequantize's arithmetic: the qmm regime multiplies T copies) into the row-major staging buffer and tiled from there.
template <class T>
static int expand_weights(pie_decoder *d, const void *packed, int N, int K, void *w16m, u16 *staging, hipStream_t st) {
    const int wf = d->mat_fmt(packed);  // per-module quantisation: a checkpoint may mix formats (models/utils.py:99-109)
    if (wf == PIE_W_DENSE) return w16m_from_w16s_launch(packed, N, K, w16m, st);
    if (wf == PIE_W_INT8_G64 || wf == PIE_W_INT8_G32) {
        const size_t w8 = (size_t)N * (K >> 2);
        if (wf == PIE_W_INT8_G32) hipLaunchKernelGGL((k_dequant_w8s<T, true>), dim3((unsigned)((w8 + 255) / 256)), dim3(256), 0, st, (const u32 *)packed, N, K, w4s_slices(K), staging);
        else hipLaunchKernelGGL((k_dequant_w8s<T, false>), dim3((unsigned)((w8 + 255) / 256)), dim3(256), 0, st, (const u32 *)packed, N, K, w4s_slices(K), staging);
        PIE_LAUNCH_CHECK();
        return w16m_from_rows_launch(staging, N, K, w16m, st);
    }
    if (wf == PIE_W_INT2_G64) {
        const size_t w2 = (size_t)N * (K >> 4);
        hipLaunchKernelGGL((k_dequant_w2s<T>), dim3((unsigned)((w2 + 255) / 256)), dim3(256), 0, st, (const u32 *)packed, N, K, w4s_slices(K), staging);
        PIE_LAUNCH_CHECK();
        return w16m_from_rows_launch(staging, N, K, w16m, st);
    }
    if (wf == PIE_W_INT6_G64) {
        const size_t w6 = (size_t)N * (K >> 3);
        hipLaunchKernelGGL((k_dequant_w6s<T>), dim3((unsigned)((w6 + 255) / 256)), dim3(256), 0, st, (const u32 *)packed, N, K, w4s_slices(K), staging);
        PIE_LAUNCH_CHECK();
        return w16m_from_rows_launch(staging, N, K, w16m, st);
    }
    const size_t words = (size_t)N * (K >> 3);
    if (wf == PIE_W_INT4_G32) hipLaunchKernelGGL((k_dequant_w4s<T, true>), dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, (const u32 *)packed, N, K, w4s_slices(K), staging);
    else hipLaunchKernelGGL((k_dequant_w4s<T, false>), dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, (const u32 *)packed, N, K, w4s_slices(K), staging);
    PIE_LAUNCH_CHECK();
    return w16m_from_rows_launch(staging, N, K, w16m, st);
}

// y[M, N] = x[M, K] . W^T for one streaming-layout matrix.  keep: a layer matrix (eligible for the resident copy); the
// lm_head of a logits-on-every-position call always goes through the scratch.
// w4m_gemm.hip: the few-row int4 GEMM on W4M tiles
size_t w4m_bytes(int N, int K);
int w4m_repack_launch(const void *w4s, int N, int K, void *w4m, hipStream_t st);
struct W4mRope {  // q|k|v epilogue arguments (defined identically in w4m_gemm.hip)
    const float *rope_cs;
    const DecState *state;
    const int *ctx_len;
    const unsigned long long *kv_table;
    u16 *slab;
    const int *block_table;
    int bt_stride, n_pages, layer, n_layers, n_heads, n_kv_heads, HD, traditional;
    u16 *q_out;
    const u16 *bias;
    size_t i8_page_bytes;  // != 0 (with slab): int8 pages with per-head scales (paged_i8.hip)
};
int w4m_gemm_launch(int dtype, const void *w4m, const void *x, int M, int N, int K, void *y, hipStream_t st, int swiglu, const void *bias, const W4mRope *rope);
int w4l_gemm_launch(int dtype, const void *w4m, const void *x, int M, int N, int K, void *y, void *workspace, hipStream_t st, void *swiglu_act, bool *fused,
                    int *slabs);  // many rows (MFMA-bound)
size_t w4l_workspace_bytes(int M, int N, int K);
// 6 .. 256 rows: the weight-streaming form (w4r_gemm.hpp); epi 0 store (+ bias), 1 SwiGLU (+ bias), 2 RoPE + append
bool w4r_serves(int M, int N, int K);
int w4r_splits(int M, int N, int K);
size_t w4r_workspace_bytes(int M, int N, int K);
bool w4m_wide_scales(const void *w4m);
int w4r_gemm_launch(int dtype, const void *w4m, const void *x, int M, int N, int K, void *y, void *workspace, hipStream_t st, int epi, const void *bias,
                    const W4mRope *rope, int *slabs, bool *bias_done, bool wide_scales);

// int4 checkpoints: prompts beyond small_rows() rows run the hand-written many-row W4 MFMA GEMM on the same W4M tiles -- no 16-bit
// copy of the weights, no hipBLASLt (which remains the path of shapes the tile kernels do not take: N not a multiple of 32).
static bool w4l_enabled() { return true; }
// Dense (16-bit) modules keep the library GEMM: for 16-bit weights hipBLASLt is a plain GEMM done well.  The hand-written 16-bit kernel of
// rounds 2-3 (k_w16l_gemm: 0-20 % slower on prompts, 15-38 % on the vision tower's small shapes) was deleted in round 4 (EXPERIMENTS.md).
// Rows up to which an int4 Linear runs on the W4M kernel instead of the T copy + hipBLASLt (PIE_KNOB_SMALL_M: 0 disables, max 32).
static int small_rows() {
    const int k = pie_knob(PIE_KNOB_SMALL_M);
    const int v = k >= 0 ? k : 32;
    return v < 0 ? 0 : (v > 32 ? 32 : v);
}

template <class T>
static int bias_rows(u16 *y, const void *bias, int M, int N, hipStream_t st) {
    const size_t n8 = (size_t)M * N / 8;
    hipLaunchKernelGGL(k_bias_rows<T>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, y, (const u16 *)bias, n8, N / 8);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

template <class T>
static int linear_rows(pie_decoder *d, const void *packed, int N, int K, const u16 *x, int M, u16 *y, hipStream_t st, bool keep = true,
                       const void *bias = nullptr, bool keep_w4m = false, u16 *act = nullptr, bool *used_act = nullptr, W4mRope *rope = nullptr,
                       W4lSlabs *slabs = nullptr) {
    // slabs: the caller's consumer takes the fp32 slabs of a K-split many-row GEMM (and the bias) instead of y; slabs->S > 1 tells whether it must
    // rope / used_act: for the packed q|k|v matrix the few-row kernel can rotate q / k and append k / v itself
    // act / used_act: for the packed gate|up matrix the few-row kernel can apply the SwiGLU itself and write act [M, N / 2]
    PrefillScratch *s = d->prefill;
    if (used_act) *used_act = false;
    if (slabs) *slabs = W4lSlabs();
    if (pie_knob(PIE_KNOB_W4L_SLABS) == 0) slabs = nullptr;  // always reduce in the GEMM's own launch (the bit-equality test)
    // 8B prompt of 64 / 128 / 256 / 512 / 700 tokens, slabs summed by the consumers vs reduce launches: 3.70 / 4.56 / 6.13 / 9.56 / 14.26 ms
    // vs 3.96 / 4.82 / 6.20 / 9.75 / 14.38.  (With ONE workgroup per row the RoPE consumer could not keep enough slab loads in flight
    // below ~200 rows -- 4.21 ms at 64 tokens -- so few rows get four workgroups each there; that split applies to the RoPE consumer, not to the
    // add + RMSNorm consumers, whose row-wide reduction keeps them at one workgroup per row.)
    const bool is_int4 = d->mat_fmt(packed) == PIE_W_INT4_G64;
    if ((d->mat_fmt(packed) == PIE_W_INT4_G32 || d->mat_fmt(packed) == PIE_W_INT8_G32 || d->mat_fmt(packed) == PIE_W_INT2_G64 || d->mat_fmt(packed) == PIE_W_INT6_G64) && M <= GEMV_ROWS_MAX && K <= 32768 && N % 2 == 0) {  // group-32 / two-bit codes, qmv regime: the streaming GEMV, one pass per row
        GemvArgs a = {};
        a.fmt = d->mat_fmt(packed) == PIE_W_INT2_G64 ? FMT_W2S : d->mat_fmt(packed) == PIE_W_INT6_G64 ? FMT_W6S : d->mat_fmt(packed) == PIE_W_INT8_G32 ? FMT_W8S32 : FMT_W4S32, a.w = (const char *)packed, a.K = K, a.N = N, a.x = x, a.y = y, a.lin_bias = (const u16 *)bias;
        return w4s_gemv_launch(d->cfg.dtype, PRO_NONE, EPI_STORE, a, M, st);
    }
    // Below 6 rows MLX multiplies row by row (qmv: exact fp32 per row, mx.quantized_matmul as reached from nn.QuantizedLinear): the
    // streaming GEMV with the rows' images side by side -- one pass over the weights, each row with the batch-1 arithmetic.
    if (is_int4 && M <= GEMV_ROWS_MAX && K % 64 == 0 && N % 2 == 0 && gemv_rows_lds_bytes(K, 1) <= 160u * 1024u)
        return w4s_gemv_rows_launch(d->cfg.dtype, packed, N, K, x, M, y, (const u16 *)bias, st);
    // resident W4M tile copy of an int4 matrix (0.5625 B per weight, built on first use within the residency budget); nullptr: no room
    auto w4m_tiles = [&](void **out) -> int {
        *out = nullptr;
        auto it = s->resident_w4m.find(packed);
        if (it != s->resident_w4m.end()) {
            *out = it->second;
            return PIE_OK;
        }
        void *wm = nullptr;
        if (resident_budget(d) >= w4m_bytes(N, K) && hipMalloc(&wm, w4m_bytes(N, K)) == hipSuccess) {
            s->resident_w4m[packed] = wm, s->resident_left -= w4m_bytes(N, K), ++s->alloc_gen;
            *out = wm;
            return w4m_repack_launch(packed, N, K, wm, st);
        }
        (void)hipGetLastError();
        return PIE_OK;
    };
    auto w4l_reserve = [&](size_t wb) -> int {  // fp32 slabs of a K-split product
        if (wb > s->w4l_ws_bytes) {
            if (s->w4l_ws) (void)hipFree(s->w4l_ws);
            s->w4l_ws = nullptr, s->w4l_ws_bytes = 0;
            PIE_HIP_TRY(hipMalloc(&s->w4l_ws, wb));
            s->w4l_ws_bytes = wb, ++s->alloc_gen;
        }
        return PIE_OK;
    };
    // 6 .. 256 rows (a chat turn behind a cached prefix, a prompt chunk, a multi-sequence step): the weight-streaming MFMA GEMM, every CU
    // streaming its slab of W4M tiles once (w4r_gemm.hpp).  q|k|v, o_proj and down split K over workgroups and hand fp32 slabs to their
    // consumers (RoPE + append, add + RMSNorm); gate|up carries the SwiGLU in its epilogue.
    if (is_int4 && w4r_serves(M, N, K)) {
        void *wm = nullptr;
        int rc = w4m_tiles(&wm);
        if (rc) return rc;
        if (wm) {
            const bool wide = w4m_wide_scales(wm);
            if (act && used_act) {  // gate|up: SwiGLU (and the Linear's bias) in the epilogue
                *used_act = true;
                return w4r_gemm_launch(d->cfg.dtype, wm, x, M, N, K, act, nullptr, st, 1, bias, nullptr, nullptr, nullptr, wide);
            }
            if (rope && used_act && w4r_splits(M, N, K) == 1) {  // q|k|v wide enough to fill the chip without a K split: RoPE + append in the epilogue
                *used_act = true;
                rope->bias = (const u16 *)bias;
                return w4r_gemm_launch(d->cfg.dtype, wm, x, M, N, K, nullptr, nullptr, st, 2, nullptr, rope, nullptr, nullptr, wide);
            }
            if ((rc = w4l_reserve(w4r_workspace_bytes(M, N, K)))) return rc;
            int n_slabs = 0;
            bool bias_done = false;
            rc = w4r_gemm_launch(d->cfg.dtype, wm, x, M, N, K, y, s->w4l_ws, st, 0, bias, nullptr, slabs ? &n_slabs : nullptr, &bias_done, wide);
            if (!rc && n_slabs > 1) {  // y was NOT written: the consumer sums the slabs, rounds and adds the bias
                slabs->part = (const float *)s->w4l_ws, slabs->S = n_slabs, slabs->MN = (size_t)M * N;
                return PIE_OK;
            }
            if (rc || !bias || bias_done) return rc;
            return bias_rows<T>(y, bias, M, N, st);
        }
    }
    // Up to 32 rows of a shape k_w4r_gemm does not take (K < 256) -- or all of them with knob PIE_KNOB_W4R = 0, the tests' comparator: round 2's
    // first few-row kernel, one workgroup per 32-column strip
    if ((keep || keep_w4m) && is_int4 && M <= small_rows() && N % 32 == 0 && K % 64 == 0) {
        void *wm = nullptr;
        int rc = w4m_tiles(&wm);
        if (rc) return rc;
        if (wm) {
            if (rope && used_act) {
                *used_act = true;
                rope->bias = (const u16 *)bias;
                return w4m_gemm_launch(d->cfg.dtype, wm, x, M, N, K, nullptr, st, 2, nullptr, rope);
            }
            if (act && used_act) {
                *used_act = true;
                return w4m_gemm_launch(d->cfg.dtype, wm, x, M, N, K, act, st, 1, bias, nullptr);
            }
            rc = w4m_gemm_launch(d->cfg.dtype, wm, x, M, N, K, y, st, 0, nullptr, nullptr);
            if (rc || !bias) return rc;
            return bias_rows<T>(y, bias, M, N, st);
        }
    }
    if (is_int4 && N % 32 == 0 && K % 64 == 0 && w4l_enabled()) {  // beyond 256 rows: the many-row tile kernels
        void *wm = nullptr;
        int rc = w4m_tiles(&wm);
        if (rc) return rc;
        if (wm) {
            if ((rc = w4l_reserve(w4l_workspace_bytes(M, N, K)))) return rc;  // fp32 partial tiles of a K-split shape (medium prompts)
            bool fused = false;  // gate|up without a Linear bias: the SwiGLU rides in the GEMM's epilogue where the shape allows
            int n_slabs = 0;
            rc = w4l_gemm_launch(d->cfg.dtype, wm, x, M, N, K, y, s->w4l_ws, st, (act && used_act && !bias) ? act : nullptr, &fused, slabs ? &n_slabs : nullptr);
            if (fused) *used_act = true;
            if (!rc && n_slabs > 1) {  // y was NOT written: the consumer sums the slabs, rounds and adds the bias
                slabs->part = (const float *)s->w4l_ws, slabs->S = n_slabs, slabs->MN = (size_t)M * N;
                return PIE_OK;
            }
            if (rc || !bias) return rc;
            return bias_rows<T>(y, bias, M, N, st);
        }
    }
    // Everything else -- dense modules, int8 and group-32 modules beyond the GEMV's rows -- multiplies a 16-bit copy of the matrix in W16M
    // tiles with the hand-written MFMA GEMM (w16_gemm.hpp); resident copies are built on first use within the budget.
    PIE_REQUIRE(K % 64 == 0, PIE_E_SHAPE, "prefill: a many-row Linear needs in_features % 64 == 0 (pie_set_knob(PIE_KNOB_PREFILL_MIN, 1000000) processes prompts as iterated decode steps)");
    u16 *wM = s->wM;
    bool ready = false;
    if (keep) {
        const size_t bytes = w16m_size(N, K);
        auto it = s->resident.find(packed);
        if (it != s->resident.end()) wM = it->second, ready = true;
        else if (resident_budget(d) >= bytes) {
            if (hipMalloc((void **)&wM, bytes) == hipSuccess) s->resident[packed] = wM, s->resident_left -= bytes, ++s->alloc_gen;
            else (void)hipGetLastError(), wM = s->wM, s->resident_left = 0;  // out of memory: scratch from here on
        }
    }
    int rc = PIE_OK;
    if (!ready && (rc = expand_weights<T>(d, packed, N, K, wM, s->wT, st))) return rc;
    if ((rc = w4l_reserve(w16l_workspace_bytes(M, N, K)))) return rc;  // fp32 slabs of a K-split shape (few rows, narrow matrices)
    bool fused = false;  // gate|up: bias and SwiGLU in the GEMM's epilogue where the shape does not split K
    rc = w16l_gemm_launch(d->cfg.dtype, wM, x, 0, M, N, K, y, s->w4l_ws, st, bias, (act && used_act) ? act : nullptr, &fused, 0);
    if (fused) *used_act = true;
    return rc;
}

template <class T>
static int prefill_t(pie_decoder *d, const int32_t *ids, const void *embeds, int L, void *logits_all, hipStream_t st) {
    const pie_decoder_config &c = d->cfg;
    const int H = c.hidden, D = c.head_dim, QD = c.n_heads * D, KVD = c.n_kv_heads * D, NQKV = QD + 2 * KVD, I = c.inter;
    const int chunk = prefill_chunk_rows() < L ? prefill_chunk_rows() : L;
    size_t w_elems = w16m_size(2 * I, H) / 2;  // the 16-bit copies are W16M tiles (rows padded to 32, columns to 64)
    if (w16m_size(NQKV, H) / 2 > w_elems) w_elems = w16m_size(NQKV, H) / 2;
    if (w16m_size(H, I) / 2 > w_elems) w_elems = w16m_size(H, I) / 2;
    if (logits_all && w16m_size(c.vocab, H) / 2 > w_elems) w_elems = w16m_size(c.vocab, H) / 2;
    int rc = scratch_reserve(d, chunk, w_elems, d->splits);
    if (rc) return rc;
    PrefillScratch *s = d->prefill;
    const bool mfma_attn = pie_knob(PIE_KNOB_PREFILL_ATTN_VALU) != 1;  // 1 forces the row-per-launch-slice VALU kernel (tests compare the two)
    for (int c0 = 0; c0 < L; c0 += chunk) {
        const int M = L - c0 < chunk ? L - c0 : chunk;
        // h = embed_tokens(inputs)  (language.py:176)
        if (embeds) {  // h = inputs_embeds (models/intern/language.py:155-158)
            PIE_HIP_TRY(hipMemcpyAsync(s->x, (const u16 *)embeds + (size_t)c0 * H, (size_t)M * H * 2, hipMemcpyDeviceToDevice, st));
            rc = PIE_OK;
        } else
            rc = d->mat_fmt(d->glob.embed_codes) == PIE_W_DENSE
                 ? pie_embedding_dense(ids + c0, M, d->glob.embed_codes, c.vocab, H, c.dtype, s->x, st)
                 : embedding_launch(ids + c0, M, d->glob.embed_codes, d->glob.embed_scales, d->glob.embed_biases, c.vocab, H, c.dtype, s->x, nullptr,
                                    nullptr, nullptr, 0, st, embed_bits(d));
        if (rc) return rc;
        hipLaunchKernelGGL(k_rope_cs_rows, dim3(M), dim3(64), 0, st, d->glob.rope_freqs, d->state, nullptr, D / 2, s->rope_cs);
        PIE_LAUNCH_CHECK();
        for (int li = 0; li < c.n_layers; ++li) {
            const pie_layer_weights &w = d->layers[li];
            // Attention.__call__ (language.py:75-108) on input_layernorm(x)
            // input_layernorm: layer 0 here; for the later layers it was fused with the previous block's residual add
            if (li == 0 && (rc = pie_rms_norm(s->x, w.attn_norm, c.rms_eps, M, H, c.dtype, s->xn, st))) return rc;
            W4mRope re = {s->rope_cs, d->state, nullptr, d->kv_table, nullptr, d->block_table, 0, d->n_pages, li, c.n_layers, c.n_heads, c.n_kv_heads, D,
                          c.rope_traditional, s->q, nullptr, 0};
            bool roped = false;
            W4lSlabs sq, so, sd;  // K-split products handed over as fp32 slabs (q|k|v only without a bias: RoPE takes T(x W^T + b))
            if ((rc = linear_rows<T>(d, w.wqkv, NQKV, H, s->xn, M, s->qkv, st, true, w.bqkv, false, nullptr, &roped, &re, w.bqkv ? nullptr : &sq)))
                return rc;
            if (!roped) {
                const unsigned row_wgs = sq.S > 1 ? (M < 512 ? 4u : 1u) : 1u;  // few rows of slabs: four workgroups per row (256 tokens: 6.03 vs 6.20 ms; from 512 rows no difference)
                decltype(&k_rope_append_rows<T, false, false>) rope_k = &k_rope_append_rows<T, false, false>;
                if (sq.S > 1) rope_k = &k_rope_append_rows<T, true, false>;
                hipLaunchKernelGGL(rope_k, dim3(M, row_wgs), dim3(256), 0, st, s->qkv, NQKV, d->glob.rope_freqs, d->state, d->kv_table, li,
                                   c.n_layers, c.n_heads, c.n_kv_heads, D, c.rope_traditional, s->q, d->block_table, d->n_pages, s->rope_cs, (const int *)nullptr, 0,
                                   (u16 *)nullptr, (const int *)nullptr, (u16 *)nullptr, (u16 *)nullptr, sq.part, sq.S, sq.MN, (size_t)0);
                PIE_LAUNCH_CHECK();
            }
            if (mfma_attn) {  // causal flash attention on the MFMA units (prefill_attn.hpp)
                PrefillAttnArgs pa = {};
                pa.q = s->q, pa.kv_table = d->kv_table, pa.layer = li, pa.n_layers = c.n_layers, pa.state = d->state;
                pa.block_table = d->block_table, pa.n_pages = d->n_pages;
                pa.M = M, pa.Hq = c.n_heads, pa.Hkv = c.n_kv_heads, pa.scale = 1.0f / sqrtf((float)D), pa.out = s->attn;
                if ((rc = prefill_attn_launch_t<T>(pa, D, st))) return rc;
            } else {  // PIE_KNOB_PREFILL_ATTN_VALU = 1: the VALU decode kernel, one query row per blockIdx.z (cross-check for the tests)
                AttnArgs a = {};
                a.q = s->q, a.kv_table = d->kv_table, a.layer = li, a.n_layers = c.n_layers, a.state = d->state;
                a.block_table = d->block_table, a.n_pages = d->n_pages, a.bt_stride = 0;  // every row reads the one sequence's table
                a.Hq = c.n_heads, a.Hkv = c.n_kv_heads, a.splits = d->splits, a.rows = M, a.scale = 1.0f / sqrtf((float)D);
                a.part_acc = s->part_acc, a.part_ml = s->part_ml, a.out = s->attn;
                if ((rc = attn_decode_launch(c.dtype, D, a, true, st))) return rc;
            }
            if ((rc = linear_rows<T>(d, w.wo, H, QD, s->attn, M, s->r, st, true, w.bo, false, nullptr, nullptr, nullptr,
                                     H <= 8192 ? &so : nullptr)))
                return rc;
            // h = x + r (language.py:151) + post_attention_layernorm(h) for MLP.__call__ (language.py:126-127)
            if ((rc = add_rms_norm_rows<T>(s->x, s->r, w.mlp_norm, c.rms_eps, M, H, c.dtype, s->xn, st, so.S > 1 ? w.bo : nullptr, &so)))
                return rc;
            bool fused_act = false;
            if ((rc = linear_rows<T>(d, w.wgateup, 2 * I, H, s->xn, M, s->gu, st, true, w.bgateup, false, s->act, &fused_act))) return rc;
            if (!fused_act) {
                const size_t n_act = (size_t)M * I;
                hipLaunchKernelGGL(k_swiglu_rows<T>, dim3((unsigned)((n_act / 4 + 255) / 256)), dim3(256), 0, st, s->gu, n_act, s->act);
                PIE_LAUNCH_CHECK();
            }
            const bool fused_next = li + 1 < c.n_layers;
            if ((rc = linear_rows<T>(d, w.wdown, H, I, s->act, M, s->r, st, true, w.bdown, false, nullptr, nullptr, nullptr,
                                     fused_next && H <= 8192 ? &sd : nullptr)))
                return rc;
            // out = h + r (language.py:153), fused with the next block's input_layernorm when there is one
            if (fused_next)
                rc = add_rms_norm_rows<T>(s->x, s->r, d->layers[li + 1].attn_norm, c.rms_eps, M, H, c.dtype, s->xn, st, sd.S > 1 ? w.bdown : nullptr, &sd);
            else rc = pie_add(s->x, s->r, (size_t)M * H, c.dtype, s->x, st);
            if (rc) return rc;
        }
        if (logits_all) {  // lm_head on every position, like the reference (language.py:205-209)
            if ((rc = pie_rms_norm(s->x, d->glob.final_norm, c.rms_eps, M, H, c.dtype, s->xn, st))) return rc;
            if ((rc = linear_rows<T>(d, d->glob.lm_head, c.vocab, H, s->xn, M, (u16 *)logits_all + (size_t)c0 * c.vocab, st, false))) return rc;
        }
        const bool last = c0 + M >= L;
        if (last)  // the last position continues through the decode step's lm_head + tail below
            PIE_HIP_TRY(hipMemcpyAsync(d->h, s->x + (size_t)(M - 1) * H, 2 * (size_t)H, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_add_pos, dim3(1), dim3(1), 0, st, d->state, last ? M - 1 : M);
        PIE_LAUNCH_CHECK();
    }
    // logits[:, -1, :] -> logprobs -> greedy token; the tail advances the device-side offset past the last prompt token
    if ((rc = enqueue_kernel(d, PIE_K_LMHEAD, 0, nullptr, d->logits, st))) return rc;
    return enqueue_kernel(d, PIE_K_TAIL, 0, nullptr, d->logits, st);
}

// mx.fast.scaled_dot_product_attention(q, k, v, scale, mask=causal) for L > 1 (models/base.py:37-53,111-113), op level.
extern "C" int pie_sdpa_prefill(const void *q, const void *k, const void *v, int Hq, int Hkv, int L, int offset, int cap, int D, float scale,
                                int dtype, void *out, void *stream) {
    PIE_REQUIRE(q && k && v && out, PIE_E_ARG, "pie_sdpa_prefill: null pointer");
    PIE_REQUIRE(L >= 1 && offset >= 0 && offset + L <= cap, PIE_E_SHAPE, "pie_sdpa_prefill: need offset + L <= cap");
    PIE_REQUIRE(D == 64 || D == 128, PIE_E_SHAPE, "pie_sdpa_prefill: head_dim must be 64 or 128");
    PIE_REQUIRE(Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= 8, PIE_E_SHAPE, "pie_sdpa_prefill: n_heads / n_kv_heads must be between 1 and 8");
    PIE_REQUIRE(pie_aligned(q, 16) && pie_aligned(k, 16) && pie_aligned(v, 16) && pie_aligned(out, 8), PIE_E_ALIGN, "pie_sdpa_prefill: misaligned pointer");
    PrefillAttnArgs a = {};
    a.q = (const u16 *)q, a.k = (const u16 *)k, a.v = (const u16 *)v, a.offset = offset, a.cap = cap;
    a.M = L, a.Hq = Hq, a.Hkv = Hkv, a.scale = scale, a.out = (u16 *)out;
    if (dtype == PIE_BF16) return prefill_attn_launch_t<BF16>(a, D, (hipStream_t)stream);
    if (dtype == PIE_F16) return prefill_attn_launch_t<F16>(a, D, (hipStream_t)stream);
    return pie::fail(PIE_E_ARG, "pie_sdpa_prefill: dtype must be PIE_BF16 or PIE_F16");
}


// attention splits of a batch of B decode rows over block tables of max_blocks pages
static int batch_attn_splits(const pie_decoder *d, int B, int max_blocks) {
    const pie_decoder_config &c = d->cfg;
    // attention splits: enough workgroups for two per CU across the batch, never more than pages per sequence
    int splits = ((d->kv_i8 ? 1024 : 512) + B * c.n_kv_heads - 1) / (B * c.n_kv_heads);  // (the int8-page kernel runs 4-wave workgroups: twice as many)
    splits = splits > ATTN_MAX_SPLITS ? ATTN_MAX_SPLITS : (splits > max_blocks ? max_blocks : splits);
    splits = splits < 1 ? 1 : splits;
    // short sequences (tables of at most 8 pages) in a batch of 64+ (sequence, kv-head) workgroups: one split, and no combine launch
    // (threshold 128 -> 64, i.e. from 8 sequences of the 8B model: 8 / 12 sequences 2.46 / 2.57 -> 2.40 / 2.46 ms at ~170 positions,
    // 2.59 / 2.78 -> 2.57 / 2.61 at ~420; at 48 the 6-sequence step lost 1 % at ~420 positions)
    if (max_blocks <= 8 && B * c.n_kv_heads >= (d->kv_i8 ? 512 : 64)) splits = 1;  // (the int8-page kernel's workgroups are half as wide)
    // a handful of short sequences: one split as well -- the split kernel + its combine launch cost 10.6 us per layer against 5.5 for the
    // batch-1 step's attention, and a few hundred positions are a handful of row blocks per wave
    if (B <= GEMV_ROWS_MAX && max_blocks <= 4 && !d->kv_i8) splits = 1;
    return splits;
}

// ---------------------------------------------------------------- one decode step for B sequences (continuous batching)
// The weights stream once for all B rows (few-row int4 GEMM for B <= 32, the T copy + hipBLASLt beyond); each row is its own
// sequence: RoPE at its own position, K/V appended to its own page, attention over its own block table (k_attn_decode PAGED,
// one sequence per blockIdx.z), lm_head + tail on every row.  What the reference's Scheduler / BatchDetails / PagedAttention
// skeleton (src/pie_core/include/engine/batch_details.hpp:10-88, scheduler.hpp) describes for decode-state sequences.
template <class T>
static int decode_batch_t(pie_decoder *d, const int32_t *tokens, const int32_t *ctx_len, const void *const *slabs, int n_pages,
                          const int32_t *block_tables, int max_blocks, int B, u16 *logits, float *logprobs, int32_t *next_tokens, hipStream_t st) {
    const pie_decoder_config &c = d->cfg;
    const int H = c.hidden, D = c.head_dim, QD = c.n_heads * D, KVD = c.n_kv_heads * D, NQKV = QD + 2 * KVD, I = c.inter;
    size_t w_elems = w16m_size(2 * I, H) / 2;  // the 16-bit copies are W16M tiles (rows padded to 32, columns to 64)
    if (w16m_size(NQKV, H) / 2 > w_elems) w_elems = w16m_size(NQKV, H) / 2;
    if (w16m_size(H, I) / 2 > w_elems) w_elems = w16m_size(H, I) / 2;
    if (w16m_size(c.vocab, H) / 2 > w_elems) w_elems = w16m_size(c.vocab, H) / 2;
    const int splits = batch_attn_splits(d, B, max_blocks);
    int rc = scratch_reserve(d, B, w_elems, splits > d->splits ? splits : d->splits);
    if (rc) return rc;
    PrefillScratch *s = d->prefill;
    const int lm_waves = w4s_gemv_waves(c.vocab, H);  // the fused few-sequence form: one log-softmax partial per GEMV wave and row
    const size_t stats_per_row = (size_t)(lm_waves > TAIL_STAT_TILES ? lm_waves : TAIL_STAT_TILES);
    if ((rc = tail_stats_reserve(s, stats_per_row * (size_t)B))) return rc;
    // Up to 5 sequences on an int4 checkpoint: the batch-1 launch sequence, once, with every GEMV multiplying all rows per weight
    // unit (k_w4s_gemv_rows: RMSNorm prologues; RoPE + page append, residual, SwiGLU and logits epilogues, per row) -- 6 launches per
    // layer instead of 9-10, each row in MLX's row-by-row fp32 regime.  Biases, int8 pages and hidden sizes beyond 8192 take the
    // general path below.
    bool fused_rows = B <= GEMV_ROWS_MAX && d->uniform_int4() && !d->kv_i8 && H <= 8192 && lm_waves <= TAIL_MAX_STATS;
    for (const pie_layer_weights &w : d->layers)
        if (w.bqkv || w.bo || w.bgateup || w.bdown) fused_rows = false;
    if (fused_rows) {
        rc = embedding_launch(tokens, B, d->glob.embed_codes, d->glob.embed_scales, d->glob.embed_biases, c.vocab, H, c.dtype, s->x, nullptr, nullptr, nullptr, 0, st, 4);
        if (rc) return rc;
        hipLaunchKernelGGL(k_rope_cs_rows, dim3(B), dim3(64), 0, st, d->glob.rope_freqs, nullptr, ctx_len, D / 2, s->rope_cs);
        PIE_LAUNCH_CHECK();
        for (int li = 0; li < c.n_layers; ++li) {
            const pie_layer_weights &w = d->layers[li];
            GemvRowsArgs g = {};
            g.w = (const char *)w.wqkv, g.K = H, g.N = NQKV, g.M = B, g.x = s->x, g.norm_w = (const u16 *)w.attn_norm, g.eps = c.rms_eps;
            g.rope_cs = s->rope_cs, g.ctx_len = ctx_len, g.block_table = block_tables, g.slab = (u16 *)slabs[li], g.q_out = s->q;
            g.bt_stride = max_blocks, g.n_pages = n_pages, g.n_heads = c.n_heads, g.n_kv_heads = c.n_kv_heads, g.head_dim = D, g.rope_traditional = c.rope_traditional;
            if ((rc = w4s_gemv_rows_fused_launch(c.dtype, PRO_RMSNORM, EPI_ROPE_KV, g, st))) return rc;
            AttnArgs a = {};
            a.q = s->q, a.slab = (const u16 *)slabs[li], a.block_table = block_tables, a.ctx_len = ctx_len, a.bt_stride = max_blocks, a.n_pages = n_pages;
            a.rows = B, a.Hq = c.n_heads, a.Hkv = c.n_kv_heads, a.splits = splits, a.scale = 1.0f / sqrtf((float)D);
            a.nt_kv = (size_t)B * max_blocks * 64 >= 2048;
            a.part_acc = s->part_acc, a.part_ml = s->part_ml, a.out = s->attn;
            if ((rc = attn_decode_launch(c.dtype, D, a, true, st))) return rc;
            g = GemvRowsArgs();
            g.w = (const char *)w.wo, g.K = QD, g.N = H, g.M = B, g.x = s->attn, g.resid = s->x;
            if ((rc = w4s_gemv_rows_fused_launch(c.dtype, PRO_NONE, EPI_RESIDUAL, g, st))) return rc;
            g = GemvRowsArgs();
            g.w = (const char *)w.wgateup, g.K = H, g.N = 2 * I, g.M = B, g.x = s->x, g.norm_w = (const u16 *)w.mlp_norm, g.eps = c.rms_eps, g.y = s->act;
            if ((rc = w4s_gemv_rows_fused_launch(c.dtype, PRO_RMSNORM, EPI_SWIGLU, g, st))) return rc;
            g = GemvRowsArgs();
            g.w = (const char *)w.wdown, g.K = I, g.N = H, g.M = B, g.x = s->act, g.resid = s->x;
            if ((rc = w4s_gemv_rows_fused_launch(c.dtype, PRO_NONE, EPI_RESIDUAL, g, st))) return rc;
        }
        GemvRowsArgs g = {};
        g.w = (const char *)d->glob.lm_head, g.K = H, g.N = c.vocab, g.M = B, g.x = s->x, g.norm_w = (const u16 *)d->glob.final_norm, g.eps = c.rms_eps;
        g.y = logits, g.stats = s->tail_stats;
        if ((rc = w4s_gemv_rows_fused_launch(c.dtype, PRO_RMSNORM, EPI_LOGITS, g, st))) return rc;
        const dim3 fg(TAIL_FINISH_BLOCKS, B);
        if (c.dtype == PIE_BF16)
            hipLaunchKernelGGL(k_logits_finish<BF16>, fg, dim3(256), 0, st, logits, c.vocab, s->tail_stats, lm_waves, logprobs, next_tokens, (DecState *)nullptr, (int *)nullptr, 0, (const unsigned *)nullptr);
        else
            hipLaunchKernelGGL(k_logits_finish<F16>, fg, dim3(256), 0, st, logits, c.vocab, s->tail_stats, lm_waves, logprobs, next_tokens, (DecState *)nullptr, (int *)nullptr, 0, (const unsigned *)nullptr);
        PIE_LAUNCH_CHECK();
        return PIE_OK;
    }
    rc = d->mat_fmt(d->glob.embed_codes) == PIE_W_DENSE
             ? pie_embedding_dense(tokens, B, d->glob.embed_codes, c.vocab, H, c.dtype, s->x, st)
             : embedding_launch(tokens, B, d->glob.embed_codes, d->glob.embed_scales, d->glob.embed_biases, c.vocab, H, c.dtype, s->x, nullptr, nullptr,
                                nullptr, 0, st, embed_bits(d));
    if (rc) return rc;
    hipLaunchKernelGGL(k_rope_cs_rows, dim3(B), dim3(64), 0, st, d->glob.rope_freqs, nullptr, ctx_len, D / 2, s->rope_cs);
    PIE_LAUNCH_CHECK();
    for (int li = 0; li < c.n_layers; ++li) {
        const pie_layer_weights &w = d->layers[li];
        if (li == 0 && (rc = pie_rms_norm(s->x, w.attn_norm, c.rms_eps, B, H, c.dtype, s->xn, st))) return rc;
        const size_t i8pb = d->kv_i8 ? pie_page_i8_bytes(c.n_kv_heads, D) : 0;  // int8 pages: the append (epilogue or row kernel) quantises
        W4mRope re = {s->rope_cs, nullptr, ctx_len, nullptr, (u16 *)slabs[li], block_tables, max_blocks, n_pages, li, c.n_layers, c.n_heads, c.n_kv_heads,
                      D, c.rope_traditional, s->q, nullptr, i8pb};
        bool roped = false;
        W4lSlabs sq;  // q|k|v as the fp32 slabs of a K-split product (no Linear bias: RoPE takes T(x W^T + b))
        if ((rc = linear_rows<T>(d, w.wqkv, NQKV, H, s->xn, B, s->qkv, st, true, w.bqkv, false, nullptr, &roped, &re, w.bqkv ? nullptr : &sq)))
            return rc;
        if (!roped) {
            decltype(&k_rope_append_rows<T, false, false>) rope_k = &k_rope_append_rows<T, false, false>;
            if (sq.S > 1) rope_k = d->kv_i8 ? &k_rope_append_rows<T, true, true> : &k_rope_append_rows<T, true, false>;
            else if (d->kv_i8) rope_k = &k_rope_append_rows<T, false, true>;
            hipLaunchKernelGGL(rope_k, dim3(B, sq.S > 1 ? 4u : 1u), dim3(256), 0, st, s->qkv, NQKV, d->glob.rope_freqs, nullptr, nullptr, li, c.n_layers,
                               c.n_heads, c.n_kv_heads, D, c.rope_traditional, s->q, block_tables, n_pages, s->rope_cs, ctx_len, max_blocks, (u16 *)slabs[li],
                               (const int *)nullptr, (u16 *)nullptr, (u16 *)nullptr, sq.part, sq.S, sq.MN, i8pb);
            PIE_LAUNCH_CHECK();
        }
        AttnArgs a = {};
        a.q = s->q, a.slab = (const u16 *)slabs[li], a.block_table = block_tables, a.ctx_len = ctx_len, a.bt_stride = max_blocks, a.n_pages = n_pages;
        a.rows = B, a.Hq = c.n_heads, a.Hkv = c.n_kv_heads, a.splits = splits, a.scale = 1.0f / sqrtf((float)D);
        a.nt_kv = (size_t)B * max_blocks * 64 >= 2048;
        a.part_acc = s->part_acc, a.part_ml = s->part_ml, a.out = s->attn;
        if ((rc = d->kv_i8 ? paged_attn_i8_launch(c.dtype, D, a, st) : attn_decode_launch(c.dtype, D, a, true, st))) return rc;
        W4lSlabs so, sd;  // o_proj / down handed over as K-split fp32 slabs where the shape qualifies
        if ((rc = linear_rows<T>(d, w.wo, H, QD, s->attn, B, s->r, st, true, w.bo, false, nullptr, nullptr, nullptr,
                                 H <= 8192 ? &so : nullptr)))
            return rc;
        if ((rc = add_rms_norm_rows<T>(s->x, s->r, w.mlp_norm, c.rms_eps, B, H, c.dtype, s->xn, st, so.S > 1 ? w.bo : nullptr, &so)))
            return rc;
        bool fused_act = false;
        if ((rc = linear_rows<T>(d, w.wgateup, 2 * I, H, s->xn, B, s->gu, st, true, w.bgateup, false, s->act, &fused_act))) return rc;
        if (!fused_act) {
            const size_t n_act = (size_t)B * I;
            hipLaunchKernelGGL(k_swiglu_rows<T>, dim3((unsigned)((n_act / 4 + 255) / 256)), dim3(256), 0, st, s->gu, n_act, s->act);
            PIE_LAUNCH_CHECK();
        }
        if ((rc = linear_rows<T>(d, w.wdown, H, I, s->act, B, s->r, st, true, w.bdown, false, nullptr, nullptr, nullptr,
                                 H <= 8192 ? &sd : nullptr)))
            return rc;
        const void *next_norm = li + 1 < c.n_layers ? d->layers[li + 1].attn_norm : d->glob.final_norm;  // final norm: language.py:187
        if ((rc = add_rms_norm_rows<T>(s->x, s->r, next_norm, c.rms_eps, B, H, c.dtype, s->xn, st, sd.S > 1 ? w.bdown : nullptr, &sd)))
            return rc;
    }
    if ((rc = linear_rows<T>(d, d->glob.lm_head, c.vocab, H, s->xn, B, logits, st, false, nullptr, true))) return rc;
    return logits_tail_rows_launch(c.dtype, logits, c.vocab, B, s->tail_stats, logprobs, next_tokens, st);
}


// ---------------------------------------------------------------- several fresh prompts in one pass
__global__ void k_gather_rows(const uint4 *x, const int *rows, int H8, uint4 *y) {  // y[s] = x[rows[s]], H8 = hidden / 8
    const uint4 *src = x + (size_t)rows[blockIdx.x] * H8;
    for (int i = threadIdx.x; i < H8; i += blockDim.x) y[(size_t)blockIdx.x * H8 + i] = src[i];
}

// The rows of S prompts concatenated ([N] ids): one pass of GEMMs over all N rows, RoPE at each row's own position, K / V into
// each prompt's own pages AND a contiguous copy the attention of this pass reads (causal inside each prompt = the segment
// kernel with seg_hi[r] = r + 1), lm_head + tail on every prompt's last row.  Fresh prompts only (nothing cached before them).
// n_decode > 0 (a mixed batch, batch_details.hpp:10-88's prefill- and decode-state sequences in one BatchDetails): rows
// [0, n_decode) are decode-state sequences -- row r is sequence r (row_seq[r] == r, block-table row r) at position
// row_ctx[r] - 1 with a trivial segment [r, r + 1) -- whose attention then runs over their pages (the multi-sequence step's
// paged kernel) and replaces the segment kernel's rows; the prompt rows follow.  One pass over the weights for both kinds.
template <class T>
static int prefill_varlen_t(pie_decoder *d, const int32_t *ids, const int32_t *row_ctx, const int32_t *row_seq, const int32_t *seg_lo,
                            const int32_t *seg_hi, const int32_t *last_rows, int N, int S, int n_decode, const void *const *slabs, int n_pages,
                            const int32_t *block_tables, int max_blocks, u16 *logits, float *logprobs, int32_t *next_tokens, hipStream_t st,
                            int n_chunks = 0, const int32_t *chunks = nullptr) {
    const pie_decoder_config &c = d->cfg;
    const int H = c.hidden, D = c.head_dim, QD = c.n_heads * D, KVD = c.n_kv_heads * D, NQKV = QD + 2 * KVD, I = c.inter;
    size_t w_elems = w16m_size(2 * I, H) / 2;  // the 16-bit copies are W16M tiles (rows padded to 32, columns to 64)
    if (w16m_size(NQKV, H) / 2 > w_elems) w_elems = w16m_size(NQKV, H) / 2;
    if (w16m_size(H, I) / 2 > w_elems) w_elems = w16m_size(H, I) / 2;
    if (w16m_size(c.vocab, H) / 2 > w_elems) w_elems = w16m_size(c.vocab, H) / 2;
    const int dsplits = n_decode > 0 ? batch_attn_splits(d, n_decode, max_blocks) : 1;
    int rc = scratch_reserve(d, N > S ? N : S, w_elems, dsplits > d->splits ? dsplits : d->splits);
    if (rc) return rc;
    PrefillScratch *s = d->prefill;
    if ((rc = tail_stats_reserve(s, (size_t)TAIL_STAT_TILES * (size_t)S))) return rc;
    rc = d->mat_fmt(d->glob.embed_codes) == PIE_W_DENSE
             ? pie_embedding_dense(ids, N, d->glob.embed_codes, c.vocab, H, c.dtype, s->x, st)
             : embedding_launch(ids, N, d->glob.embed_codes, d->glob.embed_scales, d->glob.embed_biases, c.vocab, H, c.dtype, s->x, nullptr, nullptr,
                                nullptr, 0, st, embed_bits(d));
    if (rc) return rc;
    hipLaunchKernelGGL(k_rope_cs_rows, dim3(N), dim3(64), 0, st, d->glob.rope_freqs, nullptr, row_ctx, D / 2, s->rope_cs);
    PIE_LAUNCH_CHECK();
    for (int li = 0; li < c.n_layers; ++li) {
        const pie_layer_weights &w = d->layers[li];
        if (li == 0 && (rc = pie_rms_norm(s->x, w.attn_norm, c.rms_eps, N, H, c.dtype, s->xn, st))) return rc;
        W4lSlabs sq;  // q|k|v as the fp32 slabs of a K-split product (no Linear bias: RoPE takes T(x W^T + b))
        if ((rc = linear_rows<T>(d, w.wqkv, NQKV, H, s->xn, N, s->qkv, st, true, w.bqkv, false, nullptr, nullptr, nullptr, w.bqkv ? nullptr : &sq)))
            return rc;
        decltype(&k_rope_append_rows<T, false, false>) rope_k = &k_rope_append_rows<T, false, false>;
        if (sq.S > 1) rope_k = d->kv_i8 ? &k_rope_append_rows<T, true, true> : &k_rope_append_rows<T, true, false>;
        else if (d->kv_i8) rope_k = &k_rope_append_rows<T, false, true>;
        hipLaunchKernelGGL(rope_k, dim3(N, sq.S > 1 && N < 512 ? 4u : 1u), dim3(256), 0, st, s->qkv, NQKV, d->glob.rope_freqs, nullptr, nullptr, li, c.n_layers,
                           c.n_heads, c.n_kv_heads, D, c.rope_traditional, s->q, block_tables, n_pages, s->rope_cs, row_ctx, max_blocks, (u16 *)slabs[li],
                           row_seq, s->kc, s->vc, sq.part, sq.S, sq.MN,
                           d->kv_i8 ? pie_page_i8_bytes(c.n_kv_heads, D) : (size_t)0);  // int8 pages: quantised on the way in; this pass's own attention reads the T copies
        PIE_LAUNCH_CHECK();
        PrefillAttnArgs pa = {};
        pa.q = s->q, pa.k = s->kc, pa.v = s->vc, pa.offset = 0, pa.cap = N, pa.seg_lo = seg_lo, pa.seg_hi = seg_hi;
        pa.M = N, pa.Hq = c.n_heads, pa.Hkv = c.n_kv_heads, pa.scale = 1.0f / sqrtf((float)D), pa.out = s->attn;
        if ((rc = segment_attn_gqa_launch_t<T>(pa, D, st))) return rc;
        if (n_decode > 0) {  // the decode-state rows attend to their pages (which hold the row just appended)
            AttnArgs a = {};
            a.q = s->q, a.slab = (const u16 *)slabs[li], a.block_table = block_tables, a.ctx_len = row_ctx, a.bt_stride = max_blocks, a.n_pages = n_pages;
            a.rows = n_decode, a.Hq = c.n_heads, a.Hkv = c.n_kv_heads, a.splits = dsplits, a.scale = pa.scale;
            a.nt_kv = (size_t)n_decode * max_blocks * 64 >= 2048;
            a.part_acc = s->part_acc, a.part_ml = s->part_ml, a.out = s->attn;
            if ((rc = d->kv_i8 ? paged_attn_i8_launch(c.dtype, D, a, st) : attn_decode_launch(c.dtype, D, a, true, st))) return rc;
        }
        for (int ci = 0; ci < n_chunks; ++ci) {  // prompts continuing a cached prefix: causal attention over their pages from the cached length
            const int32_t *ch = chunks + 4 * ci;
            PrefillAttnArgs ca = {};
            ca.q = s->q + (size_t)ch[0] * QD, ca.out = s->attn + (size_t)ch[0] * QD;
            ca.k = (const u16 *)slabs[li], ca.v = (const u16 *)slabs[li] + (size_t)c.n_kv_heads * PIE_PAGE_TOKENS * D;
            ca.offset = ch[2], ca.cap = PIE_PAGE_TOKENS, ca.block_table = block_tables + (size_t)ch[3] * max_blocks, ca.n_pages = n_pages;
            ca.M = ch[1], ca.Hq = c.n_heads, ca.Hkv = c.n_kv_heads, ca.scale = pa.scale;
            if ((rc = prefill_attn_launch_t<T>(ca, D, st))) return rc;
        }
        W4lSlabs so, sd;  // K-split products handed to their consumers as fp32 slabs (as in the single-prompt path)
        if ((rc = linear_rows<T>(d, w.wo, H, QD, s->attn, N, s->r, st, true, w.bo, false, nullptr, nullptr, nullptr,
                                 H <= 8192 ? &so : nullptr)))
            return rc;
        if ((rc = add_rms_norm_rows<T>(s->x, s->r, w.mlp_norm, c.rms_eps, N, H, c.dtype, s->xn, st, so.S > 1 ? w.bo : nullptr, &so)))
            return rc;
        bool fused_act = false;
        if ((rc = linear_rows<T>(d, w.wgateup, 2 * I, H, s->xn, N, s->gu, st, true, w.bgateup, false, s->act, &fused_act))) return rc;
        if (!fused_act) {
            const size_t n_act = (size_t)N * I;
            hipLaunchKernelGGL(k_swiglu_rows<T>, dim3((unsigned)((n_act / 4 + 255) / 256)), dim3(256), 0, st, s->gu, n_act, s->act);
            PIE_LAUNCH_CHECK();
        }
        if ((rc = linear_rows<T>(d, w.wdown, H, I, s->act, N, s->r, st, true, w.bdown, false, nullptr, nullptr, nullptr,
                                 H <= 8192 ? &sd : nullptr)))
            return rc;
        const void *next_norm = li + 1 < c.n_layers ? d->layers[li + 1].attn_norm : d->glob.final_norm;
        if ((rc = add_rms_norm_rows<T>(s->x, s->r, next_norm, c.rms_eps, N, H, c.dtype, s->xn, st, sd.S > 1 ? w.bdown : nullptr, &sd)))
            return rc;
    }
    // the normalised last row of every prompt -> lm_head -> tail
    hipLaunchKernelGGL(k_gather_rows, dim3(S), dim3(256), 0, st, (const uint4 *)s->xn, last_rows, H / 8, (uint4 *)s->r);
    PIE_LAUNCH_CHECK();
    if ((rc = linear_rows<T>(d, d->glob.lm_head, c.vocab, H, s->r, S, logits, st, false, nullptr, true))) return rc;
    return logits_tail_rows_launch(c.dtype, logits, c.vocab, S, s->tail_stats, logprobs, next_tokens, st);
}

static size_t active_page_bytes(const pie_decoder *d);
static int varlen_batch(pie_decoder *d, const int32_t *ids, const int32_t *row_context_lens, const int32_t *row_seq, const int32_t *seg_lo,
                        const int32_t *seg_hi, const int32_t *last_rows, int N, int S, int n_decode, const void *const *slabs, size_t n_pages, size_t slab_bytes,
                        const int32_t *block_tables, int max_blocks, void *logits, float *logprobs, int32_t *next_tokens, void *stream, int n_chunks = 0,
                        const int32_t *chunks = nullptr) {
    PIE_REQUIRE(d && ids && row_context_lens && row_seq && seg_lo && seg_hi && last_rows && slabs && block_tables && logits && logprobs && next_tokens,
                PIE_E_ARG, "pie_decoder_prefill_batch / _step_mixed: null pointer");
    PIE_REQUIRE(n_chunks >= 0 && (n_chunks == 0 || chunks), PIE_E_ARG, "pie_decoder_step_mixed: chunks without their descriptor array");
    PIE_REQUIRE(n_chunks == 0 || !d->kv_i8, PIE_E_STATE, "pie_decoder_step_mixed: a prompt continuing a cached prefix reads T pages (int8 pages: fresh prompts and decode rows only)");
    for (int ci = 0; ci < n_chunks; ++ci) {
        const int32_t *ch = chunks + 4 * ci;
        PIE_REQUIRE(ch[0] >= n_decode && ch[1] >= 1 && ch[0] + ch[1] <= N && ch[2] >= 1 && ch[3] >= 0 && ch[3] < S &&
                        (ch[2] + ch[1] + PIE_PAGE_TOKENS - 1) / PIE_PAGE_TOKENS <= max_blocks,
                    PIE_E_SHAPE, "pie_decoder_step_mixed: a chunk is {first row >= n_decode, rows >= 1, cached positions >= 1, sequence < S} inside the batch and its block table");
    }
    PIE_REQUIRE(d->glob_set, PIE_E_STATE, "pie_decoder_prefill_batch / _step_mixed: set_globals must be called first");
    for (char s : d->layer_set) PIE_REQUIRE(s, PIE_E_STATE, "pie_decoder_prefill_batch / _step_mixed: a layer has no weights (pie_decoder_set_layer)");
    PIE_REQUIRE(S >= 1 && N >= S && N <= 65535 && max_blocks > 0 && n_pages > 0 && n_pages < 0x7FFFFFFFu, PIE_E_SHAPE, "pie_decoder_prefill_batch / _step_mixed: bad batch shape");
    PIE_REQUIRE(slab_bytes >= n_pages * active_page_bytes(d), PIE_E_SHAPE,
                "pie_decoder_prefill_batch / _step_mixed: the slabs are smaller than n_pages pages of the active page format (an int8 pool needs PIE_OPT_KV_I8, a T pool must not have it)");
    PIE_REQUIRE(d->cfg.hidden % 8 == 0 && d->cfg.hidden <= 8192, PIE_E_SHAPE, "pie_decoder_prefill_batch / _step_mixed: hidden must be a multiple of 8, at most 8192");
    const int rep = d->cfg.n_heads / d->cfg.n_kv_heads;
    PIE_REQUIRE(rep >= 1 && rep <= 8, PIE_E_SHAPE, "pie_decoder_prefill_batch / _step_mixed: n_heads / n_kv_heads must be between 1 and 8");
    for (int i = 0; i < d->cfg.n_layers; ++i) PIE_REQUIRE(slabs[i] && pie_aligned(slabs[i], 16), PIE_E_ALIGN, "pie_decoder_prefill_batch / _step_mixed: null or misaligned slab");
    hipStream_t st = (hipStream_t)stream;
    return d->cfg.dtype == PIE_BF16 ? prefill_varlen_t<BF16>(d, ids, row_context_lens, row_seq, seg_lo, seg_hi, last_rows, N, S, n_decode, slabs, (int)n_pages,
                                                             block_tables, max_blocks, (u16 *)logits, logprobs, next_tokens, st, n_chunks, chunks)
                                    : prefill_varlen_t<F16>(d, ids, row_context_lens, row_seq, seg_lo, seg_hi, last_rows, N, S, n_decode, slabs, (int)n_pages,
                                                            block_tables, max_blocks, (u16 *)logits, logprobs, next_tokens, st, n_chunks, chunks);
}

extern "C" int pie_decoder_prefill_batch(pie_decoder *d, const int32_t *ids, const int32_t *row_context_lens, const int32_t *row_seq,
                                         const int32_t *seg_lo, const int32_t *seg_hi, const int32_t *last_rows, int N, int S,
                                         const void *const *slabs, size_t n_pages, size_t slab_bytes, const int32_t *block_tables, int max_blocks, void *logits,
                                         float *logprobs, int32_t *next_tokens, void *stream) {
    return varlen_batch(d, ids, row_context_lens, row_seq, seg_lo, seg_hi, last_rows, N, S, 0, slabs, n_pages, slab_bytes, block_tables,
                        max_blocks, logits, logprobs, next_tokens, stream);
}

extern "C" int pie_decoder_step_mixed(pie_decoder *d, const int32_t *ids, const int32_t *row_context_lens, const int32_t *row_seq,
                                      const int32_t *seg_lo, const int32_t *seg_hi, const int32_t *out_rows, int N, int S, int n_decode,
                                      const void *const *slabs, size_t n_pages, size_t slab_bytes, const int32_t *block_tables, int max_blocks, void *logits,
                                      float *logprobs, int32_t *next_tokens, int n_chunks, const int32_t *chunks_host, void *stream) {
    PIE_REQUIRE(n_decode >= 0 && n_decode <= S, PIE_E_SHAPE, "pie_decoder_step_mixed: n_decode must be between 0 and the number of output rows");
    return varlen_batch(d, ids, row_context_lens, row_seq, seg_lo, seg_hi, out_rows, N, S, n_decode, slabs, n_pages, slab_bytes, block_tables,
                        max_blocks, logits, logprobs, next_tokens, stream, n_chunks, chunks_host);
}

static int decode_batch(pie_decoder *d, const int32_t *tokens, const int32_t *ctx_len, const void *const *slabs, int n_pages, const int32_t *block_tables,
                        int max_blocks, int B, void *logits, float *logprobs, int32_t *next_tokens, hipStream_t st) {
    return d->cfg.dtype == PIE_BF16
               ? decode_batch_t<BF16>(d, tokens, ctx_len, slabs, n_pages, block_tables, max_blocks, B, (u16 *)logits, logprobs, next_tokens, st)
               : decode_batch_t<F16>(d, tokens, ctx_len, slabs, n_pages, block_tables, max_blocks, B, (u16 *)logits, logprobs, next_tokens, st);
}

// bytes one page of the active format takes in a layer's slab: K block + V block of T rows, or the int8 page with its scales
static size_t active_page_bytes(const pie_decoder *d) {
    const pie_decoder_config &c = d->cfg;
    return d->kv_i8 ? pie_page_i8_bytes(c.n_kv_heads, c.head_dim) : (size_t)2 * PIE_PAGE_TOKENS * c.n_kv_heads * c.head_dim * 2;
}

extern "C" int pie_decoder_step_batch(pie_decoder *d, const int32_t *tokens, const int32_t *context_lens, const void *const *slabs, size_t n_pages, size_t slab_bytes,
                                      const int32_t *block_tables, int max_blocks, int B, void *logits, float *logprobs, int32_t *next_tokens,
                                      int flags, void *stream) {
    PIE_REQUIRE(d && tokens && context_lens && slabs && block_tables && logits && logprobs && next_tokens, PIE_E_ARG, "pie_decoder_step_batch: null pointer");
    PIE_REQUIRE(d->glob_set, PIE_E_STATE, "pie_decoder_step_batch: set_globals must be called first");
    PIE_REQUIRE(!d->tp(), PIE_E_STATE, "pie_decoder_step_batch: not available on a tensor-parallel shard");
    for (char s : d->layer_set) PIE_REQUIRE(s, PIE_E_STATE, "pie_decoder_step_batch: a layer has no weights (pie_decoder_set_layer)");
    PIE_REQUIRE(B >= 1 && B <= 4096 && max_blocks > 0 && n_pages > 0 && n_pages < 0x7FFFFFFFu, PIE_E_SHAPE, "pie_decoder_step_batch: bad batch shape");
    PIE_REQUIRE(slab_bytes >= n_pages * active_page_bytes(d), PIE_E_SHAPE,
                "pie_decoder_step_batch: the slabs are smaller than n_pages pages of the active page format (an int8 pool needs PIE_OPT_KV_I8, a T pool must not have it)");
    const int rep = d->cfg.n_heads / d->cfg.n_kv_heads;
    PIE_REQUIRE(rep >= 1 && rep <= 8, PIE_E_SHAPE, "pie_decoder_step_batch: n_heads / n_kv_heads must be between 1 and 8");
    for (int i = 0; i < d->cfg.n_layers; ++i) PIE_REQUIRE(slabs[i] && pie_aligned(slabs[i], 16), PIE_E_ALIGN, "pie_decoder_step_batch: null or misaligned slab");
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & PIE_STEP_GRAPH)) return decode_batch(d, tokens, context_lens, slabs, (int)n_pages, block_tables, max_blocks, B, logits, logprobs, next_tokens, st);
    // Graph replay.  The launches bake in the caller's buffers and this library's scratch / resident weight copies, so a graph is
    // captured only for a call that repeats the previous call's buffers after that call allocated nothing, and is dropped when
    // either changes.  (First call with new buffers: eager, it may allocate; second: capture; from the third: replay.)
    std::vector<uintptr_t> key = {(uintptr_t)tokens, (uintptr_t)context_lens, (uintptr_t)block_tables, (uintptr_t)logits, (uintptr_t)logprobs,
                                  (uintptr_t)next_tokens, (uintptr_t)n_pages, (uintptr_t)max_blocks, (uintptr_t)B, (uintptr_t)d->kv_i8 /* the page format is baked into the launches too */};
    for (int i = 0; i < d->cfg.n_layers; ++i) key.push_back((uintptr_t)slabs[i]);
    if (!d->prefill) d->prefill = new PrefillScratch();
    PrefillScratch *s = d->prefill;
    if (s->batch_graph && s->batch_key == key && s->batch_gen == s->alloc_gen) {
        PIE_HIP_TRY(hipGraphLaunch(s->batch_graph, st));
        return PIE_OK;
    }
    if (s->warm_key == key && s->warm_gen == s->alloc_gen) {
        if (s->batch_graph) (void)hipGraphExecDestroy(s->batch_graph), s->batch_graph = nullptr;
        hipGraph_t g = nullptr;
        hipStream_t cs = nullptr;
        PIE_HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(cs);
            return pie::fail(PIE_E_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
        }
        const unsigned gen = s->alloc_gen;
        const int rc = decode_batch(d, tokens, context_lens, slabs, (int)n_pages, block_tables, max_blocks, B, logits, logprobs, next_tokens, cs);
        e = hipStreamEndCapture(cs, &g);
        (void)hipStreamDestroy(cs);
        if (rc || e != hipSuccess || d->prefill->alloc_gen != gen) {  // something allocated or failed under capture: run this call eagerly instead
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            d->prefill->warm_key.clear();
            return decode_batch(d, tokens, context_lens, slabs, (int)n_pages, block_tables, max_blocks, B, logits, logprobs, next_tokens, st);
        }
        s = d->prefill;
        e = hipGraphInstantiate(&s->batch_graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return pie::fail(PIE_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        s->batch_key = key, s->batch_gen = s->alloc_gen;
        PIE_HIP_TRY(hipGraphLaunch(s->batch_graph, st));
        return PIE_OK;
    }
    const int rc = decode_batch(d, tokens, context_lens, slabs, (int)n_pages, block_tables, max_blocks, B, logits, logprobs, next_tokens, st);
    d->prefill->warm_key = key, d->prefill->warm_gen = d->prefill->alloc_gen;
    return rc;
}

int prefill_batched(pie_decoder *d, const int32_t *ids, const void *embeds, int L, void *logits_all, hipStream_t st) {
    const int rep = d->cfg.n_heads / d->cfg.n_kv_heads;
    PIE_REQUIRE(rep >= 1 && rep <= 8, PIE_E_SHAPE, "prefill: n_heads / n_kv_heads must be between 1 and 8");
    return d->cfg.dtype == PIE_BF16 ? prefill_t<BF16>(d, ids, embeds, L, logits_all, st) : prefill_t<F16>(d, ids, embeds, L, logits_all, st);
}
