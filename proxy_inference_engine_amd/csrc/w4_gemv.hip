// w4_gemv.hip -- int4 group-64 weights: quantise, repack into the W4S streaming layout, embedding
// row dequant, and the op-level entry point of the streaming GEMV (pie_qgemv_w4g64).
#include <cstdlib>

#include "w4_gemv.hpp"

// ---------------------------------------------------------------- mx.quantize / mx.dequantize
// One thread per group of 64 (load-time utility, not on the decode path).  Same operation order as the
// published algorithm with IEEE fp32 division, so codes/scales/biases are
// bit-identical to the oracle's.  Built with -ffp-contract=off.
template <class T, int BITS>
__global__ void k_quantize_w4g64(const u16 *w, int N, int K, u32 *codes, u16 *scales, u16 *biases) {
    constexpr float LEVELS = (float)((1 << BITS) - 1);
    constexpr int PER_WORD = BITS == 6 ? 5 : 32 / BITS, WORDS = BITS == 6 ? 12 : 64 / PER_WORD;  // 6 bits: MLX's bit stream, 64 codes in 12 words (below)
    const int G = K >> 6;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)N * G) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(w + gid * 64);
    float v[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint4 q = src[i];
        const u32 qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[8 * i + 2 * j] = lo_f32<T>(qq[j]);
            v[8 * i + 2 * j + 1] = hi_f32<T>(qq[j]);
        }
    }
    float w_max = v[0], w_min = v[0];
#pragma unroll
    for (int i = 1; i < 64; ++i) {
        w_max = v[i] > w_max ? v[i] : w_max;
        w_min = v[i] < w_min ? v[i] : w_min;
    }
    const bool side = fabsf(w_min) > fabsf(w_max);
    float scale = fmaxf(__fdiv_rn(__fsub_rn(w_max, w_min), LEVELS), 1e-7f);
    scale = side ? scale : -scale;
    const float edge = side ? w_min : w_max;
    const float q0 = rintf(__fdiv_rn(edge, scale));
    const bool at_zero = q0 == 0.0f;
    scale = at_zero ? scale : __fdiv_rn(edge, q0);
    const float bias = at_zero ? 0.0f : edge;
    scales[gid] = T::from_f32(scale);
    biases[gid] = T::from_f32(bias);
    u32 *dst = codes + gid * WORDS;
    if constexpr (BITS == 6) {  // code k of the row at bits [6k, 6k+6): a group of 64 fills 12 words exactly
        u32 words[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            float c = rintf(__fdiv_rn(__fsub_rn(v[i], bias), scale));
            c = c < 0.0f ? 0.0f : (c > LEVELS ? LEVELS : c);
            const u32 q = (u32)c;
            words[(6 * i) >> 5] |= q << ((6 * i) & 31);
            if (((6 * i) & 31) > 26) words[((6 * i) >> 5) + 1] |= q >> (32 - ((6 * i) & 31));
        }
#pragma unroll
        for (int wd = 0; wd < 12; ++wd) dst[wd] = words[wd];
        return;
    }
#pragma unroll
    for (int wd = 0; wd < WORDS; ++wd) {
        u32 word = 0;
#pragma unroll
        for (int j = 0; j < PER_WORD; ++j) {
            float c = rintf(__fdiv_rn(__fsub_rn(v[PER_WORD * wd + j], bias), scale));
            c = c < 0.0f ? 0.0f : (c > LEVELS ? LEVELS : c);
            word |= ((u32)c) << (BITS * j);
        }
        dst[wd] = word;
    }
}

// out[n,k] = T(scale*q + bias): separate multiply and add roundings, like the oracle (dequant_word_u4, w4_gemv.hpp).
template <class T>
__device__ __forceinline__ void dequant_word(u32 word, float s, float b, u16 *out8) {
    *reinterpret_cast<uint4 *>(out8) = dequant_word_u4<T>(word, s, b);
}

// 8-bit codes: 4 per word (byte i = code 4w + i), 16 words per group of 64.
template <class T>
__device__ __forceinline__ void dequant_word8(u32 word, float s, float b, u16 *out4) {
    const float q0 = __fadd_rn(__fmul_rn(s, (float)(word & 0xFFu)), b), q1 = __fadd_rn(__fmul_rn(s, (float)((word >> 8) & 0xFFu)), b);
    const float q2 = __fadd_rn(__fmul_rn(s, (float)((word >> 16) & 0xFFu)), b), q3 = __fadd_rn(__fmul_rn(s, (float)(word >> 24)), b);
    *reinterpret_cast<uint2 *>(out4) = make_uint2(pack2<T>(q0, q1), pack2<T>(q2, q3));
}

template <class T, int BITS>
__global__ void k_dequantize_w4g64(const u32 *codes, const u16 *scales, const u16 *biases, size_t n_words, u16 *out) {
    const size_t wid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wid >= n_words) return;
    if (BITS == 4) dequant_word<T>(codes[wid], T::to_f32(scales[wid >> 3]), T::to_f32(biases[wid >> 3]), out + wid * 8);
    else dequant_word8<T>(codes[wid], T::to_f32(scales[wid >> 4]), T::to_f32(biases[wid >> 4]), out + wid * 4);
}

// nn.QuantizedEmbedding.__call__ (language.py:176): out[l,:] = dequantize(row ids[l]).
template <class T, int BITS, int GROUP = 64>  // GROUP = 32: MLX group_size 32 (4-bit codes only)
__global__ void k_embedding_w4g64(const int *ids, const u32 *codes, const u16 *scales, const u16 *biases, int V, int H,
                                  u16 *out, const float *freqs, const DecState *state, float *rope_cs, int half) {
    // decoder only: the step's RoPE table (cos, sin of pos / freqs[i], llama/utils.py:42-50) is computed once here
    // and read by the 32 QKV epilogues instead of one full-precision sincosf per row pair and layer
    if (rope_cs && blockIdx.x == 0 && (int)threadIdx.x < half) {
        const float theta = (float)state->pos * (1.0f / freqs[threadIdx.x]);
        float sn, cs;
        sincosf(theta, &sn, &cs);
        rope_cs[2 * threadIdx.x] = cs, rope_cs[2 * threadIdx.x + 1] = sn;
    }
    const int l = blockIdx.x;
    int id = ids[l];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const int words = BITS == 4 ? H >> 3 : H >> 2;
    const u32 *row = codes + (size_t)id * words;
    const u16 *srow = scales + (size_t)id * (H / GROUP), *brow = biases + (size_t)id * (H / GROUP);
    constexpr int WSH = GROUP == 32 ? 2 : 3;  // 4-bit code words per group: 4 or 8
    for (int wd = threadIdx.x; wd < words; wd += blockDim.x) {
        if (BITS == 4) dequant_word<T>(row[wd], T::to_f32(srow[wd >> WSH]), T::to_f32(brow[wd >> WSH]), out + (size_t)l * H + (size_t)wd * 8);
        else dequant_word8<T>(row[wd], T::to_f32(srow[wd >> (WSH + 1)]), T::to_f32(brow[wd >> (WSH + 1)]), out + (size_t)l * H + (size_t)wd * 4);
    }
}

// ---------------------------------------------------------------- W4S repack
// One thread per output dword.  Unit layout (2304 B): dwords [0,256) code piece 0 (lane l -> dwords 4l..4l+3),
// [256,512) code piece 1, [512,576) {scale | bias<<16} per lane.  Lane l: row = row_map[2*pair + (l>>5)],
// group = 32*slice + (l&31); piece j, dword t = source word 8*group + 4j + t with its nibbles reordered so that
// (w >> 4i) & 0x000F000F yields codes (2i, 2i+1) in the two 16-bit halves.
__global__ void k_repack_w4s(const u32 *codes, const u16 *scales, const u16 *biases, int N_src, int K, const int *row_map,
                             int n_pairs, int ns, u32 *packed) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_pairs * ns * 576;
    if (idx >= total) return;
    const size_t unit = idx / 576;
    const int dw = (int)(idx % 576);
    const int pair = (int)(unit / ns), s = (int)(unit % ns);
    int lane, j = 0, t = 0;
    const bool is_sb = dw >= 512;
    if (is_sb) {
        lane = dw - 512;
    } else {
        j = dw >> 8;
        lane = (dw & 255) >> 2;
        t = dw & 3;
    }
    const int prow = 2 * pair + (lane >> 5);
    const int row = row_map ? row_map[prow] : prow;
    const int g = 32 * s + (lane & 31);
    const int G = K >> 6;
    u32 out = 0;
    if (g < G && row >= 0 && row < N_src) {
        if (is_sb) {
            out = (u32)scales[(size_t)row * G + g] | ((u32)biases[(size_t)row * G + g] << 16);
        } else {
            const u32 src = codes[(size_t)row * (K >> 3) + 8 * g + 4 * j + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out |= ((src >> (8 * i)) & 0xFu) << (4 * i);
                out |= ((src >> (8 * i + 4)) & 0xFu) << (16 + 4 * i);
            }
        }
    }
    packed[idx] = out;
}

// MLX int4 g=32 triplet -> W4S32 (common.hpp): the W4S unit with two {scale | bias << 16} words per lane -- dwords [512, 640): lane l ->
// 512 + 2l (group 2g, its first code piece) and 512 + 2l + 1 (group 2g + 1, its second), g = the lane's 64-wide column block.
__global__ void k_repack_w4s32(const u32 *codes, const u16 *scales, const u16 *biases, int N_src, int K, const int *row_map, int n_pairs, int ns,
                               u32 *packed) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_pairs * ns * 640;
    if (idx >= total) return;
    const size_t unit = idx / 640;
    const int dw = (int)(idx % 640);
    const int pair = (int)(unit / ns), s = (int)(unit % ns);
    const bool is_sb = dw >= 512;
    const int lane = is_sb ? (dw - 512) >> 1 : (dw & 255) >> 2, j = is_sb ? (dw & 1) : dw >> 8, t = dw & 3;
    const int prow = 2 * pair + (lane >> 5);
    const int row = row_map ? row_map[prow] : prow;
    const int g = 32 * s + (lane & 31), G = K >> 6;
    u32 out = 0;
    if (g < G && row >= 0 && row < N_src) {
        if (is_sb) {
            out = (u32)scales[(size_t)row * 2 * G + 2 * g + j] | ((u32)biases[(size_t)row * 2 * G + 2 * g + j] << 16);
        } else {
            const u32 src = codes[(size_t)row * (K >> 3) + 8 * g + 4 * j + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out |= ((src >> (8 * i)) & 0xFu) << (4 * i);
                out |= ((src >> (8 * i + 4)) & 0xFu) << (16 + 4 * i);
            }
        }
    }
    packed[idx] = out;
}

// MLX int8 g=64 triplet -> W8S (common.hpp).  One thread per output dword: dwords [0,1024) = four code pieces (piece j: lane l ->
// dwords 256 j + 4 l .. + 3), [1024,1088) = {scale | bias << 16} per lane.  Piece j, dword t of lane l = source word
// 16*group + 4j + t with its bytes reordered (c0, c2, c1, c3).
__global__ void k_repack_w8s(const u32 *codes, const u16 *scales, const u16 *biases, int N_src, int K, const int *row_map, int n_pairs,
                             int ns, u32 *packed) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_pairs * ns * 1088;
    if (idx >= total) return;
    const size_t unit = idx / 1088;
    const int dw = (int)(idx % 1088);
    const int pair = (int)(unit / ns), s = (int)(unit % ns);
    const bool is_sb = dw >= 1024;
    const int lane = is_sb ? dw - 1024 : (dw & 255) >> 2, j = dw >> 8, t = dw & 3;
    const int prow = 2 * pair + (lane >> 5);
    const int row = row_map ? row_map[prow] : prow;
    const int g = 32 * s + (lane & 31), G = K >> 6;
    u32 out = 0;
    if (g < G && row >= 0 && row < N_src) {
        if (is_sb) {
            out = (u32)scales[(size_t)row * G + g] | ((u32)biases[(size_t)row * G + g] << 16);
        } else {
            const u32 src = codes[(size_t)row * (K >> 2) + 16 * g + 4 * j + t];
            out = (src & 0xFF0000FFu) | ((src & 0x0000FF00u) << 8) | ((src & 0x00FF0000u) >> 8);
        }
    }
    packed[idx] = out;
}

// MLX int2 g=64 triplet -> W2S (common.hpp).  One thread per output dword: dwords [0,256) = the code piece (lane l -> dwords 4 l .. 4 l + 3),
// [256,320) = {scale | bias << 16} per lane.  Dword t of lane l = source word 4*group + t (MLX: code k of a word at bits [2k, 2k+2)) with its
// even codes gathered in the low half and its odd codes in the high half.
__global__ void k_repack_w2s(const u32 *codes, const u16 *scales, const u16 *biases, int N_src, int K, const int *row_map, int n_pairs, int ns,
                             u32 *packed) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_pairs * ns * 320;
    if (idx >= total) return;
    const size_t unit = idx / 320;
    const int dw = (int)(idx % 320);
    const int pair = (int)(unit / ns), s = (int)(unit % ns);
    const bool is_sb = dw >= 256;
    const int lane = is_sb ? dw - 256 : dw >> 2, t = dw & 3;
    const int prow = 2 * pair + (lane >> 5);
    const int row = row_map ? row_map[prow] : prow;
    const int g = 32 * s + (lane & 31), G = K >> 6;
    u32 out = 0;
    if (g < G && row >= 0 && row < N_src) {
        if (is_sb) {
            out = (u32)scales[(size_t)row * G + g] | ((u32)biases[(size_t)row * G + g] << 16);
        } else {
            const u32 src = codes[(size_t)row * (K >> 4) + 4 * g + t];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                out |= ((src >> (4 * j)) & 0x3u) << (2 * j);
                out |= ((src >> (4 * j + 2)) & 0x3u) << (16 + 2 * j);
            }
        }
    }
    packed[idx] = out;
}

// MLX int6 g=64 triplet -> W6S (common.hpp).  One thread per output dword: dwords [0,512) = the low-nibble plane in the W4S order (piece j: lane l ->
// dwords 256 j + 4 l .. + 3), [512,768) = the high-two-bit plane in the W2S order (lane l -> dwords 512 + 4 l .. + 3), [768,832) = {scale | bias << 16}.
// MLX packs 6-bit codes as a little-endian bit stream (code k of a row at bits [6k, 6k+6)): read through a two-word window.
__device__ __forceinline__ u32 mlx_code6(const u32 *row, int k) {
    const int bit = 6 * k, wd = bit >> 5, sh = bit & 31;
    u32 v = row[wd] >> sh;
    if (sh > 26) v |= row[wd + 1] << (32 - sh);
    return v & 63u;
}
__global__ void k_repack_w6s(const u32 *codes, const u16 *scales, const u16 *biases, int N_src, int K, const int *row_map, int n_pairs, int ns,
                             u32 *packed) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_pairs * ns * 832;
    if (idx >= total) return;
    const size_t unit = idx / 832;
    const int dw = (int)(idx % 832);
    const int pair = (int)(unit / ns), s = (int)(unit % ns);
    const bool is_sb = dw >= 768, is_hi = dw >= 512 && !is_sb;
    const int lane = is_sb ? dw - 768 : (dw & 255) >> 2, t = dw & 3, j = dw >> 8;
    const int prow = 2 * pair + (lane >> 5);
    const int row = row_map ? row_map[prow] : prow;
    const int g = 32 * s + (lane & 31), G = K >> 6;
    u32 out = 0;
    if (g < G && row >= 0 && row < N_src) {
        if (is_sb) {
            out = (u32)scales[(size_t)row * G + g] | ((u32)biases[(size_t)row * G + g] << 16);
        } else {
            const u32 *src = codes + (size_t)row * (3 * (K >> 4));
            if (is_hi) {
#pragma unroll
                for (int c = 0; c < 16; ++c) out |= (mlx_code6(src, 64 * g + 16 * t + c) >> 4) << (2 * (c >> 1) + 16 * (c & 1));
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) out |= (mlx_code6(src, 64 * g + 8 * (4 * j + t) + c) & 15u) << (4 * (c >> 1) + 16 * (c & 1));
            }
        }
    }
    packed[idx] = out;
}

// MLX int8 g=32 triplet -> W8S32 (common.hpp): the W8S unit with two {scale | bias << 16} words per lane -- dwords [1024, 1152): lane l ->
// 1024 + 2l (group 2g: code pieces 0, 1) and 1024 + 2l + 1 (group 2g + 1: pieces 2, 3).
__global__ void k_repack_w8s32(const u32 *codes, const u16 *scales, const u16 *biases, int N_src, int K, const int *row_map, int n_pairs, int ns,
                               u32 *packed) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_pairs * ns * 1152;
    if (idx >= total) return;
    const size_t unit = idx / 1152;
    const int dw = (int)(idx % 1152);
    const int pair = (int)(unit / ns), s = (int)(unit % ns);
    const bool is_sb = dw >= 1024;
    const int lane = is_sb ? (dw - 1024) >> 1 : (dw & 255) >> 2, j = is_sb ? (dw & 1) : dw >> 8, t = dw & 3;
    const int prow = 2 * pair + (lane >> 5);
    const int row = row_map ? row_map[prow] : prow;
    const int g = 32 * s + (lane & 31), G = K >> 6;
    u32 out = 0;
    if (g < G && row >= 0 && row < N_src) {
        if (is_sb) {
            out = (u32)scales[(size_t)row * 2 * G + 2 * g + j] | ((u32)biases[(size_t)row * 2 * G + 2 * g + j] << 16);
        } else {
            const u32 src = codes[(size_t)row * (K >> 2) + 16 * g + 4 * j + t];
            out = (src & 0xFF0000FFu) | ((src & 0x0000FF00u) << 8) | ((src & 0x00FF0000u) >> 8);
        }
    }
    packed[idx] = out;
}

// Dense 16-bit weights [N_src, K] (nn.Linear) -> W16S.  One thread per 16-byte piece: unit (pair, slice), piece j, lane l:
// row = row_map[2*pair + (l>>5)], elements 512*slice + 16*(l&31) + 8*j .. +8 (zero past K or for an unmapped row).
__global__ void k_repack_w16s(const u16 *w, int N_src, int K, const int *row_map, int n_pairs, int ns, uint4 *packed) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_pairs * ns * 128;
    if (idx >= total) return;
    const size_t unit = idx >> 7;
    const int j = (int)((idx >> 6) & 1), lane = (int)(idx & 63);
    const int pair = (int)(unit / ns), s = (int)(unit % ns);
    const int prow = 2 * pair + (lane >> 5);
    const int row = row_map ? row_map[prow] : prow;
    const int k0 = W16S_SLICE_K * s + 16 * (lane & 31) + 8 * j;
    uint4 out = make_uint4(0, 0, 0, 0);
    if (row >= 0 && row < N_src && k0 < K) out = *reinterpret_cast<const uint4 *>(w + (size_t)row * K + k0);  // K % 8 == 0
    packed[idx] = out;
}
// W16S -> row-major [N_packed, K] in the PACKED row order (prefill GEMMs read plain matrices).
__global__ void k_unpack_w16s(const uint4 *packed, int N, int K, int ns, u16 *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte piece of the output
    const int ppr = K >> 3;
    if (idx >= (size_t)N * ppr) return;
    const int r = (int)(idx / ppr), pc = (int)(idx % ppr), k0 = pc * 8;
    const int s = k0 / W16S_SLICE_K, c = (k0 % W16S_SLICE_K) >> 4, j = (k0 >> 3) & 1;
    reinterpret_cast<uint4 *>(out)[idx] = packed[(((size_t)(r >> 1) * ns + s) << 7) + (j << 6) + (r & 1) * 32 + c];
}
int unpack_w16s_launch(const void *packed, int N, int K, void *out, hipStream_t st) {
    const size_t n = (size_t)N * (K >> 3);
    hipLaunchKernelGGL(k_unpack_w16s, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint4 *)packed, N, K, w16s_slices(K), (u16 *)out);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}
// nn.Embedding (language.py:176) for dense checkpoints: row gather.
__global__ void k_embedding_dense(const int *ids, const uint4 *table, int V, int H, uint4 *out) {
    int id = ids[blockIdx.x];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    for (int i = threadIdx.x; i < (H >> 3); i += blockDim.x) out[(size_t)blockIdx.x * (H >> 3) + i] = table[(size_t)id * (H >> 3) + i];
}

// ---------------------------------------------------------------- launch geometry
template <class T, int NPT, int FMT>
static int launch_n(int pro, int epi, const GemvArgs &a, dim3 grid, unsigned lds, hipStream_t st) {
    const dim3 block(GEMV_WAVES * 64);
    if (pro == PRO_NONE && epi == EPI_PARTIAL_F32) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_NONE, EPI_PARTIAL_F32, NPT, 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_NONE && epi == EPI_STORE) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_NONE, EPI_STORE, NPT, 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_NONE && epi == EPI_RESIDUAL) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_NONE, EPI_RESIDUAL, NPT, 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_ATTN && epi == EPI_RESIDUAL) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_ATTN, EPI_RESIDUAL, (NPT > 2 ? 2 : NPT), 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_ATTN && epi == EPI_PARTIAL_F32) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_ATTN, EPI_PARTIAL_F32, (NPT > 2 ? 2 : NPT), 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_NONE && epi == EPI_TP_PUSH) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_NONE, EPI_TP_PUSH, NPT, 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_ATTN && epi == EPI_TP_PUSH) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_ATTN, EPI_TP_PUSH, (NPT > 2 ? 2 : NPT), 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_RMSNORM && epi == EPI_ROPE_KV) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_RMSNORM, EPI_ROPE_KV, NPT, 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_EMBED && epi == EPI_ROPE_KV) {
        if constexpr (FMT == FMT_W4S) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_EMBED, EPI_ROPE_KV, NPT, 0, FMT>), grid, block, lds, st, a);
        else return pie::fail(PIE_E_ARG, "w4s_gemv: the embedding prologue is built for int4 q|k|v weights");
    }
    else if (pro == PRO_RMSNORM && epi == EPI_SWIGLU) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_RMSNORM, EPI_SWIGLU, NPT, 0, FMT>), grid, block, lds, st, a);
    else if (pro == PRO_RMSNORM && epi == EPI_LOGITS) hipLaunchKernelGGL((k_w4s_gemv<T, PRO_RMSNORM, EPI_LOGITS, NPT, 0, FMT>), grid, block, lds, st, a);
    else return pie::fail(PIE_E_ARG, "w4s_gemv: unsupported prologue/epilogue combination");
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

template <class T, int FMT>
static int launch_f(int pro, int epi, const GemvArgs &a, dim3 grid, unsigned lds, hipStream_t st) {
    const int npt = ((a.K >> 3) + GEMV_WAVES * 64 - 1) / (GEMV_WAVES * 64);  // activation pieces per staging thread
    if (npt <= 1) return launch_n<T, 1, FMT>(pro, epi, a, grid, lds, st);
    if (npt <= 2) return launch_n<T, 2, FMT>(pro, epi, a, grid, lds, st);
    if (npt <= 4) return launch_n<T, 4, FMT>(pro, epi, a, grid, lds, st);
    return launch_n<T, 8, FMT>(pro, epi, a, grid, lds, st);
}
template <class T>
static int launch_t(int pro, int epi, const GemvArgs &a, dim3 grid, unsigned lds, hipStream_t st) {
    if (a.fmt == FMT_W16S) return launch_f<T, FMT_W16S>(pro, epi, a, grid, lds, st);
    if (a.fmt == FMT_W8S) return launch_f<T, FMT_W8S>(pro, epi, a, grid, lds, st);
    if (a.fmt == FMT_W4S32) return launch_f<T, FMT_W4S32>(pro, epi, a, grid, lds, st);
    if (a.fmt == FMT_W8S32) return launch_f<T, FMT_W8S32>(pro, epi, a, grid, lds, st);
    if (a.fmt == FMT_W2S) return launch_f<T, FMT_W2S>(pro, epi, a, grid, lds, st);
    if (a.fmt == FMT_W6S) return launch_f<T, FMT_W6S>(pro, epi, a, grid, lds, st);
    return launch_f<T, FMT_W4S>(pro, epi, a, grid, lds, st);
}

// Persistent grid: one wave per row pair until the chip is full (16 waves per CU), then longer runs per wave.
int w4s_gemv_waves(int N, int K) {
    (void)K;
    constexpr int max_waves = GEMV_MAX_WAVES;
    const int n_pairs = N / 2;
    int waves = n_pairs < max_waves ? n_pairs : max_waves;
    const int need = (n_pairs + GEMV_MAX_RUN - 1) / GEMV_MAX_RUN;  // a wave's run must fit its epilogue lanes
    return waves > need ? waves : need;
}

int w4s_gemv_launch(int dtype, int pro, int epi, GemvArgs &a, int M, hipStream_t stream) {
    PIE_REQUIRE(a.K % 64 == 0 && a.K > 0, PIE_E_SHAPE, "w4s_gemv: K must be a positive multiple of 64");
    PIE_REQUIRE(a.N % 2 == 0 && a.N > 0, PIE_E_SHAPE, "w4s_gemv: N must be even");
    PIE_REQUIRE(a.K <= 32768, PIE_E_SHAPE, "w4s_gemv: K > 32768 not supported");
    PIE_REQUIRE(a.fmt >= FMT_W4S && a.fmt <= FMT_W6S, PIE_E_ARG, "w4s_gemv: unknown weight format");
    PIE_REQUIRE(pro != PRO_ATTN || (a.splits >= 1 && a.splits <= GEMV_ATTN_SPLITS && a.K <= 2 * 8 * GEMV_WAVES * 64 && a.head_dim % 8 == 0), PIE_E_SHAPE,
                "w4s_gemv: attention-merge prologue supports <= 4 splits and n_heads*head_dim <= 8192");
    // Every pointer the chosen prologue / epilogue dereferences, checked HERE so that a null can never reach a kernel (a dense
    // checkpoint legitimately leaves embed scales / biases and rope_cs null; see DESIGN.md "the 00:40 fault").
    PIE_REQUIRE(a.w, PIE_E_ARG, "w4s_gemv: null weight stream");
    PIE_REQUIRE(pro == PRO_ATTN ? (a.part_acc && a.part_ml && a.state) : (pro == PRO_EMBED || a.x != nullptr), PIE_E_ARG, "w4s_gemv: null activation input");
    PIE_REQUIRE((pro != PRO_RMSNORM && pro != PRO_EMBED) || a.norm_w, PIE_E_ARG, "w4s_gemv: RMSNorm prologue without a norm weight");
    PIE_REQUIRE(pro != PRO_EMBED || (a.emb_codes && a.emb_scales && a.emb_biases && a.token && a.emb_vocab > 0 && a.h_out && a.rope_cs_out && a.state && a.freqs && M == 1 &&
                                     epi == EPI_ROPE_KV && !a.rope_cs),
                PIE_E_ARG, "w4s_gemv: the embedding prologue needs the int4 triplet, the token, the row / RoPE-table destinations and the q|k|v epilogue");
    PIE_REQUIRE((epi != EPI_STORE && epi != EPI_LOGITS && epi != EPI_SWIGLU) || a.y, PIE_E_ARG, "w4s_gemv: null output");
    PIE_REQUIRE(epi != EPI_LOGITS || a.stats, PIE_E_ARG, "w4s_gemv: EPI_LOGITS without a partials buffer");
    PIE_REQUIRE(epi != EPI_PARTIAL_F32 || a.y32, PIE_E_ARG, "w4s_gemv: EPI_PARTIAL_F32 without an fp32 output");
    PIE_REQUIRE(epi != EPI_TP_PUSH || (a.tp_peers && a.tp_epoch && a.tp_world >= 1 && a.tp_world <= GEMV_TP_MAX_WORLD && a.tp_rank >= 0 && a.tp_rank < a.tp_world &&
                                       (unsigned)a.N <= a.tp_stride && M == 1),
                PIE_E_ARG, "w4s_gemv: EPI_TP_PUSH needs the communicator's peer table, epoch word, rank / world and a slot that holds N elements");
    PIE_REQUIRE(epi != EPI_RESIDUAL || a.resid, PIE_E_ARG, "w4s_gemv: EPI_RESIDUAL without a residual stream");
    PIE_REQUIRE(epi != EPI_ROPE_KV || (a.state && a.q_out && a.kv_table && (a.rope_cs || a.freqs)), PIE_E_ARG,
                "w4s_gemv: EPI_ROPE_KV needs state, q_out, kv_table and rope_cs or freqs");
    a.n_slices = a.fmt == FMT_W16S ? w16s_slices(a.K) : w4s_slices(a.K);
    a.n_pairs = a.N / 2;
    a.n_waves = w4s_gemv_waves(a.N, a.K);
    const size_t unit_bytes = fmt_unit_bytes(a.fmt);
    PIE_REQUIRE((size_t)a.n_pairs * a.n_slices * unit_bytes < ((size_t)1 << 32) - 8192, PIE_E_SHAPE,
                "w4s_gemv: one matrix must stay below 4 GiB (32-bit buffer offsets)");
    const unsigned lds = (unsigned)gemv_lds(a.K, a.fmt == FMT_W4S32 || a.fmt == FMT_W8S32).total;
    PIE_REQUIRE(lds <= 65536u, PIE_E_SHAPE, "w4s_gemv: activation vector does not fit the 64 KB LDS image");
    dim3 grid((a.n_waves + GEMV_WAVES - 1) / GEMV_WAVES, M);
    a.full_rounds = a.n_pairs / a.n_waves, a.rem_pairs = a.n_pairs - a.full_rounds * a.n_waves;
    a.n_blocks = a.n_waves % GEMV_WAVES == 0 ? (int)grid.x : 0;  // whole workgroups only: every wave that gets a leftover pair also has a slot in stats[]
    if (a.fuse) {  // the q|k|v launch with the step's attention behind an XCD-local seam (w4_gemv.hpp, FUSE)
        PIE_REQUIRE(epi == EPI_ROPE_KV && (pro == PRO_RMSNORM || (pro == PRO_EMBED && a.fmt == FMT_W4S)) && M == 1 && a.seam && a.block_table == a.attn.block_table, PIE_E_ARG,
                    "w4s_gemv: the fused attention follows a q|k|v launch on the same (contiguous or T-page) cache");
        PIE_REQUIRE(a.n_heads == 32 && a.n_kv_heads == 8 && a.head_dim == 128 && grid.x == 256 && a.n_waves == 2048 && a.full_rounds == 1 && a.rem_pairs == 1024 &&
                        a.attn.splits >= 1 && a.attn.splits <= 4 && a.attn.Hq == 32 && a.attn.Hkv == 8 && a.attn.part_acc && a.attn.part_ml && a.attn.q,
                    PIE_E_SHAPE, "w4s_gemv: the fused attention is built for 32 query heads and 8 kv heads of 128 (kv-group = XCD)");
        PIE_REQUIRE(a.K <= 8 * GEMV_WAVES * 64, PIE_E_SHAPE, "w4s_gemv: the fused attention takes hidden sizes up to 4096");
#define PIE_FUSE_GO(TT, PRO_, FMT_) hipLaunchKernelGGL((k_w4s_gemv<TT, PRO_, EPI_ROPE_KV, 1, 0, FMT_, 1>), grid, dim3(GEMV_WAVES * 64), lds, stream, a)
#define PIE_FUSE_FMT(TT)                                                  \
    if (pro == PRO_EMBED) PIE_FUSE_GO(TT, PRO_EMBED, FMT_W4S);            \
    else if (a.fmt == FMT_W16S) PIE_FUSE_GO(TT, PRO_RMSNORM, FMT_W16S);   \
    else if (a.fmt == FMT_W8S) PIE_FUSE_GO(TT, PRO_RMSNORM, FMT_W8S);     \
    else if (a.fmt == FMT_W4S32) PIE_FUSE_GO(TT, PRO_RMSNORM, FMT_W4S32); \
    else if (a.fmt == FMT_W8S32) PIE_FUSE_GO(TT, PRO_RMSNORM, FMT_W8S32); \
    else if (a.fmt == FMT_W2S) PIE_FUSE_GO(TT, PRO_RMSNORM, FMT_W2S);     \
    else if (a.fmt == FMT_W6S) PIE_FUSE_GO(TT, PRO_RMSNORM, FMT_W6S);     \
    else PIE_FUSE_GO(TT, PRO_RMSNORM, FMT_W4S)
        if (dtype == PIE_BF16) { PIE_FUSE_FMT(BF16); }
        else if (dtype == PIE_F16) { PIE_FUSE_FMT(F16); }
        else return pie::fail(PIE_E_ARG, "w4s_gemv: dtype must be PIE_BF16 or PIE_F16");
#undef PIE_FUSE_FMT
#undef PIE_FUSE_GO
        PIE_LAUNCH_CHECK();
        return PIE_OK;
    }
    if (dtype == PIE_BF16) return launch_t<BF16>(pro, epi, a, grid, lds, stream);
    if (dtype == PIE_F16) return launch_t<F16>(pro, epi, a, grid, lds, stream);
    return pie::fail(PIE_E_ARG, "w4s_gemv: dtype must be PIE_BF16 or PIE_F16");
}

template <class T, int MR, int PRO, int EPI>
static int rows_launch_t(const GemvRowsArgs &a, dim3 grid, unsigned lds, hipStream_t st) {
    // per call: the attribute belongs to the (function, device) pair, a process-wide "done" flag would leave a second device without it
    // (ADVICE r3); the call is cheap and legal during stream capture
    PIE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_w4s_gemv_rows<T, MR, PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    hipLaunchKernelGGL((k_w4s_gemv_rows<T, MR, PRO, EPI>), grid, dim3(GEMV_WAVES * 64), lds, st, a);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}
template <class T, int PRO, int EPI>
static int rows_launch_mr(const GemvRowsArgs &a, int mr, dim3 grid, unsigned lds, hipStream_t st) {
    switch (mr) {
        case 1: return rows_launch_t<T, 1, PRO, EPI>(a, grid, lds, st);
        case 2: return rows_launch_t<T, 2, PRO, EPI>(a, grid, lds, st);
        case 3: return rows_launch_t<T, 3, PRO, EPI>(a, grid, lds, st);
        case 4: return rows_launch_t<T, 4, PRO, EPI>(a, grid, lds, st);
        default: return rows_launch_t<T, 5, PRO, EPI>(a, grid, lds, st);
    }
}
template <class T>
static int rows_launch_pe(int pro, int epi, const GemvRowsArgs &a, int mr, dim3 grid, unsigned lds, hipStream_t st) {
    if (pro == PRO_NONE && epi == EPI_STORE) return rows_launch_mr<T, PRO_NONE, EPI_STORE>(a, mr, grid, lds, st);
    if (pro == PRO_NONE && epi == EPI_RESIDUAL) return rows_launch_mr<T, PRO_NONE, EPI_RESIDUAL>(a, mr, grid, lds, st);
    if (pro == PRO_RMSNORM && epi == EPI_SWIGLU) return rows_launch_mr<T, PRO_RMSNORM, EPI_SWIGLU>(a, mr, grid, lds, st);
    if (pro == PRO_RMSNORM && epi == EPI_ROPE_KV) return rows_launch_mr<T, PRO_RMSNORM, EPI_ROPE_KV>(a, mr, grid, lds, st);
    if (pro == PRO_RMSNORM && epi == EPI_LOGITS) return rows_launch_mr<T, PRO_RMSNORM, EPI_LOGITS>(a, mr, grid, lds, st);
    return pie::fail(PIE_E_ARG, "w4s_gemv_rows: prologue / epilogue combination not instantiated");
}

int w4s_gemv_rows_fused_launch(int dtype, int pro, int epi, GemvRowsArgs &a, hipStream_t stream) {
    PIE_REQUIRE(a.w && a.x, PIE_E_ARG, "w4s_gemv_rows: null pointer");
    PIE_REQUIRE(a.K % 64 == 0 && a.K > 0 && a.K <= 32768 && a.N % 2 == 0 && a.N > 0, PIE_E_SHAPE, "w4s_gemv_rows: K must be a multiple of 64 (<= 32768), N even");
    PIE_REQUIRE(a.M >= 1 && a.M <= 65535, PIE_E_SHAPE, "w4s_gemv_rows: M out of range");  // more than GEMV_ROWS_MAX rows: chunks of rows over blockIdx.y
    // every pointer the chosen prologue / epilogue dereferences, checked here so that a null can never reach the kernel
    PIE_REQUIRE(pro != PRO_RMSNORM || (a.norm_w && a.K <= 8192), PIE_E_ARG, "w4s_gemv_rows: the RMSNorm prologue needs its weight and K <= 8192");
    PIE_REQUIRE((epi != EPI_STORE && epi != EPI_LOGITS && epi != EPI_SWIGLU) || a.y, PIE_E_ARG, "w4s_gemv_rows: null output");
    PIE_REQUIRE(epi != EPI_RESIDUAL || a.resid, PIE_E_ARG, "w4s_gemv_rows: EPI_RESIDUAL without a residual stream");
    PIE_REQUIRE(epi != EPI_LOGITS || a.stats, PIE_E_ARG, "w4s_gemv_rows: EPI_LOGITS without a partials buffer");
    PIE_REQUIRE(epi != EPI_ROPE_KV || (a.rope_cs && a.ctx_len && a.block_table && a.slab && a.q_out && a.n_pages > 0 && a.bt_stride > 0 && a.head_dim > 0), PIE_E_ARG,
                "w4s_gemv_rows: EPI_ROPE_KV needs the rows' RoPE table, context lengths, block tables, the slab and q_out");
    a.n_slices = w4s_slices(a.K), a.n_pairs = a.N / 2, a.n_waves = w4s_gemv_waves(a.N, a.K);
    PIE_REQUIRE((size_t)a.n_pairs * a.n_slices * W4S_UNIT_BYTES < ((size_t)1 << 32) - 8192, PIE_E_SHAPE, "w4s_gemv_rows: one matrix must stay below 4 GiB");
    // rows per workgroup: all of them where their LDS images fit, else the fewest equal chunks (K = 14336: 5 rows -> 3 + 2)
    int mr_fit = GEMV_ROWS_MAX;
    while (mr_fit > 1 && gemv_rows_lds_bytes(a.K, mr_fit) > 160u * 1024u) --mr_fit;
    PIE_REQUIRE(gemv_rows_lds_bytes(a.K, mr_fit) <= 160u * 1024u, PIE_E_SHAPE, "w4s_gemv_rows: activation vector does not fit LDS");
    const int chunks = (a.M + mr_fit - 1) / mr_fit, mr = (a.M + chunks - 1) / chunks;
    const unsigned lds = gemv_rows_lds_bytes(a.K, mr);
    const dim3 grid((a.n_waves + GEMV_WAVES - 1) / GEMV_WAVES, chunks);
    a.full_rounds = a.n_pairs / a.n_waves, a.rem_pairs = a.n_pairs - a.full_rounds * a.n_waves;
    a.n_blocks = a.n_waves % GEMV_WAVES == 0 ? (int)grid.x : 0;
    if (dtype == PIE_BF16) return rows_launch_pe<BF16>(pro, epi, a, mr, grid, lds, stream);
    if (dtype == PIE_F16) return rows_launch_pe<F16>(pro, epi, a, mr, grid, lds, stream);
    return pie::fail(PIE_E_ARG, "w4s_gemv_rows: dtype must be PIE_BF16 or PIE_F16");
}

int w4s_gemv_rows_launch(int dtype, const void *packed, int N, int K, const u16 *x, int M, u16 *y, const u16 *lin_bias, hipStream_t stream) {
    PIE_REQUIRE(M >= 1 && M <= 65535, PIE_E_SHAPE, "w4s_gemv_rows: M out of range");
    if (M == 1 || gemv_rows_lds_bytes(K, 1) > 160u * 1024u) {  // one row, or a K whose image does not fit next to the row sums: the batch-1 kernel, one pass per row
        GemvArgs g = {};
        g.w = (const char *)packed, g.K = K, g.N = N, g.x = x, g.y = y, g.lin_bias = lin_bias;
        return w4s_gemv_launch(dtype, PRO_NONE, EPI_STORE, g, M, stream);
    }
    GemvRowsArgs a = {};
    a.w = (const char *)packed, a.K = K, a.N = N, a.M = M, a.x = x, a.y = y, a.lin_bias = lin_bias;
    return w4s_gemv_rows_fused_launch(dtype, PRO_NONE, EPI_STORE, a, stream);
}

// ---------------------------------------------------------------- C ABI
int embedding_launch(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H, int dtype,
                     void *out, const float *freqs, const DecState *state, float *rope_cs, int half, hipStream_t st, int bits);

extern "C" {

size_t pie_w4s_bytes(int N_out, int K) {
    if (N_out <= 0 || K <= 0 || (N_out & 1) || (K & 63)) return 0;
    return (size_t)(N_out / 2) * w4s_slices(K) * W4S_UNIT_BYTES;
}

int pie_quantize_w4g64(const void *w, int N, int K, int dtype, uint32_t *codes, void *scales, void *biases, void *stream) {
    return pie_quantize_g64(w, N, K, 4, dtype, codes, scales, biases, stream);
}

int pie_quantize_g64(const void *w, int N, int K, int bits, int dtype, uint32_t *codes, void *scales, void *biases, void *stream) {
    PIE_REQUIRE(w && codes && scales && biases, PIE_E_ARG, "pie_quantize_w4g64: null pointer");
    PIE_REQUIRE(bits == 2 || bits == 4 || bits == 6 || bits == 8, PIE_E_ARG, "pie_quantize_g64: bits must be 2, 4, 6 or 8");
    PIE_REQUIRE(N > 0 && K > 0 && K % 64 == 0, PIE_E_SHAPE, "pie_quantize_w4g64: K must be a multiple of 64");
    PIE_REQUIRE(pie_aligned(w, 16) && pie_aligned(codes, 16), PIE_E_ALIGN, "pie_quantize_w4g64: 16-byte alignment required");
    const size_t groups = (size_t)N * (K / 64);
    dim3 grid((unsigned)((groups + 127) / 128)), block(128);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PIE_BF16 && bits == 6)
        hipLaunchKernelGGL((k_quantize_w4g64<BF16, 6>), grid, block, 0, st, (const u16 *)w, N, K, codes, (u16 *)scales, (u16 *)biases);
    else if (dtype == PIE_F16 && bits == 6)
        hipLaunchKernelGGL((k_quantize_w4g64<F16, 6>), grid, block, 0, st, (const u16 *)w, N, K, codes, (u16 *)scales, (u16 *)biases);
    else if (dtype == PIE_BF16 && bits == 2)
        hipLaunchKernelGGL((k_quantize_w4g64<BF16, 2>), grid, block, 0, st, (const u16 *)w, N, K, codes, (u16 *)scales, (u16 *)biases);
    else if (dtype == PIE_F16 && bits == 2)
        hipLaunchKernelGGL((k_quantize_w4g64<F16, 2>), grid, block, 0, st, (const u16 *)w, N, K, codes, (u16 *)scales, (u16 *)biases);
    else if (dtype == PIE_BF16 && bits == 4)
        hipLaunchKernelGGL((k_quantize_w4g64<BF16, 4>), grid, block, 0, st, (const u16 *)w, N, K, codes, (u16 *)scales, (u16 *)biases);
    else if (dtype == PIE_F16 && bits == 4)
        hipLaunchKernelGGL((k_quantize_w4g64<F16, 4>), grid, block, 0, st, (const u16 *)w, N, K, codes, (u16 *)scales, (u16 *)biases);
    else if (dtype == PIE_BF16)
        hipLaunchKernelGGL((k_quantize_w4g64<BF16, 8>), grid, block, 0, st, (const u16 *)w, N, K, codes, (u16 *)scales, (u16 *)biases);
    else if (dtype == PIE_F16)
        hipLaunchKernelGGL((k_quantize_w4g64<F16, 8>), grid, block, 0, st, (const u16 *)w, N, K, codes, (u16 *)scales, (u16 *)biases);
    else
        return pie::fail(PIE_E_ARG, "pie_quantize_w4g64: bad dtype");
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_dequantize_w4g64(const uint32_t *codes, const void *scales, const void *biases, int N, int K, int dtype, void *w_out,
                         void *stream) {
    return pie_dequantize_g64(codes, scales, biases, N, K, 4, dtype, w_out, stream);
}

int pie_dequantize_g64(const uint32_t *codes, const void *scales, const void *biases, int N, int K, int bits, int dtype, void *w_out,
                       void *stream) {
    PIE_REQUIRE(codes && scales && biases && w_out, PIE_E_ARG, "pie_dequantize_w4g64: null pointer");
    PIE_REQUIRE(bits == 4 || bits == 8, PIE_E_ARG, "pie_dequantize_g64: bits must be 4 or 8");
    PIE_REQUIRE(N > 0 && K > 0 && K % 64 == 0, PIE_E_SHAPE, "pie_dequantize_w4g64: K must be a multiple of 64");
    PIE_REQUIRE(pie_aligned(w_out, 16), PIE_E_ALIGN, "pie_dequantize_w4g64: output must be 16-byte aligned");
    const size_t n_words = (size_t)N * (K / (32 / bits));
    dim3 grid((unsigned)((n_words + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PIE_BF16 && bits == 4)
        hipLaunchKernelGGL((k_dequantize_w4g64<BF16, 4>), grid, block, 0, st, codes, (const u16 *)scales, (const u16 *)biases, n_words, (u16 *)w_out);
    else if (dtype == PIE_F16 && bits == 4)
        hipLaunchKernelGGL((k_dequantize_w4g64<F16, 4>), grid, block, 0, st, codes, (const u16 *)scales, (const u16 *)biases, n_words, (u16 *)w_out);
    else if (dtype == PIE_BF16)
        hipLaunchKernelGGL((k_dequantize_w4g64<BF16, 8>), grid, block, 0, st, codes, (const u16 *)scales, (const u16 *)biases, n_words, (u16 *)w_out);
    else if (dtype == PIE_F16)
        hipLaunchKernelGGL((k_dequantize_w4g64<F16, 8>), grid, block, 0, st, codes, (const u16 *)scales, (const u16 *)biases, n_words, (u16 *)w_out);
    else
        return pie::fail(PIE_E_ARG, "pie_dequantize_w4g64: bad dtype");
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_repack_w4g64(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map,
                     int N_out, void *packed, void *stream) {
    PIE_REQUIRE(codes && scales && biases && packed, PIE_E_ARG, "pie_repack_w4g64: null pointer");
    PIE_REQUIRE(N_src > 0 && N_out > 0 && (N_out % 2) == 0, PIE_E_SHAPE, "pie_repack_w4g64: N_out must be even");
    PIE_REQUIRE(K > 0 && K % 64 == 0, PIE_E_SHAPE, "pie_repack_w4g64: K must be a multiple of 64");
    PIE_REQUIRE(w4s_slices(K) <= 16, PIE_E_SHAPE, "pie_repack_w4g64: K > 32768 not supported");
    PIE_REQUIRE(pie_aligned(packed, 256), PIE_E_ALIGN, "pie_repack_w4g64: packed must be 256-byte aligned");
    const int n_pairs = N_out / 2, ns = w4s_slices(K);
    const size_t total = (size_t)n_pairs * ns * 576;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_repack_w4s, grid, block, 0, (hipStream_t)stream, codes, (const u16 *)scales, (const u16 *)biases, N_src, K,
                       row_map, n_pairs, ns, (u32 *)packed);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_qgemv_w4g64_f32(const void *x, int M, const void *packed, int N, int K, float *y, int dtype, void *stream) {
    PIE_REQUIRE(x && packed && y, PIE_E_ARG, "pie_qgemv_w4g64_f32: null pointer");
    PIE_REQUIRE(M > 0 && M <= 65535, PIE_E_SHAPE, "pie_qgemv_w4g64_f32: M out of range");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(packed, 16) && pie_aligned(y, 8), PIE_E_ALIGN, "pie_qgemv_w4g64_f32: misaligned pointer");
    GemvArgs a = {};
    a.w = (const char *)packed;
    a.K = K, a.N = N;
    a.x = (const u16 *)x;
    a.y32 = y;
    return w4s_gemv_launch(dtype, PRO_NONE, EPI_PARTIAL_F32, a, M, (hipStream_t)stream);
}

size_t pie_w8s_bytes(int N_out, int K) {
    if (N_out <= 0 || K <= 0 || (N_out & 1) || (K & 63)) return 0;
    return (size_t)(N_out / 2) * w4s_slices(K) * W8S_UNIT_BYTES;
}

int pie_repack_w8g64(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream) {
    PIE_REQUIRE(codes && scales && biases && packed, PIE_E_ARG, "pie_repack_w8g64: null pointer");
    PIE_REQUIRE(N_src > 0 && N_out > 0 && (N_out % 2) == 0, PIE_E_SHAPE, "pie_repack_w8g64: N_out must be even");
    PIE_REQUIRE(K > 0 && K % 64 == 0 && K <= 32768, PIE_E_SHAPE, "pie_repack_w8g64: K must be a multiple of 64, at most 32768");
    PIE_REQUIRE(pie_aligned(packed, 256), PIE_E_ALIGN, "pie_repack_w8g64: packed must be 256-byte aligned");
    const int n_pairs = N_out / 2, ns = w4s_slices(K);
    const size_t total = (size_t)n_pairs * ns * 1088;
    hipLaunchKernelGGL(k_repack_w8s, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codes, (const u16 *)scales,
                       (const u16 *)biases, N_src, K, row_map, n_pairs, ns, (u32 *)packed);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_qgemv_w8g64(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream) {
    PIE_REQUIRE(x && packed && y, PIE_E_ARG, "pie_qgemv_w8g64: null pointer");
    PIE_REQUIRE(M > 0 && M <= 65535, PIE_E_SHAPE, "pie_qgemv_w8g64: M out of range");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(packed, 16) && pie_aligned(y, 4), PIE_E_ALIGN, "pie_qgemv_w8g64: misaligned pointer");
    GemvArgs a = {};
    a.fmt = FMT_W8S;
    a.w = (const char *)packed;
    a.K = K, a.N = N;
    a.x = (const u16 *)x;
    a.y = (u16 *)y;
    a.lin_bias = (const u16 *)lin_bias;
    return w4s_gemv_launch(dtype, PRO_NONE, EPI_STORE, a, M, (hipStream_t)stream);
}

size_t pie_w2s_bytes(int N_out, int K) {
    if (N_out <= 0 || K <= 0 || (N_out & 1) || (K & 63)) return 0;
    return (size_t)(N_out / 2) * w4s_slices(K) * W2S_UNIT_BYTES;
}

int pie_repack_w2g64(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream) {
    PIE_REQUIRE(codes && scales && biases && packed, PIE_E_ARG, "pie_repack_w2g64: null pointer");
    PIE_REQUIRE(N_src > 0 && N_out > 0 && (N_out % 2) == 0, PIE_E_SHAPE, "pie_repack_w2g64: N_out must be even");
    PIE_REQUIRE(K > 0 && K % 64 == 0 && K <= 32768, PIE_E_SHAPE, "pie_repack_w2g64: K must be a multiple of 64, at most 32768");
    PIE_REQUIRE(pie_aligned(packed, 256), PIE_E_ALIGN, "pie_repack_w2g64: packed must be 256-byte aligned");
    const int n_pairs = N_out / 2, ns = w4s_slices(K);
    const size_t total = (size_t)n_pairs * ns * 320;
    hipLaunchKernelGGL(k_repack_w2s, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codes, (const u16 *)scales,
                       (const u16 *)biases, N_src, K, row_map, n_pairs, ns, (u32 *)packed);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_qgemv_w2g64(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream) {
    PIE_REQUIRE(x && packed && y, PIE_E_ARG, "pie_qgemv_w2g64: null pointer");
    PIE_REQUIRE(M > 0 && M <= 65535, PIE_E_SHAPE, "pie_qgemv_w2g64: M out of range");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(packed, 16) && pie_aligned(y, 4), PIE_E_ALIGN, "pie_qgemv_w2g64: misaligned pointer");
    GemvArgs a = {};
    a.fmt = FMT_W2S;
    a.w = (const char *)packed;
    a.K = K, a.N = N;
    a.x = (const u16 *)x;
    a.y = (u16 *)y;
    a.lin_bias = (const u16 *)lin_bias;
    return w4s_gemv_launch(dtype, PRO_NONE, EPI_STORE, a, M, (hipStream_t)stream);
}

size_t pie_w6s_bytes(int N_out, int K) {
    if (N_out <= 0 || K <= 0 || (N_out & 1) || (K & 63)) return 0;
    return (size_t)(N_out / 2) * w4s_slices(K) * W6S_UNIT_BYTES;
}

int pie_repack_w6g64(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream) {
    PIE_REQUIRE(codes && scales && biases && packed, PIE_E_ARG, "pie_repack_w6g64: null pointer");
    PIE_REQUIRE(N_src > 0 && N_out > 0 && (N_out % 2) == 0, PIE_E_SHAPE, "pie_repack_w6g64: N_out must be even");
    PIE_REQUIRE(K > 0 && K % 64 == 0 && K <= 32768, PIE_E_SHAPE, "pie_repack_w6g64: K must be a multiple of 64, at most 32768");
    PIE_REQUIRE(pie_aligned(packed, 256), PIE_E_ALIGN, "pie_repack_w6g64: packed must be 256-byte aligned");
    const int n_pairs = N_out / 2, ns = w4s_slices(K);
    const size_t total = (size_t)n_pairs * ns * 832;
    hipLaunchKernelGGL(k_repack_w6s, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codes, (const u16 *)scales,
                       (const u16 *)biases, N_src, K, row_map, n_pairs, ns, (u32 *)packed);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_qgemv_w6g64(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream) {
    PIE_REQUIRE(x && packed && y, PIE_E_ARG, "pie_qgemv_w6g64: null pointer");
    PIE_REQUIRE(M > 0 && M <= 65535, PIE_E_SHAPE, "pie_qgemv_w6g64: M out of range");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(packed, 16) && pie_aligned(y, 4), PIE_E_ALIGN, "pie_qgemv_w6g64: misaligned pointer");
    GemvArgs a = {};
    a.fmt = FMT_W6S;
    a.w = (const char *)packed;
    a.K = K, a.N = N;
    a.x = (const u16 *)x;
    a.y = (u16 *)y;
    a.lin_bias = (const u16 *)lin_bias;
    return w4s_gemv_launch(dtype, PRO_NONE, EPI_STORE, a, M, (hipStream_t)stream);
}

size_t pie_w4s32_bytes(int N_out, int K) {
    if (N_out <= 0 || K <= 0 || (N_out & 1) || (K & 63)) return 0;
    return (size_t)(N_out / 2) * w4s_slices(K) * W4S32_UNIT_BYTES;
}

int pie_repack_w4g32(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream) {
    PIE_REQUIRE(codes && scales && biases && packed, PIE_E_ARG, "pie_repack_w4g32: null pointer");
    PIE_REQUIRE(N_src > 0 && N_out > 0 && (N_out % 2) == 0, PIE_E_SHAPE, "pie_repack_w4g32: N_out must be even");
    PIE_REQUIRE(K > 0 && K % 64 == 0 && K <= 32768, PIE_E_SHAPE, "pie_repack_w4g32: K must be a multiple of 64, at most 32768");
    PIE_REQUIRE(pie_aligned(packed, 256), PIE_E_ALIGN, "pie_repack_w4g32: packed must be 256-byte aligned");
    const int n_pairs = N_out / 2, ns = w4s_slices(K);
    const size_t total = (size_t)n_pairs * ns * 640;
    hipLaunchKernelGGL(k_repack_w4s32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codes, (const u16 *)scales,
                       (const u16 *)biases, N_src, K, row_map, n_pairs, ns, (u32 *)packed);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_qgemv_w4g32(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream) {
    PIE_REQUIRE(x && packed && y, PIE_E_ARG, "pie_qgemv_w4g32: null pointer");
    PIE_REQUIRE(M > 0 && M <= 65535, PIE_E_SHAPE, "pie_qgemv_w4g32: M out of range");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(packed, 16) && pie_aligned(y, 4), PIE_E_ALIGN, "pie_qgemv_w4g32: misaligned pointer");
    GemvArgs a = {};
    a.fmt = FMT_W4S32;
    a.w = (const char *)packed;
    a.K = K, a.N = N;
    a.x = (const u16 *)x;
    a.y = (u16 *)y;
    a.lin_bias = (const u16 *)lin_bias;
    return w4s_gemv_launch(dtype, PRO_NONE, EPI_STORE, a, M, (hipStream_t)stream);
}

int pie_embedding_g32(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H, int bits, int dtype, void *out,
                      void *stream) {
    PIE_REQUIRE(bits == 4 || bits == 8, PIE_E_ARG, "pie_embedding_g32: bits must be 4 or 8");
    return embedding_launch(ids, L, codes, scales, biases, V, H, dtype, out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, bits == 8 ? PIE_EMBED_W8G32 : PIE_EMBED_W4G32);
}

size_t pie_w8s32_bytes(int N_out, int K) {
    if (N_out <= 0 || K <= 0 || (N_out & 1) || (K & 63)) return 0;
    return (size_t)(N_out / 2) * w4s_slices(K) * W8S32_UNIT_BYTES;
}

int pie_repack_w8g32(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream) {
    PIE_REQUIRE(codes && scales && biases && packed, PIE_E_ARG, "pie_repack_w8g32: null pointer");
    PIE_REQUIRE(N_src > 0 && N_out > 0 && (N_out % 2) == 0, PIE_E_SHAPE, "pie_repack_w8g32: N_out must be even");
    PIE_REQUIRE(K > 0 && K % 64 == 0 && K <= 32768, PIE_E_SHAPE, "pie_repack_w8g32: K must be a multiple of 64, at most 32768");
    PIE_REQUIRE(pie_aligned(packed, 256), PIE_E_ALIGN, "pie_repack_w8g32: packed must be 256-byte aligned");
    const int n_pairs = N_out / 2, ns = w4s_slices(K);
    const size_t total = (size_t)n_pairs * ns * 1152;
    hipLaunchKernelGGL(k_repack_w8s32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codes, (const u16 *)scales,
                       (const u16 *)biases, N_src, K, row_map, n_pairs, ns, (u32 *)packed);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_qgemv_w8g32(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream) {
    PIE_REQUIRE(x && packed && y, PIE_E_ARG, "pie_qgemv_w8g32: null pointer");
    PIE_REQUIRE(M > 0 && M <= 65535, PIE_E_SHAPE, "pie_qgemv_w8g32: M out of range");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(packed, 16) && pie_aligned(y, 4), PIE_E_ALIGN, "pie_qgemv_w8g32: misaligned pointer");
    GemvArgs a = {};
    a.fmt = FMT_W8S32;
    a.w = (const char *)packed;
    a.K = K, a.N = N;
    a.x = (const u16 *)x;
    a.y = (u16 *)y;
    a.lin_bias = (const u16 *)lin_bias;
    return w4s_gemv_launch(dtype, PRO_NONE, EPI_STORE, a, M, (hipStream_t)stream);
}

size_t pie_w16s_bytes(int N_out, int K) {
    if (N_out <= 0 || K <= 0 || (N_out & 1) || (K & 63)) return 0;
    return (size_t)(N_out / 2) * w16s_slices(K) * W16S_UNIT_BYTES;
}

int pie_repack_dense(const void *w, int N_src, int K, const int32_t *row_map, int N_out, void *packed, void *stream) {
    PIE_REQUIRE(w && packed, PIE_E_ARG, "pie_repack_dense: null pointer");
    PIE_REQUIRE(N_src > 0 && N_out > 0 && (N_out % 2) == 0, PIE_E_SHAPE, "pie_repack_dense: N_out must be even");
    PIE_REQUIRE(K > 0 && K % 64 == 0 && K <= 32768, PIE_E_SHAPE, "pie_repack_dense: K must be a multiple of 64, at most 32768");
    PIE_REQUIRE(pie_aligned(packed, 256) && pie_aligned(w, 16), PIE_E_ALIGN, "pie_repack_dense: packed needs 256-byte, w 16-byte alignment");
    const int n_pairs = N_out / 2, ns = w16s_slices(K);
    const size_t total = (size_t)n_pairs * ns * 128;
    hipLaunchKernelGGL(k_repack_w16s, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16 *)w, N_src, K, row_map,
                       n_pairs, ns, (uint4 *)packed);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_gemv_dense(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream) {
    PIE_REQUIRE(x && packed && y, PIE_E_ARG, "pie_gemv_dense: null pointer");
    PIE_REQUIRE(M > 0 && M <= 65535, PIE_E_SHAPE, "pie_gemv_dense: M out of range");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(packed, 16) && pie_aligned(y, 4), PIE_E_ALIGN, "pie_gemv_dense: misaligned pointer");
    GemvArgs a = {};
    a.fmt = FMT_W16S;
    a.w = (const char *)packed;
    a.K = K, a.N = N;
    a.x = (const u16 *)x;
    a.y = (u16 *)y;
    a.lin_bias = (const u16 *)lin_bias;
    return w4s_gemv_launch(dtype, PRO_NONE, EPI_STORE, a, M, (hipStream_t)stream);
}

int pie_embedding_dense(const int32_t *ids, int L, const void *table, int V, int H, int dtype, void *out, void *stream) {
    PIE_REQUIRE(ids && table && out, PIE_E_ARG, "pie_embedding_dense: null pointer");
    PIE_REQUIRE(L > 0 && V > 0 && H > 0 && H % 8 == 0, PIE_E_SHAPE, "pie_embedding_dense: H must be a multiple of 8");
    PIE_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16, PIE_E_ARG, "pie_embedding_dense: bad dtype");
    PIE_REQUIRE(pie_aligned(table, 16) && pie_aligned(out, 16), PIE_E_ALIGN, "pie_embedding_dense: 16-byte alignment required");
    hipLaunchKernelGGL(k_embedding_dense, dim3(L), dim3(256), 0, (hipStream_t)stream, ids, (const uint4 *)table, V, H, (uint4 *)out);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_qgemv_w4g64(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype,
                    void *stream) {
    PIE_REQUIRE(x && packed && y, PIE_E_ARG, "pie_qgemv_w4g64: null pointer");
    PIE_REQUIRE(M > 0 && M <= 65535, PIE_E_SHAPE, "pie_qgemv_w4g64: M out of range");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(packed, 16) && pie_aligned(y, 4), PIE_E_ALIGN, "pie_qgemv_w4g64: misaligned pointer");
    // one pass over the weights per up to GEMV_ROWS_MAX rows (k_w4s_gemv_rows); every row with the batch-1 arithmetic
    return w4s_gemv_rows_launch(dtype, packed, N, K, (const u16 *)x, M, (u16 *)y, (const u16 *)lin_bias, (hipStream_t)stream);
}

int pie_embedding_w4g64(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H,
                        int dtype, void *out, void *stream) {
    return embedding_launch(ids, L, codes, scales, biases, V, H, dtype, out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, 4);
}
int pie_embedding_g64(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H, int bits,
                      int dtype, void *out, void *stream) {
    PIE_REQUIRE(bits == 4 || bits == 8, PIE_E_ARG, "pie_embedding_g64: bits must be 4 or 8");
    return embedding_launch(ids, L, codes, scales, biases, V, H, dtype, out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, bits);
}
}  // extern "C"

int embedding_launch(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H, int dtype,
                     void *out, const float *freqs, const DecState *state, float *rope_cs, int half, hipStream_t st, int bits) {
    PIE_REQUIRE(ids && codes && scales && biases && out, PIE_E_ARG, "pie_embedding_w4g64: null pointer");
    PIE_REQUIRE(L > 0 && V > 0 && H > 0 && H % 64 == 0, PIE_E_SHAPE, "pie_embedding_w4g64: H must be a multiple of 64");
    PIE_REQUIRE(pie_aligned(out, 16), PIE_E_ALIGN, "pie_embedding_w4g64: out must be 16-byte aligned");
    PIE_REQUIRE(half <= 256, PIE_E_SHAPE, "embedding: head_dim too large for the RoPE table");
#define PIE_EMB(TT, BB)                                                                                                            \
    hipLaunchKernelGGL((k_embedding_w4g64<TT, BB>), dim3(L), dim3(256), 0, st, ids, codes, (const u16 *)scales, (const u16 *)biases, V, H, \
                       (u16 *)out, freqs, state, rope_cs, half)
    if (dtype == PIE_BF16 && bits == PIE_EMBED_W4G32)
        hipLaunchKernelGGL((k_embedding_w4g64<BF16, 4, 32>), dim3(L), dim3(256), 0, st, ids, codes, (const u16 *)scales, (const u16 *)biases, V, H, (u16 *)out, freqs, state,
                           rope_cs, half);
    else if (dtype == PIE_F16 && bits == PIE_EMBED_W4G32)
        hipLaunchKernelGGL((k_embedding_w4g64<F16, 4, 32>), dim3(L), dim3(256), 0, st, ids, codes, (const u16 *)scales, (const u16 *)biases, V, H, (u16 *)out, freqs, state,
                           rope_cs, half);
    else if (dtype == PIE_BF16 && bits == PIE_EMBED_W8G32)
        hipLaunchKernelGGL((k_embedding_w4g64<BF16, 8, 32>), dim3(L), dim3(256), 0, st, ids, codes, (const u16 *)scales, (const u16 *)biases, V, H, (u16 *)out, freqs, state,
                           rope_cs, half);
    else if (dtype == PIE_F16 && bits == PIE_EMBED_W8G32)
        hipLaunchKernelGGL((k_embedding_w4g64<F16, 8, 32>), dim3(L), dim3(256), 0, st, ids, codes, (const u16 *)scales, (const u16 *)biases, V, H, (u16 *)out, freqs, state,
                           rope_cs, half);
    else if (dtype == PIE_BF16 && bits == 8) PIE_EMB(BF16, 8);
    else if (dtype == PIE_F16 && bits == 8) PIE_EMB(F16, 8);
    else if (dtype == PIE_BF16) PIE_EMB(BF16, 4);
    else if (dtype == PIE_F16) PIE_EMB(F16, 4);
#undef PIE_EMB
    else
        return pie::fail(PIE_E_ARG, "pie_embedding_w4g64: bad dtype");
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

extern "C" {

int pie_qkv_row_map(int n_heads, int n_kv_heads, int head_dim, int32_t *map) {
    if (!map || n_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0 || (head_dim & 1)) return pie::fail(PIE_E_ARG, "pie_qkv_row_map: bad argument");
    const int D = head_dim, half = D / 2;
    int r = 0;
    for (int h = 0; h < n_heads + n_kv_heads; ++h)  // q heads then k heads: RoPE partners (i, i+D/2) adjacent
        for (int i = 0; i < half; ++i) {
            map[r++] = h * D + i;
            map[r++] = h * D + i + half;
        }
    for (int j = 0; j < n_kv_heads * D; ++j) map[r++] = (n_heads + n_kv_heads) * D + j;  // v rows in natural order
    return PIE_OK;
}

int pie_gateup_row_map(int inter, int32_t *map) {
    if (!map || inter <= 0) return pie::fail(PIE_E_ARG, "pie_gateup_row_map: bad argument");
    for (int i = 0; i < inter; ++i) {
        map[2 * i] = i;              // gate_i
        map[2 * i + 1] = inter + i;  // up_i
    }
    return PIE_OK;
}

}  // extern "C"
