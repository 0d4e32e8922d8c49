// attention.hpp -- split-KV decode attention (L = 1) over the ReusableKVCache buffers.
//
// Replaces mx.fast.scaled_dot_product_attention(q, k, v, scale, mask=None) at models/base.py:111-113
// (<- models/llama/language.py:98-105) for the decode step.  HBM-bound KV read; K/V rows go straight to
// VGPRs in 16-byte pieces (a wave-load covers 64/LPT whole rows, coalesced), every q-head of the GQA
// group is scored against the same loaded K/V so the cache is read once; online softmax in fp32 per
// lane-group, merged through LDS; one partial (m, l, acc[D]) per (q-head, split), merged by
// k_attn_combine (fp32 until the single rounding at the end = MLX fused-kernel contract).
#pragma once
#include "common.hpp"
#include "w4_gemv.hpp"  // DecState

constexpr int ATTN_MAX_SPLITS = 64;
constexpr float ATTN_NEG = -3.0e38f;

struct AttnArgs {
    const u16 *q;         // [Hq, D]
    const u16 *k, *v;     // [Hkv, cap, D] (op-level) ...
    const unsigned long long *kv_table;  // ... or the decoder's pointer table (k == nullptr)
    int layer, n_layers;
    const DecState *state;  // nullable: T = state->pos + 1, cap = state->cap
    int T, cap;
    int Hq, Hkv, splits;
    float scale;
    float *part_acc;  // [Hq, splits, D]
    float *part_ml;   // [Hq, splits, 2]
    u16 *out;         // [Hq, D]
};

template <class T, int D, int REP>
__global__ void __launch_bounds__(256) k_attn_decode(const AttnArgs a) {
    constexpr int LPT = D / 8;     // lanes per token row (16 B each)
    constexpr int TPW = 64 / LPT;  // token rows per wave-load
    constexpr int NSUB = 4;        // one merged online-softmax stream per wave reaches LDS
    __shared__ float s_m[REP][NSUB], s_l[REP][NSUB];
    __shared__ float s_acc[REP][NSUB][D];

    const int g = blockIdx.x, split = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ts = lane / LPT, dc = lane % LPT;
    const int Ttot = a.state ? a.state->pos + 1 : a.T;
    const int cap = a.state ? a.state->cap : a.cap;
    const int chunk = (Ttot + a.splits - 1) / a.splits;
    const int t_begin = split * chunk;
    const int t_end = min(Ttot, t_begin + chunk);

    const u16 *kbase = a.k ? a.k : reinterpret_cast<const u16 *>(a.kv_table[a.layer]);
    const u16 *vbase = a.k ? a.v : reinterpret_cast<const u16 *>(a.kv_table[a.n_layers + a.layer]);
    kbase += (size_t)g * cap * D + dc * 8;
    vbase += (size_t)g * cap * D + dc * 8;

    u32 qr[REP][4];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        uint4 qv = *reinterpret_cast<const uint4 *>(a.q + (size_t)(g * REP + h) * D + dc * 8);
        qr[h][0] = qv.x, qr[h][1] = qv.y, qr[h][2] = qv.z, qr[h][3] = qv.w;
    }
    float m[REP], l[REP], acc[REP][8];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        m[h] = ATTN_NEG, l[h] = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = 0.0f;
    }

    for (int t0 = t_begin + wave * TPW; t0 < t_end; t0 += 4 * TPW) {
        const int t = t0 + ts;
        const bool valid = t < t_end;
        const int tc = valid ? t : t_end - 1;
        const uint4 kv = *reinterpret_cast<const uint4 *>(kbase + (size_t)tc * D);
        const uint4 vv = *reinterpret_cast<const uint4 *>(vbase + (size_t)tc * D);
        const u32 kw[4] = {kv.x, kv.y, kv.z, kv.w};
        float vf[8];
        vf[0] = lo_f32<T>(vv.x), vf[1] = hi_f32<T>(vv.x), vf[2] = lo_f32<T>(vv.y), vf[3] = hi_f32<T>(vv.y);
        vf[4] = lo_f32<T>(vv.z), vf[5] = hi_f32<T>(vv.z), vf[6] = lo_f32<T>(vv.w), vf[7] = hi_f32<T>(vv.w);
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            float sc = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sc = T::dot2(qr[h][j], kw[j], sc);
#pragma unroll
            for (int o = LPT / 2; o > 0; o >>= 1) sc += __shfl_xor(sc, o, 64);
            sc *= a.scale;
            const float m_new = valid ? fmaxf(m[h], sc) : m[h];
            const float alpha = expf(m[h] - m_new);
            const float p = valid ? expf(sc - m_new) : 0.0f;
            l[h] = l[h] * alpha + p;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[h][j] = fmaf(acc[h][j], alpha, p * vf[j]);
            m[h] = m_new;
        }
    }

    // merge the TPW lane-groups of the wave (lanes with equal dc) with xor shuffles, then one stream per wave
#pragma unroll
    for (int o = LPT; o < 64; o <<= 1) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            const float m_o = __shfl_xor(m[h], o, 64), l_o = __shfl_xor(l[h], o, 64);
            const float m_new = fmaxf(m[h], m_o);
            const float wa = expf(m[h] - m_new), wb = expf(m_o - m_new);
            l[h] = l[h] * wa + l_o * wb;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[h][j] = acc[h][j] * wa + __shfl_xor(acc[h][j], o, 64) * wb;
            m[h] = m_new;
        }
    }
    const int sub = wave;
    if (ts == 0) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            if (dc == 0) s_m[h][sub] = m[h], s_l[h][sub] = l[h];
#pragma unroll
            for (int j = 0; j < 8; ++j) s_acc[h][sub][dc * 8 + j] = acc[h][j];
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < REP * D; o += 256) {
        const int h = o / D, d = o % D;
        float M = ATTN_NEG;
        for (int i = 0; i < NSUB; ++i) M = fmaxf(M, s_m[h][i]);
        float L = 0.0f, A = 0.0f;
        for (int i = 0; i < NSUB; ++i) {
            const float w = expf(s_m[h][i] - M);
            L = fmaf(w, s_l[h][i], L);
            A = fmaf(w, s_acc[h][i][d], A);
        }
        const int hq = g * REP + h;
        a.part_acc[((size_t)hq * a.splits + split) * D + d] = A;
        if (d == 0) {
            a.part_ml[((size_t)hq * a.splits + split) * 2 + 0] = M;
            a.part_ml[((size_t)hq * a.splits + split) * 2 + 1] = L;
        }
    }
}

// out[h, d] = T( sum_j w_j acc_j[d] / sum_j w_j l_j ),  w_j = exp(m_j - max m).  Grid Hq, block D.
template <class T>
__global__ void k_attn_combine(const AttnArgs a, int D) {
    const int h = blockIdx.x, d = threadIdx.x;
    float M = ATTN_NEG;
    for (int j = 0; j < a.splits; ++j) M = fmaxf(M, a.part_ml[((size_t)h * a.splits + j) * 2]);
    float L = 0.0f, A = 0.0f;
    for (int j = 0; j < a.splits; ++j) {
        const float w = expf(a.part_ml[((size_t)h * a.splits + j) * 2] - M);
        L = fmaf(w, a.part_ml[((size_t)h * a.splits + j) * 2 + 1], L);
        A = fmaf(w, a.part_acc[((size_t)h * a.splits + j) * D + d], A);
    }
    a.out[(size_t)h * D + d] = T::from_f32(A / L);
}

int attn_decode_launch(int dtype, int D, AttnArgs &a, hipStream_t stream);
