// attn_head.hpp -- decode attention (L = 1) of ONE q-head by one workgroup, unsplit: the attention plan of the persistent step
// (step_engine.hip) and its launched twin (k_attn_head), which share every arithmetic line below so the two paths stay
// bit-identical.
//
// Replaces mx.fast.scaled_dot_product_attention(q, k, v, scale, mask=None) at models/base.py:111-113 (<- models/llama/language.py:98-105)
// for short caches.  Why per q-head and unsplit (the launched default, attention.hpp, scores a GQA group's 4 heads against K/V rows
// loaded once and splits the positions 4 ways): inside the persistent step a split needs a merge, and a merge is one more
// all-to-all hand-off (~3 us) -- a workgroup per q-head publishes the finished head and the K/V rows' 4x re-read is served by L2.
// Same numerics contract as attention.hpp: fp32 throughout, base-2 online softmax per 16-lane token group, ONE rounding at the end.
//
// Work split: NW waves; lane = (ts, dc): token slot ts of the wave's row block, 16-byte column chunk dc.  Row block b of wave w
// covers tokens (b NW + w) TPW + [0, TPW).  The row of the CURRENT position (appended by this very step) and q come from an LDS
// staging area {q[D] | k_new[D] | v_new[D]} (16-bit): the engine fills it from the q|k|v hand-off granules, the launched twin from
// global memory; all older rows are read from the cache and may be requested before q exists (they do not depend on this step).
#pragma once
#include "attention.hpp"

constexpr int AH_DEPTH = 8;  // row blocks a wave keeps in registers: with 6 waves x 4 rows, 192 positions are in flight before q arrives

template <int DA>
struct AttnHeadRing {
    uint4 kq[DA], vq[DA];
};

// tokens of row block b for this lane; clamped to the last OLD row (never the row this step appends: it comes from the stage)
template <int D, int NW>
__device__ __forceinline__ int ah_token(int b, int wv, int ts) { return (b * NW + wv) * (64 / (D / 8)) + ts; }

template <class T, int D, int NW, int DA>
__device__ __forceinline__ void attn_head_issue(AttnHeadRing<DA> &r, int d, int b, const u16 *kbase, const u16 *vbase, int pos, int wv, int ts) {
    int t = ah_token<D, NW>(b, wv, ts);
    t = t < pos ? t : (pos > 0 ? pos - 1 : 0);  // clamp, never branch around a load; pos == 0: row 0 is read and ignored
    // global address space, explicitly: a pointer that came out of a table is a FLAT pointer to hipcc, and flat loads count on
    // lgkmcnt as well as vmcnt -- every LDS wait behind them would wait for the cache rows
    typedef u32 ah_u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const ah_u32x4 g_u32x4;
    const ah_u32x4 kv = *(g_u32x4 *)(unsigned long long)(kbase + (size_t)t * D), vv = *(g_u32x4 *)(unsigned long long)(vbase + (size_t)t * D);
    r.kq[d] = make_uint4(kv.x, kv.y, kv.z, kv.w);
    r.vq[d] = make_uint4(vv.x, vv.y, vv.z, vv.w);
}
// kbase / vbase: this kv-head's [cap, D] rows + the lane's column chunk (dc * 8)
template <class T, int D, int NW, int DA>
__device__ __forceinline__ void attn_head_preload(AttnHeadRing<DA> &r, const u16 *kbase, const u16 *vbase, int pos, int wv, int ts) {
#pragma unroll
    for (int d = 0; d < DA; ++d) attn_head_issue<T, D, NW, DA>(r, d, d, kbase, vbase, pos, wv, ts);
}

// Scores this wave's row blocks and leaves one online-softmax stream per (wave, token slot) in LDS:
// s_m / s_l [NW * TPW], s_acc [NW * TPW][D].  `stage` = the LDS rows {q | k_new | v_new}.
template <class T, int D, int NW, int DA>
__device__ __forceinline__ void attn_head_score(AttnHeadRing<DA> &r, const u16 *kbase, const u16 *vbase, const u16 *stage, int pos, int wv, int lane,
                                                float *s_m, float *s_l, float *s_acc) {
    constexpr int LPT = D / 8, TPW = 64 / LPT;
    const int ts = lane / LPT, dc = lane % LPT;
    const int T_tot = pos + 1;
    const float sl2 = (1.0f / sqrtf((float)D)) * ATTN_LOG2E;
    const int first = wv * TPW;
    const int n_blk = first < T_tot ? (T_tot - first + NW * TPW - 1) / (NW * TPW) : 0;
    u32 qr[4];
    {
        const uint4 qv = *reinterpret_cast<const uint4 *>(stage + dc * 8);
        qr[0] = qv.x, qr[1] = qv.y, qr[2] = qv.z, qr[3] = qv.w;
    }
    const uint4 k_new = *reinterpret_cast<const uint4 *>(stage + D + dc * 8), v_new = *reinterpret_cast<const uint4 *>(stage + 2 * D + dc * 8);
    float m = ATTN_NEG, l = 0.0f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    for (int base = 0; base < n_blk; base += DA) {
#pragma unroll
        for (int d = 0; d < DA; ++d) {
            const int b = base + d;
            if (b < n_blk) {  // wave-uniform
                const int t = ah_token<D, NW>(b, wv, ts);
                const bool valid = t < T_tot, is_new = t == pos;
                const uint4 kk = is_new ? k_new : r.kq[d], vv = is_new ? v_new : r.vq[d];
                const u32 kw[4] = {kk.x, kk.y, kk.z, kk.w};
                float vf[8];
                vf[0] = lo_f32<T>(vv.x), vf[1] = hi_f32<T>(vv.x), vf[2] = lo_f32<T>(vv.y), vf[3] = hi_f32<T>(vv.y);
                vf[4] = lo_f32<T>(vv.z), vf[5] = hi_f32<T>(vv.z), vf[6] = lo_f32<T>(vv.w), vf[7] = hi_f32<T>(vv.w);
                float sc = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) sc = T::dot2(qr[j], kw[j], sc);
                sc += __builtin_amdgcn_update_dpp(0.0f, sc, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                sc += __builtin_amdgcn_update_dpp(0.0f, sc, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
                sc += __builtin_amdgcn_update_dpp(0.0f, sc, 0x141, 0xF, 0xF, true);  // row_half_mirror
                if (LPT == 16) sc += __builtin_amdgcn_update_dpp(0.0f, sc, 0x140, 0xF, 0xF, true);  // row_mirror
                sc = valid ? sc * sl2 : ATTN_NEG;
                if (sc > m) {  // rarely taken after the first blocks
                    const float alpha = attn_exp2(m - sc);
                    l *= alpha;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] *= alpha;
                    m = sc;
                }
                const float p = valid ? attn_exp2(sc - m) : 0.0f;
                l += p;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(p, vf[j], acc[j]);
            }
            if (base + d + DA < n_blk) attn_head_issue<T, D, NW, DA>(r, d, base + d + DA, kbase, vbase, pos, wv, ts);  // wave-uniform
        }
    }
    const int str = wv * TPW + ts;
    if (dc == 0) s_m[str] = m, s_l[str] = l;
    *reinterpret_cast<float4 *>(&s_acc[str * D + dc * 8]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(&s_acc[str * D + dc * 8 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// Final pass, one thread per dim pair (o < D / 2), after the streams are visible: the head's output dims (2 o, 2 o + 1) as two T values.
template <class T, int D, int NW>
__device__ __forceinline__ u32 attn_head_finish(const float *s_m, const float *s_l, const float *s_acc, int o) {
    constexpr int NSTR = NW * (64 / (D / 8));
    const int d = 2 * o;
    float M = ATTN_NEG;
#pragma unroll
    for (int i = 0; i < NSTR; ++i) M = fmaxf(M, s_m[i]);
    float Lsum = 0.0f, A0 = 0.0f, A1 = 0.0f;
#pragma unroll
    for (int i = 0; i < NSTR; ++i) {
        const float w = attn_exp2(s_m[i] - M);
        const float2 av = *reinterpret_cast<const float2 *>(&s_acc[i * D + d]);
        Lsum = fmaf(w, s_l[i], Lsum);
        A0 = fmaf(w, av.x, A0), A1 = fmaf(w, av.y, A1);
    }
    return pack2<T>(A0 / Lsum, A1 / Lsum);
}
constexpr int attn_head_lds_bytes(int D, int NW) { return NW * (64 / (D / 8)) * (D + 2) * 4 + 3 * D * 2; }  // streams + the {q | k | v} stage

struct AttnHeadArgs {
    const u16 *q;  // [Hq, D] (RoPE applied)
    const unsigned long long *kv_table;
    int layer, n_layers, Hq, Hkv;
    const DecState *state;
    u16 *out;  // [Hq, D]
};

// The launched twin: grid = Hq, block = NW waves.
template <class T, int D, int NW>
__global__ void __launch_bounds__(NW * 64) k_attn_head(const AttnHeadArgs a) {
    constexpr int LPT = D / 8, TPW = 64 / LPT, NSTR = NW * TPW;
    __shared__ __attribute__((aligned(16))) float s_acc[NSTR * D];
    __shared__ float s_m[NSTR], s_l[NSTR];
    __shared__ __attribute__((aligned(16))) u16 stage[3 * D];
    const int h = blockIdx.x, g = h / (a.Hq / a.Hkv);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ts = lane / LPT, dc = lane % LPT;
    const int pos = a.state->pos, cap = a.state->cap;
    const u16 *kb = reinterpret_cast<const u16 *>(a.kv_table[a.layer]) + (size_t)g * cap * D;
    const u16 *vb = reinterpret_cast<const u16 *>(a.kv_table[a.n_layers + a.layer]) + (size_t)g * cap * D;
    AttnHeadRing<AH_DEPTH> ring;
    attn_head_preload<T, D, NW, AH_DEPTH>(ring, kb + dc * 8, vb + dc * 8, pos, wv, ts);
    if (threadIdx.x < 3 * LPT) {  // 16-byte pieces of q, the new K row, the new V row
        const int which = threadIdx.x / LPT, c = threadIdx.x % LPT;
        const u16 *src = which == 0 ? a.q + (size_t)h * D : (which == 1 ? kb : vb) + (size_t)pos * D;
        *reinterpret_cast<uint4 *>(stage + which * D + c * 8) = *reinterpret_cast<const uint4 *>(src + c * 8);
    }
    __syncthreads();
    attn_head_score<T, D, NW, AH_DEPTH>(ring, kb + dc * 8, vb + dc * 8, stage, pos, wv, lane, s_m, s_l, s_acc);
    __syncthreads();
    if ((int)threadIdx.x < D / 2) *reinterpret_cast<u32 *>(a.out + (size_t)h * D + 2 * threadIdx.x) = attn_head_finish<T, D, NW>(s_m, s_l, s_acc, threadIdx.x);
}
