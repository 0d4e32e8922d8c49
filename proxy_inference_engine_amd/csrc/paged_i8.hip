// paged_i8.hip -- int8 KV pages with per-head scales: the storage pie_core's KVPage declares
// (/root/reference/src/pie_core/include/engine/page.hpp:25-32,109-117: key_cache_ / value_cache_ int8 [64, heads, head_dim],
// key_cache_scale_ / value_cache_scale_ float16 [heads, 1] initialised to ones, "head-wise quant for now"), which nothing in the
// reference writes or reads yet (its paged attention kernel is a dummy, src/kernels/paged_attention.metal:6-23).  SURVEY.md 8 row f2.
//
// The reference fixes the storage, not the arithmetic; this file defines the obvious one and the oracle restates it
// (oracle/pie_oracle.py: kv_i8_quantize, kv_i8_dequantize + its sdpa on the dequantised rows):
//   store  q = clamp(rint(x / s), -127, 127)            x: the T-rounded K / V element as fp32, s: the page's fp16 scale of that head
//   read   x' = fp32(q) * fp32(s)                        used in fp32 by the attention (no rounding to T in between)
// The scales of a page are whatever its owner wrote there (pie_page_i8_set_scales; ones after the pool's construction, like the
// reference's constructor) -- a page's scale never changes under rows already stored, so appends and concurrent readers do not race.
//
// Page layout (pie_page_i8_bytes): int8 K [Hkv][64][D] | int8 V [Hkv][64][D] | fp16 K scales [Hkv] | fp16 V scales [Hkv] | pad to 256 B
// (head-major blocks like the 16-bit pages: one head's 64 rows are one 4-8 KB burst).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/pie_hip.h"
#include "attention.hpp"
#include "common.hpp"

namespace {

#ifndef I8_DEPTH
#define I8_DEPTH 2  // row blocks in flight per wave
#endif

__device__ __forceinline__ float f16_bits_to_f32(u16 h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float i8_to_f32(u32 word, int j) { return (float)(int)(signed char)(word >> (8 * j)); }

// scales of the listed pages (page_ids == nullptr: pages 0 .. n - 1); ks / vs == nullptr: ones
__global__ void __launch_bounds__(256) k_page_i8_set_scales(char *slab, size_t page_bytes, size_t scale_off, const int *page_ids, int n, int n_pages, int Hkv,
                                                            const u16 *ks, const u16 *vs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * Hkv) return;
    const int pi = i / Hkv, g = i % Hkv;
    const unsigned pg = page_ids ? (unsigned)page_ids[pi] : (unsigned)pi;
    if (pg >= (unsigned)n_pages) return;  // a bad id must not become a wild store
    u16 *sc = reinterpret_cast<u16 *>(slab + (size_t)pg * page_bytes + scale_off);
    sc[g] = ks ? ks[g] : (u16)0x3C00u;
    sc[Hkv + g] = vs ? vs[g] : (u16)0x3C00u;
}

// One thread per 8-element piece of the new K and V rows (T [B, Hkv, D]): row (sequence s, kv-head g) is quantised with the scales
// of page block_table[s][positions[s] / 64] and stored at its slot positions[s] % 64.  positions[s] < 0 = idle slot.
// staged (the single-sequence decoder step, B = 1): k / v are a STAGING page [Hkv, 64, D] the q|k|v GEMV's RoPE + append epilogue wrote the T rows
// into (row positions[0] % 64 of every head), not [B, Hkv, D] rows.
template <class T>
__global__ void __launch_bounds__(256) k_paged_kv_append_i8(const uint4 *k, const uint4 *v, char *slab, size_t page_bytes, const int *block_table, int bt_stride,
                                                            const int *positions, int B, int Hkv, int D, int n_pages, int staged) {
    const int ppr = D >> 3, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * Hkv * ppr) return;
    const int s = i / (Hkv * ppr), g = (i / ppr) % Hkv, pc = i % ppr;
    const int pos = positions[s];
    if (pos < 0 || (pos >> 6) >= bt_stride) return;
    if (staged) {
        const size_t src = ((size_t)g * 64 + (pos & 63)) * ppr + pc;
        k += src - i, v += src - i;  // k[i] / v[i] below then read the staged piece
    }
    const unsigned pg = min((unsigned)block_table[(size_t)s * bt_stride + (pos >> 6)], (unsigned)n_pages - 1u);
    char *page = slab + (size_t)pg * page_bytes;
    const size_t blk = (size_t)Hkv * 64 * D;
    const u16 *sc = reinterpret_cast<const u16 *>(page + 2 * blk);
    const float sk = f16_bits_to_f32(sc[g]), sv = f16_bits_to_f32(sc[Hkv + g]);
    auto quant8 = [](const uint4 &x, float sc_) {
        const u32 w[4] = {x.x, x.y, x.z, x.w};
        u32 o[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xf = (j & 1) ? hi_f32<T>(w[j >> 1]) : lo_f32<T>(w[j >> 1]);
            float q = rintf(xf / sc_);
            q = q < -127.0f ? -127.0f : (q > 127.0f ? 127.0f : q);  // (NaN -- a zero scale under a zero -- stores 0 below)
            const int qi = q == q ? (int)q : 0;
            o[j >> 2] |= ((u32)qi & 0xFFu) << (8 * (j & 3));
        }
        return make_uint2(o[0], o[1]);
    };
    const size_t off = ((size_t)g * 64 + (pos & 63)) * D + pc * 8;
    *reinterpret_cast<uint2 *>(page + off) = quant8(k[i], sk);
    *reinterpret_cast<uint2 *>(page + blk + off) = quant8(v[i], sv);
}

// Split-KV decode attention over int8 pages: one workgroup per (kv-head g, split, sequence).  Same lane geometry as k_attn_decode
// (D / 8 lanes per token row, 8 dims each -- here 8 bytes), every token group keeps its own online-softmax stream in the base-2 domain,
// the streams meet in LDS and the split's partial (m, l, acc[D]) goes to the workspace k_attn_combine merges.  K's scale multiplies the
// reduced score (q . q8 through v_dot2 on exact T copies of the codes), V's scale the converted value: x' = fp32(q8) * fp32(s) exactly,
// everything after it in fp32.
template <class T, int D, int REP>
__global__ void __launch_bounds__((REP > 4 ? 2 : 4) * 64) k_paged_attn_i8(const AttnArgs a, size_t page_bytes) {
    constexpr int WAVES = REP > 4 ? 2 : 4;
    constexpr int LPT = D / 8, TPW = 64 / LPT, NSTR = WAVES * TPW, NT = WAVES * 64;
    __shared__ float s_m[REP][NSTR], s_l[REP][NSTR];
    __shared__ float s_acc[REP][NSTR][D];
    const int g = blockIdx.x, split = blockIdx.y, row = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ts = lane / LPT, dc = lane % LPT;
    const int Ttot = a.ctx_len ? a.ctx_len[row] : (a.state ? a.state->pos + 1 : a.T);  // no ctx_len: the single-sequence decoder step (position in the device-side state)
    const AttnSplit sp = attn_split(Ttot, a.splits);
    if (a.splits == 1 && sp.active == 0) {  // idle slot of a one-split batch: zeros, as k_attn_combine leaves it
        for (int o = threadIdx.x; o < REP * D; o += NT) a.out[((size_t)row * a.Hq + g * REP) * D + o] = 0;
        return;
    }
    if (split >= sp.active) return;  // uniform; the combine only reads `active` partials (0 for an idle slot)
    const int t_begin = split * sp.chunk, t_end = min(Ttot, t_begin + sp.chunk);
    const float sl2 = a.scale * ATTN_LOG2E;
    const char *slab = reinterpret_cast<const char *>(a.slab);
    const int *bt = a.block_table + (size_t)row * a.bt_stride;
    const unsigned last_page = (unsigned)a.n_pages - 1u;
    const size_t blk = (size_t)a.Hkv * 64 * D;

    // q stays packed (T pairs) for v_dot2; the K bytes enter as OFFSET codes u = q8 + 128 (one xor per word, then v_cvt_f32_ubyteN -- the
    // signed route costs a bit-field extract more per element), so sum q (u - 128) = dot(q, u) - 128 sum q: the second term once per kernel
    u32 qr[REP][4];
    float q128[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        const uint4 qv = *reinterpret_cast<const uint4 *>(a.q + ((size_t)row * a.Hq + g * REP + h) * D + dc * 8);
        qr[h][0] = qv.x, qr[h][1] = qv.y, qr[h][2] = qv.z, qr[h][3] = qv.w;
        float sq = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sq += lo_f32<T>(qr[h][j]) + hi_f32<T>(qr[h][j]);
        q128[h] = 128.0f * sq;
    }
    auto ub = [](u32 w, int j) { return (float)((w >> (8 * j)) & 0xFFu); };  // v_cvt_f32_ubyteN
    auto pk = [](float lo, float hi) {  // both exact in T (integers <= 255)
        typedef float f2_t __attribute__((ext_vector_type(2)));
        if constexpr (std::is_same<T, BF16>::value) {
            typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
            return __builtin_bit_cast(u32, __builtin_convertvector((f2_t){lo, hi}, b2_t));
        } else {
            typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
            return __builtin_bit_cast(u32, __builtin_convertvector((f2_t){lo, hi}, h2_t));
        }
    };
    float m[REP], l[REP], acc[REP][8];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        m[h] = ATTN_NEG, l[h] = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = 0.0f;
    }
    const int first = t_begin + wave * TPW;
    const int n_blk = first < t_end ? (t_end - first + WAVES * TPW - 1) / (WAVES * TPW) : 0;
    constexpr int DA = I8_DEPTH;
    uint2 kq[DA], vq[DA];
    u32 sq[DA];  // K scale | V scale << 16 of the block's page
    auto issue = [&](int d, int b) {
        int t = first + b * WAVES * TPW + ts;
        t = t < t_end ? t : t_end - 1;  // clamp, never branch around a load
        const char *page = slab + (size_t)min((unsigned)bt[t >> 6], last_page) * page_bytes;
        const size_t off = ((size_t)g * 64 + (t & 63)) * D + dc * 8;
        kq[d] = *reinterpret_cast<const uint2 *>(page + off);
        vq[d] = *reinterpret_cast<const uint2 *>(page + blk + off);
        const u16 *sc = reinterpret_cast<const u16 *>(page + 2 * blk);
        sq[d] = (u32)sc[g] | ((u32)sc[a.Hkv + g] << 16);
    };
#pragma unroll
    for (int d = 0; d < DA; ++d)
        if (d < n_blk) issue(d, d);
    for (int base = 0; base < n_blk; base += DA) {
#pragma unroll
        for (int d = 0; d < DA; ++d) {
            const int b = base + d;
            if (b < n_blk) {  // wave-uniform
                const bool valid = first + b * WAVES * TPW + ts < t_end;
                const float ks = f16_bits_to_f32((u16)(sq[d] & 0xFFFFu)), vs = f16_bits_to_f32((u16)(sq[d] >> 16));
                const u32 kx = kq[d].x ^ 0x80808080u, ky = kq[d].y ^ 0x80808080u, vx = vq[d].x ^ 0x80808080u, vy = vq[d].y ^ 0x80808080u;
                const u32 kp[4] = {pk(ub(kx, 0), ub(kx, 1)), pk(ub(kx, 2), ub(kx, 3)), pk(ub(ky, 0), ub(ky, 1)), pk(ub(ky, 2), ub(ky, 3))};
                const float vneg = -128.0f * vs;  // (u - 128) * vs = fma(u, vs, -128 vs): exact, the product has <= 19 significant bits
                float vf[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) vf[j] = fmaf(ub(vx, j), vs, vneg), vf[4 + j] = fmaf(ub(vy, j), vs, vneg);
                float sc[REP];
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    sc[h] = -q128[h];
#pragma unroll
                    for (int j = 0; j < 4; ++j) sc[h] = T::dot2(qr[h][j], kp[j], sc[h]);
                }
#pragma unroll
                for (int h = 0; h < REP; ++h) sc[h] += __builtin_amdgcn_update_dpp(0.0f, sc[h], 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
#pragma unroll
                for (int h = 0; h < REP; ++h) sc[h] += __builtin_amdgcn_update_dpp(0.0f, sc[h], 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
#pragma unroll
                for (int h = 0; h < REP; ++h) sc[h] += __builtin_amdgcn_update_dpp(0.0f, sc[h], 0x141, 0xF, 0xF, true);  // row_half_mirror
                if (LPT == 16) {
#pragma unroll
                    for (int h = 0; h < REP; ++h) sc[h] += __builtin_amdgcn_update_dpp(0.0f, sc[h], 0x140, 0xF, 0xF, true);  // row_mirror
                }
                bool grow = false;
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    sc[h] = valid ? (sc[h] * ks) * sl2 : ATTN_NEG;
                    grow |= sc[h] > m[h];
                }
                if (grow) {
#pragma unroll
                    for (int h = 0; h < REP; ++h) {
                        const float m_new = sc[h] > m[h] ? sc[h] : m[h];
                        const float alpha = attn_exp2(m[h] - m_new);
                        l[h] *= alpha;
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[h][j] *= alpha;
                        m[h] = m_new;
                    }
                }
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    const float p = valid ? attn_exp2(sc[h] - m[h]) : 0.0f;
                    l[h] += p;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[h][j] = fmaf(p, vf[j], acc[h][j]);
                }
                if (b + DA < n_blk) issue(d, b + DA);
            }
        }
    }
    const int str = wave * TPW + ts;
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        if (dc == 0) s_m[h][str] = m[h], s_l[h][str] = l[h];
        *reinterpret_cast<float4 *>(&s_acc[h][str][dc * 8]) = make_float4(acc[h][0], acc[h][1], acc[h][2], acc[h][3]);
        *reinterpret_cast<float4 *>(&s_acc[h][str][dc * 8 + 4]) = make_float4(acc[h][4], acc[h][5], acc[h][6], acc[h][7]);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < REP * (D / 2); o += NT) {
        const int h = o / (D / 2), d = (o % (D / 2)) * 2;
        float M = ATTN_NEG;
#pragma unroll
        for (int i = 0; i < NSTR; ++i) M = fmaxf(M, s_m[h][i]);
        float Lsum = 0.0f, A0 = 0.0f, A1 = 0.0f;
#pragma unroll
        for (int i = 0; i < NSTR; ++i) {
            const float w = attn_exp2(s_m[h][i] - M);
            const float2 av = *reinterpret_cast<const float2 *>(&s_acc[h][i][d]);
            Lsum = fmaf(w, s_l[h][i], Lsum);
            A0 = fmaf(w, av.x, A0), A1 = fmaf(w, av.y, A1);
        }
        const size_t hq = (size_t)row * a.Hq + g * REP + h;
        if (a.splits == 1) {  // one split per sequence: the partial is the result (what k_attn_combine would compute, bit for bit); no combine launch
            *reinterpret_cast<u32 *>(a.out + hq * D + d) = pack2<T>(A0 / Lsum, A1 / Lsum);
            continue;
        }
        *reinterpret_cast<float2 *>(a.part_acc + (hq * a.splits + split) * D + d) = make_float2(A0, A1);
        if (d == 0) {
            a.part_ml[(hq * a.splits + split) * 2 + 0] = M;
            a.part_ml[(hq * a.splits + split) * 2 + 1] = Lsum;
        }
    }
}

template <class T, int D>
int attn_i8_launch(int rep, const AttnArgs &a, size_t page_bytes, hipStream_t st) {
    const dim3 grid(a.Hkv, a.splits, a.rows);
#define I8_GO(R) hipLaunchKernelGGL((k_paged_attn_i8<T, D, R>), grid, dim3((R > 4 ? 2 : 4) * 64), 0, st, a, page_bytes); break
    switch (rep) {
        case 1: I8_GO(1);
        case 2: I8_GO(2);
        case 3: I8_GO(3);
        case 4: I8_GO(4);
        case 5: I8_GO(5);
        case 6: I8_GO(6);
        case 7: I8_GO(7);
        case 8: I8_GO(8);
        default: return pie::fail(PIE_E_SHAPE, "pie_paged_attn_decode_i8: n_heads / n_kv_heads must be between 1 and 8");
    }
#undef I8_GO
    PIE_LAUNCH_CHECK();
    if (a.splits > 1) {
        hipLaunchKernelGGL(k_attn_combine<T>, dim3(a.Hq, a.rows), dim3(256), 0, st, a, D);
        PIE_LAUNCH_CHECK();
    }
    return PIE_OK;
}

}  // namespace

// for the decoder's multi-sequence step on int8 pages (prefill.hip): the AttnArgs of its T-page launch, slab = the layer's int8 slab
int paged_attn_i8_launch(int dtype, int D, const AttnArgs &a, hipStream_t st) {
    PIE_REQUIRE(a.Hkv > 0 && a.Hq % a.Hkv == 0 && a.slab && a.block_table && (a.ctx_len || a.state), PIE_E_ARG, "paged_attn_i8: bad arguments");
    const size_t pb = pie_page_i8_bytes(a.Hkv, D);
    const int rep = a.Hq / a.Hkv;
    if (dtype == PIE_BF16 && D == 128) return attn_i8_launch<BF16, 128>(rep, a, pb, st);
    if (dtype == PIE_BF16 && D == 64) return attn_i8_launch<BF16, 64>(rep, a, pb, st);
    if (dtype == PIE_F16 && D == 128) return attn_i8_launch<F16, 128>(rep, a, pb, st);
    if (dtype == PIE_F16 && D == 64) return attn_i8_launch<F16, 64>(rep, a, pb, st);
    return pie::fail(PIE_E_SHAPE, "paged_attn_i8: head_dim must be 64 or 128 and dtype bf16/f16");
}

// The single-sequence decoder step on int8 pages (decoder.hip): quantise the row the q|k|v GEMV left in the staging page into the sequence's page.
int paged_kv_append_i8_staged_launch(int dtype, const void *stage_k, const void *stage_v, void *slab, int n_pages, const int *block_table, int max_blocks,
                                     const int *position, int Hkv, int D, hipStream_t st) {
    const int n = Hkv * (D >> 3);
    const size_t pb = pie_page_i8_bytes(Hkv, D);
    if (dtype == PIE_BF16)
        hipLaunchKernelGGL(k_paged_kv_append_i8<BF16>, dim3((n + 255) / 256), dim3(256), 0, st, (const uint4 *)stage_k, (const uint4 *)stage_v, (char *)slab, pb, block_table,
                           max_blocks, position, 1, Hkv, D, n_pages, 1);
    else
        hipLaunchKernelGGL(k_paged_kv_append_i8<F16>, dim3((n + 255) / 256), dim3(256), 0, st, (const uint4 *)stage_k, (const uint4 *)stage_v, (char *)slab, pb, block_table,
                           max_blocks, position, 1, Hkv, D, n_pages, 1);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

extern "C" {

int pie_page_i8_set_scales(void *slab, size_t n_pages, int Hkv, int D, const int32_t *page_ids, int n, const void *k_scales, const void *v_scales, void *stream) {
    PIE_REQUIRE(slab, PIE_E_ARG, "pie_page_i8_set_scales: null slab");
    PIE_REQUIRE(n_pages > 0 && n_pages < 0x7FFFFFFFu && Hkv > 0 && D > 0 && D % 8 == 0, PIE_E_SHAPE, "pie_page_i8_set_scales: bad shape");
    PIE_REQUIRE(n > 0 && (page_ids || (size_t)n <= n_pages), PIE_E_ARG, "pie_page_i8_set_scales: n pages out of range");
    const size_t pb = pie_page_i8_bytes(Hkv, D);
    hipLaunchKernelGGL(k_page_i8_set_scales, dim3((unsigned)(((size_t)n * Hkv + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (char *)slab, pb,
                       2 * (size_t)64 * Hkv * D, page_ids, n, (int)n_pages, Hkv, (const u16 *)k_scales, (const u16 *)v_scales);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_paged_kv_append_i8(const void *k, const void *v, void *slab, size_t n_pages, const int32_t *block_table, int max_blocks, const int32_t *positions,
                           int B, int Hkv, int D, int dtype, void *stream) {
    PIE_REQUIRE(k && v && slab && block_table && positions, PIE_E_ARG, "pie_paged_kv_append_i8: null pointer");
    PIE_REQUIRE(B > 0 && Hkv > 0 && max_blocks > 0 && n_pages > 0 && n_pages < 0x7FFFFFFFu, PIE_E_SHAPE, "pie_paged_kv_append_i8: bad shape");
    PIE_REQUIRE(D == 64 || D == 128, PIE_E_SHAPE, "pie_paged_kv_append_i8: head_dim must be 64 or 128");
    PIE_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16, PIE_E_ARG, "pie_paged_kv_append_i8: the rows' dtype must be PIE_BF16 or PIE_F16");
    PIE_REQUIRE(pie_aligned(k, 16) && pie_aligned(v, 16) && pie_aligned(slab, 16), PIE_E_ALIGN, "pie_paged_kv_append_i8: 16-byte alignment required");
    const int n = B * Hkv * (D >> 3);
    const size_t pb = pie_page_i8_bytes(Hkv, D);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PIE_BF16)
        hipLaunchKernelGGL(k_paged_kv_append_i8<BF16>, dim3((n + 255) / 256), dim3(256), 0, st, (const uint4 *)k, (const uint4 *)v, (char *)slab, pb, block_table,
                           max_blocks, positions, B, Hkv, D, (int)n_pages, 0);
    else
        hipLaunchKernelGGL(k_paged_kv_append_i8<F16>, dim3((n + 255) / 256), dim3(256), 0, st, (const uint4 *)k, (const uint4 *)v, (char *)slab, pb, block_table,
                           max_blocks, positions, B, Hkv, D, (int)n_pages, 0);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int pie_paged_attn_decode_i8(const void *q, const void *slab, size_t n_pages, const int32_t *block_table, int max_blocks, const int32_t *context_lens, int B,
                             int Hq, int Hkv, int D, float scale, int dtype, void *out, void *workspace, void *stream) {
    PIE_REQUIRE(q && slab && block_table && context_lens && out && workspace, PIE_E_ARG, "pie_paged_attn_decode_i8: null pointer");
    PIE_REQUIRE(B > 0 && B <= 65535 && max_blocks > 0 && n_pages > 0 && n_pages < 0x7FFFFFFFu, PIE_E_SHAPE, "pie_paged_attn_decode_i8: bad shape");
    PIE_REQUIRE(Hkv > 0 && Hq % Hkv == 0, PIE_E_SHAPE, "pie_paged_attn_decode_i8: Hq must be a multiple of Hkv");
    PIE_REQUIRE(pie_aligned(q, 16) && pie_aligned(slab, 16) && pie_aligned(out, 16), PIE_E_ALIGN, "pie_paged_attn_decode_i8: 16-byte alignment required");
    AttnArgs a = {};
    a.q = (const u16 *)q, a.slab = (const u16 *)slab, a.block_table = block_table, a.ctx_len = context_lens;
    a.bt_stride = max_blocks, a.n_pages = (int)n_pages, a.rows = B;
    a.Hq = Hq, a.Hkv = Hkv, a.scale = scale;
    // enough workgroups across the batch -- 1024 of these 4-wave workgroups put as many waves on the chip as 512 of the T-page kernel's 8-wave
    // ones (with 512: 313 us instead of 223 at 64 sequences x 4096 positions) --, never more splits than pages per sequence
    int splits = (1024 + B * Hkv - 1) / (B * Hkv);
    splits = splits > ATTN_MAX_SPLITS ? ATTN_MAX_SPLITS : splits;
    splits = splits > max_blocks ? max_blocks : splits;
    a.splits = splits < 1 ? 1 : splits;
    a.part_acc = (float *)workspace;
    a.part_ml = a.part_acc + (size_t)B * Hq * a.splits * D;
    a.out = (u16 *)out;
    return paged_attn_i8_launch(dtype, D, a, (hipStream_t)stream);
}

}  // extern "C"
