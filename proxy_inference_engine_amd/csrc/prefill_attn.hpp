// prefill_attn.hpp -- causal flash attention for prompt chunks on the MFMA units (head_dim 128 or 64).
//
// Replaces mx.fast.scaled_dot_product_attention(q, k, v, scale, mask=causal) at models/base.py:111-113 for L > 1
// (mask built by models/base.py:18-53: query row r at absolute position offset + r sees keys 0 .. offset + r).
// The VALU decode kernel run once per query row costs 27 wave-instructions per (query, key, kv-group): O(L^2) and the
// whole prefill beyond ~1k tokens.  Here one workgroup owns (32 query rows) x (one kv-head group); wave w owns q-head
// g*REP + w, so a staged K/V block serves all q-heads of the group.
//
// Per 32-key block, per wave (v_mfma_f32_32x32x16, fp32 accumulators):
//   S^T[key, row]  = K . Q^T        8 MFMAs;  A = K rows from the LDS image (ds_read_b128), B = Q fragments kept in VGPRs.
//                                   Orientation chosen so the softmax reduction runs over REGISTERS (keys), not lanes,
//                                   and so P^T is directly the B operand of the next product (no lane movement);
//   online softmax in fp32, base 2 (one v_exp_f32 per score), running max / sum per query row = per lane pair;
//   O^T[dim, row] += V^T . P^T      A = V^T via ds_read_b64_tr_b16 (hardware transpose of the row-major V image),
//                                   B = P^T split into hi + lo halves of T (2 x 8 MFMAs): P keeps ~16 mantissa bits, the
//                                   "fp32 until one rounding" contract of the fused kernel within 2^-17.
// K/V blocks are fetched to registers one block ahead and written to a 16 KB swizzled LDS image (the guide's dual-use
// layout (b): conflict-free for both the row reads and the transposed reads).
#pragma once
#include "attention.hpp"
#include "common.hpp"

#include <type_traits>

typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct PrefillAttnArgs {
    const u16 *q;                        // [M, Hq, D]
    const unsigned long long *kv_table;  // K buffers then V buffers, [Hkv, cap, D] each
    int layer, n_layers;
    const DecState *state;               // pos = offset of row 0, cap  (decoder) ...
    const u16 *k, *v;                    // ... or direct [Hkv, cap, D] buffers with host-side offset / capacity (op-level, state == nullptr)
    int offset, cap;
    int M, Hq, Hkv;
    float scale;
    u16 *out;                            // [M, Hq, D]
    // SEG instantiation (vision tower: block-diagonal, non-causal attention -- the mask of models/intern/vision.py:160-167):
    // query row r attends keys [seg_lo[r], seg_hi[r]); both int32 [M], non-decreasing in r.  Op level only (k, v, cap = M).
    const int *seg_lo, *seg_hi;
    const int *block_table;              // paged KV (nullable, decoder only): kv_table holds the layers' slab K / V bases; key t
    int n_pages;                         //   lives in page block_table[t / 64] (each [Hkv, 64, D]) at row t % 64
};

template <class T> struct MfmaT;
template <> struct MfmaT<BF16> {
    static __device__ __forceinline__ f32x16_t run(const uint4 &a, const uint4 &b, const f32x16_t &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct MfmaT<F16> {
    static __device__ __forceinline__ f32x16_t run(const uint4 &a, const uint4 &b, const f32x16_t &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

// byte offset of 16-byte chunk `ch` (0 .. D/8-1) of row `row` (0..31) in a [32][D x 16-bit] LDS tile: the guide's dual-use
// swizzle (b) for 256-byte rows; for D = 64 the same XOR folded to the row's 8 chunks (a bijection inside the row, which is
// all correctness needs -- both the writes and the two kinds of reads go through it)
template <int I, int N, class F>
__device__ __forceinline__ void pa_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pa_static_for<I + 1, N>(f);
    }
}

template <int D>
__device__ __forceinline__ int pa_off(int row, int ch) {
    return 2 * D * row + 16 * (ch ^ ((((row & 3) << 2) | ((row >> 2) & 3)) & (D / 8 - 1)));
}

// QT: 32-row query tiles per workgroup (1 or 2).  With QT = 2 the workgroup has 2*REP waves -- waves [0, REP) own the first
// tile, [REP, 2 REP) the second -- and one staged K/V block serves 64 query rows: half the L2 -> LDS traffic and barriers per
// unit of work, at the same number of waves per CU (one 8-wave workgroup instead of two 4-wave ones for REP = 4).
template <class T, int D, int REP, int QT, bool SEG = false>
__global__ void __launch_bounds__(REP * QT * 64) k_prefill_attn(const PrefillAttnArgs a) {
    constexpr int BK = 32, NT = REP * QT * 64, CH = D / 8, KS = D / 16, DT = D / 32;  // chunks per row, k-steps of Q.K^T, 32-dim output tiles
    constexpr int BM = 32 * QT;
    constexpr int CPT = (BK * CH + NT - 1) / NT;                                   // 16-byte chunks per thread and tile
    __shared__ __attribute__((aligned(16))) char k_lds[BK * 2 * D], v_lds[BK * 2 * D];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    // XCD-aware tile mapping: workgroups go round-robin over the 8 XCDs by linear id, so the kv-head group is the FASTEST
    // index (id % Hkv): with 8 kv-heads every XCD's L2 only ever holds one group's K/V (4 MB at 4096 tokens) instead of all
    // of them.  Query tiles longest first (a tile's work grows with its position: causal).
    const int g = (int)blockIdx.x % a.Hkv, tile = (int)blockIdx.x / a.Hkv;
    const int wg_r0 = ((a.M + BM - 1) / BM - 1 - tile) * BM;                  // first query row of the workgroup
    const int r0 = wg_r0 + 32 * (wave / REP), hq = g * REP + wave % REP;     // this wave's 32-row tile and q-head
    // paged KV: a key block (BK = 32 rows) never straddles a 64-row page, so its page is one scalar table lookup per fetch;
    // the contiguous case is the same code with one "page" of `cap` rows (mask keeps every row bit, page offset 0)
    const bool paged = a.block_table != nullptr;
    const int pos0 = a.state ? a.state->pos : a.offset, cap = paged ? 64 : a.state ? a.state->cap : a.cap;
    const int row_mask = paged ? 63 : 0x7FFFFFFF;
    const size_t page_elems = (size_t)2 * 64 * a.Hkv * D;
    const int r_last = (wg_r0 + BM - 1 < a.M ? wg_r0 + BM - 1 : a.M - 1);
    // last / first key any row of this tile attends (segment bounds are non-decreasing in the row index)
    // (bounds clamped to the buffer: a bad segment table must not become a wild address)
    const int t_last = SEG ? max(min(a.seg_hi[r_last], a.cap), 1) - 1 : pos0 + r_last;
    const int b_first = SEG ? min(max(a.seg_lo[wg_r0], 0), t_last) / BK : 0;
    const int n_blocks = t_last / BK + 1;
    const u16 *kbase = (a.state ? reinterpret_cast<const u16 *>(a.kv_table[a.layer]) : a.k) + (size_t)g * cap * D;
    const u16 *vbase = (a.state ? reinterpret_cast<const u16 *>(a.kv_table[a.n_layers + a.layer]) : a.v) + (size_t)g * cap * D;

    // staging: chunk id x = threadIdx.x + i*NT -> (row = x / CH, ch = x % CH); keys past t_last are clamped (masked later).
    // Eight NAMED register pairs, used up to CPT: as arrays (even with compile-time indices) the compiler parked them in
    // scratch memory and reloaded them every block (80-272 B per lane).
    uint4 k0, k1, k2, k3, k4, k5, k6, k7, v0, v1, v2, v3, v4, v5, v6, v7;
#define PA_FETCH1(i, kr, vr)                                                                   \
    if constexpr (i < CPT) {                                                                     \
        int x = threadIdx.x + i * NT;                                                            \
        x = x < BK * CH ? x : BK * CH - 1;                                                       \
        int t = b * BK + x / CH;                                                                 \
        t = t < t_last ? t : t_last;                                                             \
        kr = *reinterpret_cast<const uint4 *>(kbase + pg_off + (size_t)(t & row_mask) * D + (x % CH) * 8); \
        vr = *reinterpret_cast<const uint4 *>(vbase + pg_off + (size_t)(t & row_mask) * D + (x % CH) * 8); \
    }
#define PA_PUBLISH1(i, kr, vr)                                                                 \
    if constexpr (i < CPT) {                                                                     \
        const int x = threadIdx.x + i * NT;                                                      \
        if (x < BK * CH) {                                                                       \
            *reinterpret_cast<uint4 *>(k_lds + pa_off<D>(x / CH, x % CH)) = kr;                  \
            *reinterpret_cast<uint4 *>(v_lds + pa_off<D>(x / CH, x % CH)) = vr;                  \
        }                                                                                        \
    }
    auto fetch = [&](int b) {
        const size_t pg_off = paged ? (size_t)min((unsigned)a.block_table[(b * BK) >> 6], (unsigned)a.n_pages - 1u) * page_elems : 0;
        PA_FETCH1(0, k0, v0) PA_FETCH1(1, k1, v1) PA_FETCH1(2, k2, v2) PA_FETCH1(3, k3, v3)
        PA_FETCH1(4, k4, v4) PA_FETCH1(5, k5, v5) PA_FETCH1(6, k6, v6) PA_FETCH1(7, k7, v7)
    };
    auto publish = [&]() {
        PA_PUBLISH1(0, k0, v0) PA_PUBLISH1(1, k1, v1) PA_PUBLISH1(2, k2, v2) PA_PUBLISH1(3, k3, v3)
        PA_PUBLISH1(4, k4, v4) PA_PUBLISH1(5, k5, v5) PA_PUBLISH1(6, k6, v6) PA_PUBLISH1(7, k7, v7)
    };
#undef PA_FETCH1
#undef PA_PUBLISH1
    static_assert(CPT <= 8, "staging registers");
    fetch(b_first);

    // Q fragments (B operand of K . Q^T): lane (row c, half h), k-step s holds Q[r0 + c][hq][16 s + 8 h .. + 8]
    const int qrow = r0 + c < a.M ? r0 + c : a.M - 1;
    uint4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const uint4 *>(a.q + ((size_t)qrow * a.Hq + hq) * D + 16 * s + 8 * h);
    // last (and, SEG, first) key this lane's query row attends (rows past M are padding: never stored)
    const int t_row = SEG ? a.seg_hi[qrow] - 1 : pos0 + r0 + c;
    const int t_lo = SEG ? a.seg_lo[qrow] : 0;
    const float sl2 = a.scale * ATTN_LOG2E;

    f32x16_t oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.0f;
    float m_run = ATTN_NEG, l_run = 0.0f;

    // transposed-read addressing (guide T10): lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of a
    // 4-row x 16-column block; lane i of the group receives column i, row q in element q
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;

    for (int b = b_first; b < n_blocks; ++b) {
        __syncthreads();  // every wave is done reading the previous block's image
        publish();
        __syncthreads();
        if (b + 1 < n_blocks) fetch(b + 1);

        // S^T = K . Q^T : rows = keys (registers), cols = query rows (lanes)
        f32x16_t sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 kf = *reinterpret_cast<const uint4 *>(k_lds + pa_off<D>(c, 2 * s + h));
            sacc = MfmaT<T>::run(kf, qf[s], sacc);
        }
        // online softmax over keys: register i <-> key t0 + (i & 3) + 8 (i >> 2) + 4 h
        const int t0 = b * BK;
        float sc[16], mloc = ATTN_NEG;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = t0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            sc[i] = (t <= t_row && (!SEG || t >= t_lo)) ? sacc[i] * sl2 : ATTN_NEG;
            mloc = fmaxf(mloc, sc[i]);
        }
        mloc = xor32_max(mloc);  // the other 16 keys of the block live in the partner lane
        const float m_new = fmaxf(m_run, mloc);
        if (m_new > m_run) {  // rare after the first blocks
            const float alpha = attn_exp2(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
            m_run = m_new;
        }
        u32 phi[8], plo[8];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const int ta = t0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const float pa = (ta <= t_row && (!SEG || ta >= t_lo)) ? attn_exp2(sc[i] - m_run) : 0.0f;
            const float pb = (ta + 1 <= t_row && (!SEG || ta + 1 >= t_lo)) ? attn_exp2(sc[i + 1] - m_run) : 0.0f;
            l_run += pa + pb;
            const float ha = round_T<T>(pa), hb = round_T<T>(pb);
            phi[i >> 1] = pack2<T>(ha, hb);
            plo[i >> 1] = pack2<T>(pa - ha, pb - hb);
        }
        // O^T += V^T . P^T : k-step s covers keys 16 s .. 16 s + 15 in the operand order 16 s + 8 (j >> 2) + 4 h + (j & 3)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint4 bh = make_uint4(phi[4 * s], phi[4 * s + 1], phi[4 * s + 2], phi[4 * s + 3]);
            const uint4 bl = make_uint4(plo[4 * s], plo[4 * s + 1], plo[4 * s + 2], plo[4 * s + 3]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int ch = 4 * dt + 2 * (grp & 1) + (tp >> 1);
                const v4i16_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) v4i16_t *)(v_lds + pa_off<D>(16 * s + 4 * h + tq, ch) + 8 * (tp & 1)));
                const v4i16_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) v4i16_t *)(v_lds + pa_off<D>(16 * s + 8 + 4 * h + tq, ch) + 8 * (tp & 1)));
                const uint2 a0 = __builtin_bit_cast(uint2, lo4), a1 = __builtin_bit_cast(uint2, hi4);
                const uint4 vf = make_uint4(a0.x, a0.y, a1.x, a1.y);
                oacc[dt] = MfmaT<T>::run(vf, bh, oacc[dt]);
                oacc[dt] = MfmaT<T>::run(vf, bl, oacc[dt]);
            }
        }
    }
    // normalise: the pair (lane, lane ^ 32) shares the query row and holds the two halves of the row sum
    const float inv = 1.0f / xor32_sum(l_run);
    if (r0 + c < a.M) {
        u16 *orow = a.out + ((size_t)(r0 + c) * a.Hq + hq) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // registers 4k .. 4k+3 <-> dims 32 dt + 8 k + 4 h + {0..3}
                const uint2 o = make_uint2(pack2<T>(oacc[dt][4 * k] * inv, oacc[dt][4 * k + 1] * inv),
                                           pack2<T>(oacc[dt][4 * k + 2] * inv, oacc[dt][4 * k + 3] * inv));
                *reinterpret_cast<uint2 *>(orow + 32 * dt + 8 * k + 4 * h) = o;
            }
    }
}

template <class T, int D>
static int prefill_attn_launch_d(const PrefillAttnArgs &a, hipStream_t st) {
    const int rep = a.Hq / a.Hkv;
    const int qt = pie_knob(PIE_KNOB_PREFILL_QT);  // test knob: 1 forces one 32-row query tile per workgroup, 2 forces two wherever they fit
    // 2 tiles need 2*rep <= 8 waves of up to 256 registers, and only pay while the halved grid still fills the 256 CUs
    // (8B model: 4096 tokens 54.9 -> 53.9 ms, 8000 tokens 126.3 -> 125.0 ms; at 1024 tokens it would idle half the chip)
    const bool two = rep <= 4 && a.M > 32 && (qt > 0 ? qt == 2 : ((a.M + 63) / 64) * a.Hkv >= 256);
    const dim3 grid(((a.M + (two ? 63 : 31)) / (two ? 64 : 32)) * a.Hkv);
#define PA_LAUNCH(R)                                                                                         \
    if (two && R <= 4) hipLaunchKernelGGL((k_prefill_attn<T, D, R, (R <= 4 ? 2 : 1)>), grid, dim3(R * 128), 0, st, a); \
    else hipLaunchKernelGGL((k_prefill_attn<T, D, R, 1>), grid, dim3(R * 64), 0, st, a);                     \
    break
    switch (rep) {
        case 1: PA_LAUNCH(1);
        case 2: PA_LAUNCH(2);
        case 3: PA_LAUNCH(3);
        case 4: PA_LAUNCH(4);
        case 5: PA_LAUNCH(5);
        case 6: PA_LAUNCH(6);
        case 7: PA_LAUNCH(7);
        case 8: PA_LAUNCH(8);
        default: return pie::fail(PIE_E_SHAPE, "prefill attention: n_heads / n_kv_heads must be between 1 and 8");
    }
#undef PA_LAUNCH
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}
// block-diagonal attention (SEG): multi-head only (Hq == Hkv, the vision tower), two query tiles per workgroup while the
// halved grid still fills the chip
template <class T, int D>
static int segment_attn_launch_d(const PrefillAttnArgs &a, hipStream_t st) {
    if (a.Hq != a.Hkv) return pie::fail(PIE_E_SHAPE, "segment attention: n_heads must equal n_kv_heads");
    const bool two = a.M > 32 && ((a.M + 63) / 64) * a.Hkv >= 512;
    const dim3 grid(((a.M + (two ? 63 : 31)) / (two ? 64 : 32)) * a.Hkv);
    if (two) hipLaunchKernelGGL((k_prefill_attn<T, D, 1, 2, true>), grid, dim3(128), 0, st, a);
    else hipLaunchKernelGGL((k_prefill_attn<T, D, 1, 1, true>), grid, dim3(64), 0, st, a);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}
// the same with grouped-query heads (several prompts prefilled in one pass: causal inside each prompt = seg_hi[r] = r + 1)
template <class T, int D>
static int segment_attn_gqa_launch_d(const PrefillAttnArgs &a, hipStream_t st) {
    const int rep = a.Hq / a.Hkv;
    const dim3 grid(((a.M + 31) / 32) * a.Hkv);
#define PA_SEG(R) hipLaunchKernelGGL((k_prefill_attn<T, D, R, 1, true>), grid, dim3(R * 64), 0, st, a); break
    switch (rep) {
        case 1: PA_SEG(1);
        case 2: PA_SEG(2);
        case 3: PA_SEG(3);
        case 4: PA_SEG(4);
        case 5: PA_SEG(5);
        case 6: PA_SEG(6);
        case 7: PA_SEG(7);
        case 8: PA_SEG(8);
        default: return pie::fail(PIE_E_SHAPE, "segment attention: n_heads / n_kv_heads must be between 1 and 8");
    }
#undef PA_SEG
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}
template <class T>
static int segment_attn_gqa_launch_t(const PrefillAttnArgs &a, int D, hipStream_t st) {
    return D == 128 ? segment_attn_gqa_launch_d<T, 128>(a, st) : segment_attn_gqa_launch_d<T, 64>(a, st);
}

template <class T>
static int segment_attn_launch_t(const PrefillAttnArgs &a, int D, hipStream_t st) {
    return D == 128 ? segment_attn_launch_d<T, 128>(a, st) : segment_attn_launch_d<T, 64>(a, st);
}

template <class T>
static int prefill_attn_launch_t(const PrefillAttnArgs &a, int D, hipStream_t st) {
    return D == 128 ? prefill_attn_launch_d<T, 128>(a, st) : prefill_attn_launch_d<T, 64>(a, st);
}
