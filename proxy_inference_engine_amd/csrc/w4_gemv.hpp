// w4_gemv.hpp -- the W4S streaming GEMV: device kernel template + launch arguments.
//
// Replaces mx.quantized_matmul(x, w, scales, biases, transpose=True, group_size=64, bits=4) as reached
// through nn.QuantizedLinear at models/llama/language.py:83,108,127,207-209 of the reference, with the
// neighbouring elementwise ops of the decode graph fused into its prologue / epilogue.
//
// HBM-bound (3.6 FLOP/B at M=1): the design goal is bytes in flight, not math.
//   * a wave owns one 2048-wide K slice; lane (l & 31) owns one quantisation group of that slice and keeps
//     its 64 activations in 32 VGPRs; lanes 0-31 / 32-63 work on the two rows of a pair;
//   * activations: the workgroup loads x ONCE, coalesced (16 B per thread), BEFORE the weight stream is issued
//     (vmcnt retires in order: x must not queue behind the weights), applies the fused RMSNorm on 8 elements per
//     thread, and publishes x through a conflict-free LDS image [8 pieces][groups|1][16 B]; each lane then
//     pulls its group with 8 x ds_read_b128.  Per-group activation sums travel the same way;
//   * weights: per unit a lane issues 2 x global_load_dwordx4 (codes) + 1 x global_load_dword ({scale,bias}),
//     all U units of a wave up front, straight to VGPRs (a streamed-once operand gains nothing from LDS);
//   * dequant = v_and_or_b32 with the magic-exponent trick, 2 codes per op; multiply-accumulate =
//     v_dot2c_f32_bf16 against the packed activations; the +128 offset and the group bias fold into one fma
//     with the group's activation sum:  scale*(d - 128*sx) + bias*sx;
//   * 32-lane DPP reduction per row, cross-slice reduction through LDS, epilogue on <= 32 threads.
#pragma once
#include "common.hpp"

enum { PRO_NONE = 0, PRO_RMSNORM = 1 };
enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_ROPE_KV = 2, EPI_SWIGLU = 3, EPI_LOGITS = 4 };

// Device-resident decode state: lets one captured graph serve every step.
struct DecState {
    int pos;    // cache.offset before the step (reusable.py:111)
    int token;  // input token of the step / greedy output after it
    int cap;    // capacity of the per-layer KV buffers (tokens)
    int pad;
};

struct LogitStat {  // per-tile partial of the log-softmax / argmax tail
    float max, sumexp;
    int argmax, pad;
};

struct GemvArgs {
    const char *w;  // W4S
    int n_pairs, n_slices, row_lanes, K, N;
    const u16 *x;         // [M,K]
    const u16 *norm_w;    // PRO_RMSNORM
    float eps;
    u16 *y;               // EPI_STORE / EPI_LOGITS [M,N]; EPI_SWIGLU act [N/2]
    const u16 *lin_bias;  // EPI_STORE, optional
    u16 *resid;           // EPI_RESIDUAL: residual stream, updated in place
    // EPI_ROPE_KV
    const float *freqs;
    const DecState *state;
    u16 *q_out;
    const unsigned long long *kv_table;  // [2*n_layers] device pointers: K buffers then V buffers
    int layer, n_layers, n_heads, n_kv_heads, head_dim;
    LogitStat *stats;     // EPI_LOGITS
};

// LDS carve-up (dynamic, 16-byte aligned): x image | group sums | cross-slice partials | block reduction scratch
struct GemvLds {
    int stride;  // 16-byte slots per piece row: groups | 1 (odd -> conflict-free ds_write_b128 across the 8 pieces)
    int off_sx, off_part, off_red, total;
};
static inline __host__ __device__ GemvLds gemv_lds(int K, int n_slices) {
    GemvLds l;
    const int G = K >> 6;
    l.stride = G | 1;
    l.off_sx = 8 * l.stride * 16;
    l.off_part = l.off_sx + ((G * 4 + 15) & ~15);
    l.off_red = l.off_part + n_slices * 64 * 4;
    l.total = l.off_red + 32 * 4;
    return l;
}

template <class T>
__device__ __forceinline__ float w4s_unit_dot(const uint4 &c0, const uint4 &c1, const u32 (&xr)[32]) {
    float d = 0.0f;
    const u32 w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 q2 = T::codes2((w[t] >> (4 * i)) & 0x000F000Fu);  // codes (2i, 2i+1) of word t as two T values
            d = T::dot2(q2, xr[4 * t + i], d);
        }
    }
    return d;
}

template <class T>
__device__ __forceinline__ float sum8(const uint4 &v) {
    return ((lo_f32<T>(v.x) + hi_f32<T>(v.x)) + (lo_f32<T>(v.y) + hi_f32<T>(v.y))) +
           ((lo_f32<T>(v.z) + hi_f32<T>(v.z)) + (lo_f32<T>(v.w) + hi_f32<T>(v.w)));
}

template <class T, int PRO, int EPI, int U>
__global__ void __launch_bounds__(1024) k_w4s_gemv(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ns = a.n_slices, RL = a.row_lanes;
    const int s = wave % ns, rl = wave / ns;
    const int P = U * RL;
    const int pair0 = blockIdx.x * P;
    const int m = blockIdx.y;
    const int NT = blockDim.x;
    const GemvLds L = gemv_lds(a.K, ns);
    float *sxs = reinterpret_cast<float *>(smem + L.off_sx);
    float *part = reinterpret_cast<float *>(smem + L.off_part);
    float *red = reinterpret_cast<float *>(smem + L.off_red);

    // 1. activations first (<= 4 pieces of 8 elements per thread, coalesced), then the weight stream.
    const int n_pieces = a.K >> 3;
    const uint4 *xg = reinterpret_cast<const uint4 *>(a.x + (size_t)m * a.K);
    uint4 xv[4], nv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int j = threadIdx.x + i * NT;
        j = j < n_pieces ? j : n_pieces - 1;
        xv[i] = xg[j];
        if (PRO == PRO_RMSNORM) nv[i] = reinterpret_cast<const uint4 *>(a.norm_w)[j];
    }
    uint4 c0[U], c1[U];
    u32 sb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int p = pair0 + rl + u * RL;
        p = p < a.n_pairs ? p : a.n_pairs - 1;  // clamp (never branch around a load); the store is masked instead
        const char *unit = a.w + ((size_t)p * ns + s) * W4S_UNIT_BYTES;
        c0[u] = *reinterpret_cast<const uint4 *>(unit + lane * 16);
        c1[u] = *reinterpret_cast<const uint4 *>(unit + 1024 + lane * 16);
        sb[u] = *reinterpret_cast<const u32 *>(unit + 2048 + lane * 4);
    }

    // 2. fused mx.fast.rms_norm (nn.RMSNorm, language.py:137-141,168): w * T(x * rsqrt(mean(x^2) + eps)), 8 elements per thread
    if (PRO == PRO_RMSNORM) {
        float ssq = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (threadIdx.x + i * NT < n_pieces) {
                const u32 v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = lo_f32<T>(v[k]), hi = hi_f32<T>(v[k]);
                    ssq = fmaf(lo, lo, ssq);
                    ssq = fmaf(hi, hi, ssq);
                }
            }
        }
        ssq = wave_sum(ssq);
        if (lane == 0) red[wave] = ssq;
        __syncthreads();
        float tot = 0.0f;
        for (int i = 0; i < (NT >> 6); ++i) tot += red[i];
        const float inv = 1.0f / sqrtf(tot / (float)a.K + a.eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, g[4] = {nv[i].x, nv[i].y, nv[i].z, nv[i].w};
            u32 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = pack2<T>(round_T<T>(lo_f32<T>(v[k]) * inv) * lo_f32<T>(g[k]), round_T<T>(hi_f32<T>(v[k]) * inv) * hi_f32<T>(g[k]));
            xv[i] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    // publish x: piece j = 8*group + r lands at slot (r*stride + group); group sums via 3 xor-shuffles
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = threadIdx.x + i * NT;
        const bool ok = j < n_pieces;
        float ps = ok ? sum8<T>(xv[i]) : 0.0f;
        ps += __shfl_xor(ps, 1, 64);
        ps += __shfl_xor(ps, 2, 64);
        ps += __shfl_xor(ps, 4, 64);
        if (ok) {
            *reinterpret_cast<uint4 *>(smem + ((size_t)(j & 7) * L.stride + (j >> 3)) * 16) = xv[i];
            if ((j & 7) == 0) sxs[j >> 3] = ps;
        }
    }
    __syncthreads();

    // 3. this lane's group: 64 activations in 32 VGPRs + their sum
    const int n_groups = a.K >> 6;
    const int g = s * 32 + (lane & 31);
    const bool gvalid = g < n_groups;
    const int gc = gvalid ? g : n_groups - 1;
    u32 xr[32];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint4 v = *reinterpret_cast<const uint4 *>(smem + ((size_t)r * L.stride + gc) * 16);
        xr[4 * r + 0] = gvalid ? v.x : 0u;
        xr[4 * r + 1] = gvalid ? v.y : 0u;
        xr[4 * r + 2] = gvalid ? v.z : 0u;
        xr[4 * r + 3] = gvalid ? v.w : 0u;
    }
    const float sx = gvalid ? sxs[gc] : 0.0f;

    // 4. dequant + dot, 32-lane reduction, partials to LDS.
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float d = w4s_unit_dot<T>(c0[u], c1[u], xr);
        const float scale = lo_f32<T>(sb[u]), bias = hi_f32<T>(sb[u]);
        float pr = fmaf(scale, d - T::OFFSET * sx, bias * sx);
        pr = half_wave_sum(pr);
        if ((lane & 31) == 31) part[s * 64 + 2 * (rl + u * RL) + (lane >> 5)] = pr;
    }
    __syncthreads();

    // 5. cross-slice sum + epilogue: thread t < P owns pair t of the tile (rows 2t, 2t+1).
    const int t = threadIdx.x;
    float va = 0.0f, vb = 0.0f;
    const int pair = pair0 + t;
    const bool live = t < P && pair < a.n_pairs;
    if (t < P) {
        for (int i = 0; i < ns; ++i) {
            va += part[i * 64 + 2 * t];
            vb += part[i * 64 + 2 * t + 1];
        }
    }
    const int R = 2 * pair;  // packed row index of va; vb is row R+1

    if (EPI == EPI_STORE || EPI == EPI_LOGITS) {
        float oa = round_T<T>(va), ob = round_T<T>(vb);
        if (EPI == EPI_STORE && a.lin_bias && live) {
            u32 lb = *reinterpret_cast<const u32 *>(a.lin_bias + R);
            oa = round_T<T>(oa + lo_f32<T>(lb));
            ob = round_T<T>(ob + hi_f32<T>(lb));
        }
        if (live) *reinterpret_cast<u32 *>(a.y + (size_t)m * a.N + R) = pack2<T>(oa, ob);
        if (EPI == EPI_LOGITS) {
            if (wave == 0) {  // P <= 32: the whole tile lives in wave 0
                float mx = live ? fmaxf(oa, ob) : -INFINITY;
                int ix = live ? (ob > oa ? R + 1 : R) : 0x7fffffff;
                const float tile_max = wave_max(mx);
                int cand = (live && mx == tile_max) ? ix : 0x7fffffff;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
                float se = live ? expf(oa - tile_max) + expf(ob - tile_max) : 0.0f;
                se = wave_sum(se);
                if (lane == 0) {
                    LogitStat st;
                    st.max = tile_max, st.sumexp = se, st.argmax = cand, st.pad = 0;
                    a.stats[blockIdx.x] = st;
                }
            }
        }
    } else if (EPI == EPI_RESIDUAL) {
        // h = x + r (language.py:151,153): Linear output rounded to T, then the add rounded to T
        if (live) {
            u32 *hp = reinterpret_cast<u32 *>(a.resid + R);
            u32 h = *hp;
            *hp = pack2<T>(lo_f32<T>(h) + round_T<T>(va), hi_f32<T>(h) + round_T<T>(vb));
        }
    } else if (EPI == EPI_SWIGLU) {
        // down_proj input: nn.silu(gate) * up (language.py:127); packed rows (2i, 2i+1) = (gate_i, up_i)
        if (live) {
            const float gte = round_T<T>(va), up = round_T<T>(vb);
            const float sl = round_T<T>(gte / (1.0f + expf(-gte)));
            a.y[pair] = T::from_f32(sl * up);
        }
    } else if (EPI == EPI_ROPE_KV) {
        // packed rows of [q;k;v]: for q/k heads (2i, 2i+1) = dims (i, i + D/2) of one head; v rows natural.
        // RoPE (llama/utils.py:42-50, offset = cache.offset) + cache append (reusable.py:136-137).
        if (live) {
            const int D = a.head_dim, half = D >> 1;
            const int pos = a.state->pos, cap = a.state->cap;
            const int q_rows = a.n_heads * D, k_rows = a.n_kv_heads * D;
            const float ra = round_T<T>(va), rb = round_T<T>(vb);
            if (R < q_rows + k_rows) {
                const int rr = R < q_rows ? R : R - q_rows;
                const int head = rr / D, i = (rr % D) >> 1;
                const float theta = (float)pos * (1.0f / a.freqs[i]);
                float sn, cs;
                sincosf(theta, &sn, &cs);
                const u16 o1 = T::from_f32(__fsub_rn(__fmul_rn(ra, cs), __fmul_rn(rb, sn)));
                const u16 o2 = T::from_f32(__fadd_rn(__fmul_rn(ra, sn), __fmul_rn(rb, cs)));
                u16 *dst = R < q_rows ? a.q_out + (size_t)head * D
                                      : reinterpret_cast<u16 *>(a.kv_table[a.layer]) + ((size_t)head * cap + pos) * D;
                dst[i] = o1;
                dst[i + half] = o2;
            } else {
                const int rr = R - q_rows - k_rows;
                const int head = rr / D, dd = rr % D;
                u16 *dst = reinterpret_cast<u16 *>(a.kv_table[a.n_layers + a.layer]) + ((size_t)head * cap + pos) * D + dd;
                *reinterpret_cast<u32 *>(dst) = pack2<T>(ra, rb);
            }
        }
    }
}

// Host-side launch: picks the geometry (row lanes, unroll) for (N, K) and dispatches the template.
int w4s_gemv_launch(int dtype, int pro, int epi, GemvArgs &a, int M, hipStream_t stream);
// Row lanes / unroll the launcher will use for an [N,K] weight (tile = 2*row_lanes*unroll rows per workgroup).
int w4s_gemv_geometry(int N, int K, int *row_lanes, int *unroll);
