// vision.hip -- op-level kernels of the Qwen2.5-VL vision tower (reference models/intern/vision.py:87-442).
//
// The tower is dense 16-bit GEMMs (the hand-written MFMA GEMM of w16_gemm.hpp on W16M-tiled weights: pie_linear_w16m, with the bias and
// the MLP's SiLU * up in its epilogue) plus four hand-written pieces: the rotate-half rotary embedding with per-patch (row, column) angles fused with the q / k / v re-layout the attention
// kernel wants, block-diagonal non-causal attention on the MFMA units (prefill_attn.hpp, SEG instantiation: full-image and
// 64-patch-window layers are the same kernel with different segment tables), bias add for any column count, and erf-GELU.
// RMSNorm, SiLU*up and residual adds are the ops the text tower already has (ops.hip).
#include "prefill_attn.hpp"

// w4m_gemm.hip / w16_gemm.hpp
size_t w16m_size(int N, int K);
int w16m_from_rows_launch(const void *w, int N, int K, void *w16m, hipStream_t st);
size_t w16l_workspace_bytes(int M, int N, int K);
int w16l_gemm_launch(int dtype, const void *w16m, const void *x, int ldx, int M, int N, int K, void *y, void *workspace, hipStream_t st, const void *bias,
                     void *swiglu_act, bool *fused, int ldy);

// y[m, n] = T(y[m, n] + b[n]), any N (two columns per thread; rows of odd length end in a single column)
template <class T>
__global__ void k_bias_any(u16 *y, const u16 *b, int M, int N) {
    const int n2 = (N + 1) >> 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * n2) return;
    const int m = (int)(i / n2), n = (int)(i % n2) * 2;
    u16 *p = y + (size_t)m * N + n;
    p[0] = T::from_f32(T::to_f32(p[0]) + T::to_f32(b[n]));
    if (n + 1 < N) p[1] = T::from_f32(T::to_f32(p[1]) + T::to_f32(b[n + 1]));
}

// nn.GELU() (exact): x * (1 + erf(x / sqrt 2)) / 2  (PatchMerger.mlp[1], vision.py:130)
template <class T>
__global__ void k_gelu(const u16 *x, size_t n, u16 *y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = T::to_f32(x[i]);
    y[i] = T::from_f32(v * (1.0f + erff(v * 0.70710678118654752440f)) * 0.5f);
}

// Attention.__call__, vision.py:152-158 + the layout step: qkv [N, 3, H, D] (the qkv Linear's output) ->
//   q [N, H, DP] = rope(q), k [H, N, DP] = rope(k), v [H, N, DP] = v, head dims D..DP-1 zero (DP = 64 or 128: the MFMA
//   attention's head sizes; zero columns change neither the scores nor the output's first D dims).
// rope = apply_rotary_pos_emb_vision (vision.py:55-70): x * cos + rotate_half(x) * sin in fp32, one rounding;
// cos / sin fp32 [N, D/2] (the row's angles, tiled twice over the head dim).  One thread per (row, head, pair d < D/2).
// `bias` (nullable, [3 * H * D]): the qkv Linear's bias, added here -- T(x + b), the Linear's own rounding -- instead of in a pass
// of its own over the GEMM output.
template <class T>
__global__ void k_vision_qkv_rope(const u16 *qkv, const u16 *bias, const float *cs, const float *sn, int N, int H, int D, int DP, u16 *q, u16 *k,
                                  u16 *v) {
    const int half = D >> 1, hp = DP >> 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * H * hp) return;
    const int d = (int)(i % hp), hh = (int)((i / hp) % H), n = (int)(i / ((size_t)hp * H));
    u16 *qo = q + ((size_t)n * H + hh) * DP, *ko = k + ((size_t)hh * N + n) * DP, *vo = v + ((size_t)hh * N + n) * DP;
    if (d >= half) {  // padding pairs: columns D + 2 (d - half), + 1
        const int c0 = D + 2 * (d - half);
        if (c0 + 1 < DP) qo[c0] = qo[c0 + 1] = ko[c0] = ko[c0 + 1] = vo[c0] = vo[c0 + 1] = 0;
        return;
    }
    const u16 *row = qkv + (size_t)n * 3 * H * D + (size_t)hh * D;
    const u16 *brow = bias ? bias + (size_t)hh * D : nullptr;
    auto ld = [&](int part, int col) {  // element `col` of this head's q (0) / k (1) / v (2) row, bias applied
        const float x = T::to_f32(row[(size_t)part * H * D + col]);
        return brow ? round_T<T>(x + T::to_f32(brow[(size_t)part * H * D + col])) : x;
    };
    const float c = cs[(size_t)n * half + d], s = sn[(size_t)n * half + d];
    {
        const float a = ld(0, d), b = ld(0, d + half);
        qo[d] = T::from_f32(__fadd_rn(__fmul_rn(a, c), __fmul_rn(-b, s)));
        qo[d + half] = T::from_f32(__fadd_rn(__fmul_rn(b, c), __fmul_rn(a, s)));
    }
    {
        const float a = ld(1, d), b = ld(1, d + half);
        ko[d] = T::from_f32(__fadd_rn(__fmul_rn(a, c), __fmul_rn(-b, s)));
        ko[d + half] = T::from_f32(__fadd_rn(__fmul_rn(b, c), __fmul_rn(a, s)));
    }
    vo[d] = T::from_f32(ld(2, d)), vo[d + half] = T::from_f32(ld(2, d + half));
}

// The same, four pairs per thread with 8-byte accesses (D % 8 == 0: every head size the tower uses); the scalar kernel above is
// the fallback.  At 4096 patches: 50 -> see DESIGN.md 2d (the scalar form moved 2 bytes per lane per access).
template <class T>
__global__ void k_vision_qkv_rope_v4(const u16 *qkv, const u16 *bias, const float *cs, const float *sn, int N, int H, int D, int DP, u16 *q, u16 *k,
                                     u16 *v) {
    const int half = D >> 1, hp4 = DP >> 3;  // groups of 4 pairs per padded head
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * H * hp4) return;
    const int d0 = (int)(i % hp4) * 4, hh = (int)((i / hp4) % H), n = (int)(i / ((size_t)hp4 * H));
    u16 *qo = q + ((size_t)n * H + hh) * DP, *ko = k + ((size_t)hh * N + n) * DP, *vo = v + ((size_t)hh * N + n) * DP;
    if (d0 >= half) {  // 8 padding columns
        const int c0 = D + 2 * (d0 - half);
        const uint4 z = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(qo + c0) = z, *reinterpret_cast<uint4 *>(ko + c0) = z, *reinterpret_cast<uint4 *>(vo + c0) = z;
        return;
    }
    const u16 *row = qkv + (size_t)n * 3 * H * D + (size_t)hh * D;
    const u16 *brow = bias ? bias + (size_t)hh * D : nullptr;
    const float4 c = *reinterpret_cast<const float4 *>(cs + (size_t)n * half + d0), s = *reinterpret_cast<const float4 *>(sn + (size_t)n * half + d0);
    const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x, s.y, s.z, s.w};
    auto ld4 = [&](int part, int col, float *o) {  // 4 elements of this head's q / k / v row, bias applied
        const uint2 x = *reinterpret_cast<const uint2 *>(row + (size_t)part * H * D + col);
        o[0] = lo_f32<T>(x.x), o[1] = hi_f32<T>(x.x), o[2] = lo_f32<T>(x.y), o[3] = hi_f32<T>(x.y);
        if (brow) {
            const uint2 b = *reinterpret_cast<const uint2 *>(brow + (size_t)part * H * D + col);
            o[0] = round_T<T>(o[0] + lo_f32<T>(b.x)), o[1] = round_T<T>(o[1] + hi_f32<T>(b.x));
            o[2] = round_T<T>(o[2] + lo_f32<T>(b.y)), o[3] = round_T<T>(o[3] + hi_f32<T>(b.y));
        }
    };
    auto st4 = [&](u16 *dst, const float *o) { *reinterpret_cast<uint2 *>(dst) = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])); };
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        float a[4], b[4], lo[4], hi[4];
        ld4(part, d0, a), ld4(part, d0 + half, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = __fadd_rn(__fmul_rn(a[j], cc[j]), __fmul_rn(-b[j], ss[j]));
            hi[j] = __fadd_rn(__fmul_rn(b[j], cc[j]), __fmul_rn(a[j], ss[j]));
        }
        u16 *dst = part == 0 ? qo : ko;
        st4(dst + d0, lo), st4(dst + d0 + half, hi);
    }
    float a[4], b[4];
    ld4(2, d0, a), ld4(2, d0 + half, b);
    st4(vo + d0, a), st4(vo + d0 + half, b);
}

// MLP.__call__ (vision.py:196-197) between the GEMMs: act = T(T(silu(T(g + bg))) * T(u + bu)); two columns per thread.
template <class T>
__global__ void k_bias_silu_mul(const u16 *g, const u16 *u, const u16 *bg, const u16 *bu, int M, int N, int ld, u16 *y) {
    const int n2 = (N + 1) >> 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * n2) return;
    const int n = (int)(i % n2) * 2;
    const size_t o = (i / n2) * (size_t)N + n, oi = (i / n2) * (size_t)ld + n;  // output / input (row stride ld) offsets
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (n + j >= N) break;
        const float gv = round_T<T>(T::to_f32(g[oi + j]) + T::to_f32(bg[n + j])), uv = round_T<T>(T::to_f32(u[oi + j]) + T::to_f32(bu[n + j]));
        y[o + j] = T::from_f32(round_T<T>(gv / (1.0f + expf(-gv))) * uv);
    }
}

// four columns per thread, 8-byte accesses (N % 4 == 0)
template <class T>
__global__ void k_bias_silu_mul_v4(const u16 *g, const u16 *u, const u16 *bg, const u16 *bu, int M, int N, int ld, u16 *y) {
    const int n4 = N >> 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * n4) return;
    const int n = (int)(i % n4) * 4;
    const size_t o = (i / n4) * (size_t)N + n, oi = (i / n4) * (size_t)ld + n;
    const uint2 gv = *reinterpret_cast<const uint2 *>(g + oi), uv = *reinterpret_cast<const uint2 *>(u + oi);
    const uint2 bgv = *reinterpret_cast<const uint2 *>(bg + n), buv = *reinterpret_cast<const uint2 *>(bu + n);
    const float gg[4] = {lo_f32<T>(gv.x), hi_f32<T>(gv.x), lo_f32<T>(gv.y), hi_f32<T>(gv.y)};
    const float uu[4] = {lo_f32<T>(uv.x), hi_f32<T>(uv.x), lo_f32<T>(uv.y), hi_f32<T>(uv.y)};
    const float bgg[4] = {lo_f32<T>(bgv.x), hi_f32<T>(bgv.x), lo_f32<T>(bgv.y), hi_f32<T>(bgv.y)};
    const float buu[4] = {lo_f32<T>(buv.x), hi_f32<T>(buv.x), lo_f32<T>(buv.y), hi_f32<T>(buv.y)};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = round_T<T>(gg[j] + bgg[j]), b = round_T<T>(uu[j] + buu[j]);
        r[j] = round_T<T>(a / (1.0f + expf(-a))) * b;
    }
    *reinterpret_cast<uint2 *>(y + o) = make_uint2(pack2<T>(r[0], r[1]), pack2<T>(r[2], r[3]));
}

// hidden_states + Linear(...) (vision.py:212-218) with the Linear's bias folded in: y = T(x + T(r + b)).
template <class T>
__global__ void k_add_bias(const u16 *x, const u16 *r, const u16 *b, int M, int N, u16 *y) {
    const int n2 = (N + 1) >> 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * n2) return;
    const int n = (int)(i % n2) * 2;
    const size_t o = (i / n2) * (size_t)N + n;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (n + j >= N) break;
        y[o + j] = T::from_f32(T::to_f32(x[o + j]) + round_T<T>(T::to_f32(r[o + j]) + T::to_f32(b[n + j])));
    }
}

// The residual add (with the preceding Linear's bias) and the RMSNorm that follows it in one pass over the row (vision.py:212-218 ->
// :213 / :218's norm): y = T(x + T(r + b)), xn = w * T(y * rsqrt(mean(y^2) + eps)).  One workgroup per row, H % 8 == 0, H <= 8192.
template <class T>
__global__ void __launch_bounds__(256) k_add_bias_rms_norm(const u16 *x, const u16 *r, const u16 *b, const u16 *w, float eps, int H, u16 *y, u16 *xn) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * H;
    constexpr int MAXP = 4;
    uint4 hv[MAXP];
    float ssq = 0.0f;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int i = (threadIdx.x + p * 256) * 8;
        if (i < H) {
            const uint4 a = *reinterpret_cast<const uint4 *>(x + base + i), c = *reinterpret_cast<const uint4 *>(r + base + i);
            const uint4 bb = *reinterpret_cast<const uint4 *>(b + i);
            const u32 av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w};
            u32 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = pack2<T>(lo_f32<T>(av[j]) + round_T<T>(lo_f32<T>(cv[j]) + lo_f32<T>(bv[j])),
                                hi_f32<T>(av[j]) + round_T<T>(hi_f32<T>(cv[j]) + hi_f32<T>(bv[j])));
                const float lo = lo_f32<T>(o[j]), hi = hi_f32<T>(o[j]);
                ssq = fmaf(lo, lo, ssq);
                ssq = fmaf(hi, hi, ssq);
            }
            hv[p] = make_uint4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<uint4 *>(y + base + i) = hv[p];
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
            *reinterpret_cast<uint4 *>(xn + base + i) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <class F16F, class BF16F>
static int by_dt(int dtype, F16F &&f16, BF16F &&bf16, const char *who) {
    if (dtype == PIE_F16) f16();
    else if (dtype == PIE_BF16) bf16();
    else return pie::fail(PIE_E_ARG, std::string(who) + ": dtype must be PIE_BF16 or PIE_F16");
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

// y[m, :] = T(y[m, :] + bias) for any N (w4m_gemm.hip's dense GEMM uses it behind a K split)
int bias_any_launch(int dtype, void *y, const void *bias, int M, int N, hipStream_t st) {
    const size_t n = (size_t)M * ((N + 1) >> 1);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    return by_dt(
        dtype, [&] { hipLaunchKernelGGL(k_bias_any<F16>, grid, block, 0, st, (u16 *)y, (const u16 *)bias, M, N); },
        [&] { hipLaunchKernelGGL(k_bias_any<BF16>, grid, block, 0, st, (u16 *)y, (const u16 *)bias, M, N); }, "bias");
}

extern "C" {

size_t pie_w16m_bytes(int N, int K) { return N > 0 && K > 0 ? w16m_size(N, K) : 0; }

int pie_repack_w16m(const void *w, int N, int K, int dtype, void *w16m, void *stream) {
    PIE_REQUIRE(w && w16m, PIE_E_ARG, "pie_repack_w16m: null pointer");
    PIE_REQUIRE(N > 0 && K > 0, PIE_E_SHAPE, "pie_repack_w16m: empty matrix");
    PIE_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16, PIE_E_ARG, "pie_repack_w16m: dtype must be PIE_BF16 or PIE_F16");
    PIE_REQUIRE(pie_aligned(w, 16) && pie_aligned(w16m, 256), PIE_E_ALIGN, "pie_repack_w16m: w needs 16-byte, w16m 256-byte alignment");
    return w16m_from_rows_launch(w, N, K, w16m, (hipStream_t)stream);
}

size_t pie_linear_w16m_workspace(int M, int N, int K) { return M > 0 && N > 0 && K > 0 ? w16l_workspace_bytes(M, N, K) : 0; }

int pie_linear_w16m(const void *x, int ldx, const void *w16m, const void *bias, int M, int N, int K, int dtype, void *y, int ldy, int swiglu,
                    void *workspace, size_t workspace_bytes, void *stream) {
    PIE_REQUIRE(x && w16m && y, PIE_E_ARG, "pie_linear_w16m: null pointer");
    PIE_REQUIRE(M > 0 && N > 0 && K > 0, PIE_E_SHAPE, "pie_linear_w16m: empty operand");
    PIE_REQUIRE(!swiglu || N % 8 == 0, PIE_E_SHAPE, "pie_linear_w16m: the fused SiLU * up needs N % 8 == 0");
    PIE_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16, PIE_E_ARG, "pie_linear_w16m: dtype must be PIE_BF16 or PIE_F16");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(w16m, 16) && pie_aligned(y, 16) && (!bias || pie_aligned(bias, (N & 3) ? 2 : 8)), PIE_E_ALIGN,
                "pie_linear_w16m: 16-byte alignment required");
    const size_t need = swiglu ? 0 : w16l_workspace_bytes(M, N, K);
    PIE_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), PIE_E_ARG, "pie_linear_w16m: workspace smaller than pie_linear_w16m_workspace(M, N, K)");
    return w16l_gemm_launch(dtype, w16m, x, ldx, M, N, K, swiglu ? nullptr : y, workspace, (hipStream_t)stream, bias, swiglu ? y : nullptr, nullptr, ldy);
}

int pie_gelu(const void *x, size_t n, int dtype, void *y, void *stream) {
    PIE_REQUIRE(x && y, PIE_E_ARG, "pie_gelu: null pointer");
    PIE_REQUIRE(n > 0, PIE_E_SHAPE, "pie_gelu: empty input");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    return by_dt(
        dtype, [&] { hipLaunchKernelGGL(k_gelu<F16>, grid, block, 0, st, (const u16 *)x, n, (u16 *)y); },
        [&] { hipLaunchKernelGGL(k_gelu<BF16>, grid, block, 0, st, (const u16 *)x, n, (u16 *)y); }, "pie_gelu");
}

int pie_vision_qkv_rope(const void *qkv, const void *bias, const float *cos_t, const float *sin_t, int N, int H, int D, int DP, int dtype, void *q,
                        void *k, void *v, void *stream) {
    PIE_REQUIRE(qkv && cos_t && sin_t && q && k && v, PIE_E_ARG, "pie_vision_qkv_rope: null pointer");
    PIE_REQUIRE(N > 0 && H > 0 && D > 0 && D % 2 == 0 && (DP == 64 || DP == 128) && D <= DP, PIE_E_SHAPE,
                "pie_vision_qkv_rope: head_dim must be even and fit the padded size 64 or 128");
    hipStream_t st = (hipStream_t)stream;
    if (D % 8 == 0 && pie_aligned(qkv, 8) && pie_aligned(q, 16) && pie_aligned(k, 16) && pie_aligned(v, 16) && (!bias || pie_aligned(bias, 8)) &&
        pie_aligned(cos_t, 16) && pie_aligned(sin_t, 16)) {
        const size_t n4 = (size_t)N * H * (DP / 8);
        const dim3 grid4((unsigned)((n4 + 255) / 256)), block4(256);
        return by_dt(
            dtype,
            [&] { hipLaunchKernelGGL(k_vision_qkv_rope_v4<F16>, grid4, block4, 0, st, (const u16 *)qkv, (const u16 *)bias, cos_t, sin_t, N, H, D, DP, (u16 *)q, (u16 *)k, (u16 *)v); },
            [&] { hipLaunchKernelGGL(k_vision_qkv_rope_v4<BF16>, grid4, block4, 0, st, (const u16 *)qkv, (const u16 *)bias, cos_t, sin_t, N, H, D, DP, (u16 *)q, (u16 *)k, (u16 *)v); },
            "pie_vision_qkv_rope");
    }
    const size_t n = (size_t)N * H * (DP / 2);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    return by_dt(
        dtype,
        [&] { hipLaunchKernelGGL(k_vision_qkv_rope<F16>, grid, block, 0, st, (const u16 *)qkv, (const u16 *)bias, cos_t, sin_t, N, H, D, DP, (u16 *)q, (u16 *)k, (u16 *)v); },
        [&] { hipLaunchKernelGGL(k_vision_qkv_rope<BF16>, grid, block, 0, st, (const u16 *)qkv, (const u16 *)bias, cos_t, sin_t, N, H, D, DP, (u16 *)q, (u16 *)k, (u16 *)v); },
        "pie_vision_qkv_rope");
}

int pie_bias_silu_mul(const void *gate, const void *up, const void *bias_gate, const void *bias_up, int M, int N, int ld, int dtype, void *y,
                      void *stream) {
    PIE_REQUIRE(gate && up && bias_gate && bias_up && y, PIE_E_ARG, "pie_bias_silu_mul: null pointer");
    PIE_REQUIRE(M > 0 && N > 0, PIE_E_SHAPE, "pie_bias_silu_mul: empty input");
    if (ld == 0) ld = N;
    PIE_REQUIRE(ld >= N, PIE_E_SHAPE, "pie_bias_silu_mul: row stride smaller than the row");
    hipStream_t st = (hipStream_t)stream;
    if (N % 4 == 0 && ld % 4 == 0 && pie_aligned(gate, 8) && pie_aligned(up, 8) && pie_aligned(bias_gate, 8) && pie_aligned(bias_up, 8) && pie_aligned(y, 8)) {
        const size_t n4 = (size_t)M * (N / 4);
        const dim3 grid4((unsigned)((n4 + 255) / 256)), block4(256);
        return by_dt(
            dtype,
            [&] { hipLaunchKernelGGL(k_bias_silu_mul_v4<F16>, grid4, block4, 0, st, (const u16 *)gate, (const u16 *)up, (const u16 *)bias_gate, (const u16 *)bias_up, M, N, ld, (u16 *)y); },
            [&] { hipLaunchKernelGGL(k_bias_silu_mul_v4<BF16>, grid4, block4, 0, st, (const u16 *)gate, (const u16 *)up, (const u16 *)bias_gate, (const u16 *)bias_up, M, N, ld, (u16 *)y); },
            "pie_bias_silu_mul");
    }
    const size_t n = (size_t)M * ((N + 1) >> 1);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    return by_dt(
        dtype,
        [&] { hipLaunchKernelGGL(k_bias_silu_mul<F16>, grid, block, 0, st, (const u16 *)gate, (const u16 *)up, (const u16 *)bias_gate, (const u16 *)bias_up, M, N, ld, (u16 *)y); },
        [&] { hipLaunchKernelGGL(k_bias_silu_mul<BF16>, grid, block, 0, st, (const u16 *)gate, (const u16 *)up, (const u16 *)bias_gate, (const u16 *)bias_up, M, N, ld, (u16 *)y); },
        "pie_bias_silu_mul");
}

int pie_add_bias(const void *x, const void *r, const void *bias, int M, int N, int dtype, void *y, void *stream) {
    PIE_REQUIRE(x && r && bias && y, PIE_E_ARG, "pie_add_bias: null pointer");
    PIE_REQUIRE(M > 0 && N > 0, PIE_E_SHAPE, "pie_add_bias: empty input");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)M * ((N + 1) >> 1);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    return by_dt(
        dtype, [&] { hipLaunchKernelGGL(k_add_bias<F16>, grid, block, 0, st, (const u16 *)x, (const u16 *)r, (const u16 *)bias, M, N, (u16 *)y); },
        [&] { hipLaunchKernelGGL(k_add_bias<BF16>, grid, block, 0, st, (const u16 *)x, (const u16 *)r, (const u16 *)bias, M, N, (u16 *)y); }, "pie_add_bias");
}

int pie_add_bias_rms_norm(const void *x, const void *r, const void *bias, const void *norm_w, float eps, int M, int N, int dtype, void *y, void *xn,
                          void *stream) {
    PIE_REQUIRE(x && r && bias && norm_w && y && xn, PIE_E_ARG, "pie_add_bias_rms_norm: null pointer");
    PIE_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && N <= 8192, PIE_E_SHAPE, "pie_add_bias_rms_norm: N must be a multiple of 8, at most 8192");
    PIE_REQUIRE(pie_aligned(x, 16) && pie_aligned(r, 16) && pie_aligned(bias, 16) && pie_aligned(norm_w, 16) && pie_aligned(y, 16) && pie_aligned(xn, 16),
                PIE_E_ALIGN, "pie_add_bias_rms_norm: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    return by_dt(
        dtype,
        [&] { hipLaunchKernelGGL(k_add_bias_rms_norm<F16>, dim3(M), dim3(256), 0, st, (const u16 *)x, (const u16 *)r, (const u16 *)bias, (const u16 *)norm_w, eps, N, (u16 *)y, (u16 *)xn); },
        [&] { hipLaunchKernelGGL(k_add_bias_rms_norm<BF16>, dim3(M), dim3(256), 0, st, (const u16 *)x, (const u16 *)r, (const u16 *)bias, (const u16 *)norm_w, eps, N, (u16 *)y, (u16 *)xn); },
        "pie_add_bias_rms_norm");
}

int pie_sdpa_segments(const void *q, const void *k, const void *v, const int32_t *seg_lo, const int32_t *seg_hi, int N, int H, int D, float scale,
                      int dtype, void *out, void *stream) {
    PIE_REQUIRE(q && k && v && seg_lo && seg_hi && out, PIE_E_ARG, "pie_sdpa_segments: null pointer");
    PIE_REQUIRE(N > 0 && H > 0 && (D == 64 || D == 128), PIE_E_SHAPE, "pie_sdpa_segments: head_dim must be 64 or 128");
    PIE_REQUIRE(pie_aligned(q, 16) && pie_aligned(k, 16) && pie_aligned(v, 16) && pie_aligned(out, 8), PIE_E_ALIGN, "pie_sdpa_segments: misaligned pointer");
    PrefillAttnArgs a = {};
    a.q = (const u16 *)q, a.k = (const u16 *)k, a.v = (const u16 *)v, a.offset = 0, a.cap = N;
    a.seg_lo = seg_lo, a.seg_hi = seg_hi;
    a.M = N, a.Hq = H, a.Hkv = H, a.scale = scale, a.out = (u16 *)out;
    if (dtype == PIE_BF16) return segment_attn_launch_t<BF16>(a, D, (hipStream_t)stream);
    if (dtype == PIE_F16) return segment_attn_launch_t<F16>(a, D, (hipStream_t)stream);
    return pie::fail(PIE_E_ARG, "pie_sdpa_segments: dtype must be PIE_BF16 or PIE_F16");
}

}  // extern "C"
