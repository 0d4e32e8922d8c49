/*
 * pie_oracle.c -- CPU ORACLE for the PIE decode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (proxy_inference_engine_amd) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (TheProxyCompany/proxy-inference-engine @ 2025-05-09)
 * holds no golden vector, known-answer test or fixture for this path (its tests are an
 * import smoke test and page-allocator unit tests), its arithmetic lives in the
 * un-vendored third-party dependency MLX (`mlx`, `mlx_lm`: unpinned in pyproject.toml:29-30,
 * contemporaneous release line 0.25.x) and neither the reference nor MLX can run in the
 * build container.  This file restates the reference's algorithm from its call sites and
 * from MLX's published op contracts; it is cross-checked against an
 * independent implementation (HF transformers LlamaForCausalLM on torch-CPU) and closed-form
 * known answers in tests/, which is weaker than a reference-pinned oracle.
 *
 * Conventions
 *   - Activations travel as float arrays whose values are exactly representable in the
 *     activation dtype T (PIE_F32 / PIE_BF16 / PIE_F16); `rnd(v, T)` is applied wherever
 *     MLX materialises an array (every op boundary), math inside an op is fp32.
 *   - Parameters (scales, biases, norm weights, dense weights) are stored in T
 *     (uint16_t for bf16/f16, float for f32) and read through ldT().
 *   - Each function cites the reference file:line it follows (paths relative to
 *     /root/reference/src/proxy_inference_engine/).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { PIE_F32 = 0, PIE_BF16 = 1, PIE_F16 = 2 };

/* ------------------------------------------------------------------ dtype helpers */

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
    u += 0x7fffu + ((u >> 16) & 1u);                                           /* RNE */
    return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u2f(sign);
        float v = (float)man * 5.9604644775390625e-08f; /* 2^-24 */
        return sign ? -v : v;
    }
    if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
    return u2f(sign | ((exp + 112u) << 23) | (man << 13));
}
static inline uint16_t f32_to_f16(float f) {
    uint32_t u = f2u(f), sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (a >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (a < 0x38800000u) { /* subnormal half or zero: value * 2^24 rounded RNE */
        float v = u2f(a) * 16777216.0f; /* exact scaling */
        float r = nearbyintf(v);        /* default rounding mode = RNE */
        return (uint16_t)(sign | (uint32_t)r);
    }
    uint32_t m = a - 0x38000000u; /* rebias exponent */
    m += 0xfffu + ((m >> 13) & 1u);
    return (uint16_t)(sign | (m >> 13));
}

float orc_round(float v, int dtype) {
    if (dtype == PIE_BF16) return bf16_to_f32(f32_to_bf16(v));
    if (dtype == PIE_F16) return f16_to_f32(f32_to_f16(v));
    return v;
}
#define rnd orc_round

static inline float ldT(const void *p, size_t i, int dtype) {
    if (dtype == PIE_BF16) return bf16_to_f32(((const uint16_t *)p)[i]);
    if (dtype == PIE_F16) return f16_to_f32(((const uint16_t *)p)[i]);
    return ((const float *)p)[i];
}
static inline void stT(void *p, size_t i, float v, int dtype) {
    if (dtype == PIE_BF16) ((uint16_t *)p)[i] = f32_to_bf16(v);
    else if (dtype == PIE_F16) ((uint16_t *)p)[i] = f32_to_f16(v);
    else ((float *)p)[i] = v;
}

/* bulk converters used by the python wrapper */
void orc_to_T(const float *src, void *dst, size_t n, int dtype) {
    for (size_t i = 0; i < n; ++i) stT(dst, i, src[i], dtype);
}
void orc_from_T(const void *src, float *dst, size_t n, int dtype) {
    for (size_t i = 0; i < n; ++i) dst[i] = ldT(src, i, dtype);
}
void orc_round_inplace(float *x, size_t n, int dtype) {
    for (size_t i = 0; i < n; ++i) x[i] = rnd(x[i], dtype);
}
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Code k of a packed row.  MLX packs a row's codes as a little-endian bit stream, code k at bits [k*bits, (k+1)*bits): for bits 2 / 4 / 8
 * that is word k / (32/bits) at bit bits * (k % (32/bits)) (mlx/backend/common/quantized.cpp); for bits 3 and 6 codes straddle bytes
 * (8 codes per 3 bytes, 4 codes per 3 bytes: the `bits == 3 || bits == 6` byte-packed layouts of mlx >= 0.21).  A row takes K*bits/32 words. */
static inline uint32_t orc_code(const uint32_t *row, size_t k, int bits) {
    const uint8_t *b = (const uint8_t *)row;
    const size_t bit = k * (size_t)bits, byte = bit >> 3;
    const unsigned sh = (unsigned)(bit & 7);
    uint32_t v = b[byte];
    if (sh + (unsigned)bits > 8) v |= (uint32_t)b[byte + 1] << 8;
    return (v >> sh) & ((1u << bits) - 1u);
}
static inline void orc_put_code(uint32_t *row, size_t k, int bits, uint32_t c) {  /* into a zeroed row */
    uint8_t *b = (uint8_t *)row;
    const size_t bit = k * (size_t)bits, byte = bit >> 3;
    const unsigned sh = (unsigned)(bit & 7);
    const uint32_t v = c << sh;
    b[byte] |= (uint8_t)v;
    if (sh + (unsigned)bits > 8) b[byte + 1] |= (uint8_t)(v >> 8);
}

/* ------------------------------------------------------------------ A.1 mx.quantize / mx.dequantize
 * Reference call sites: cache/kv_cache/cache.py:144-147, quantized.py:91-96; implicitly the
 * checkpoint format consumed by models/utils.py:96-111 (weight uint32 [N,K/8], scales/biases T [N,K/G]).
 * Per group: w_max,w_min; side=|w_min|>|w_max|; scale=max((w_max-w_min)/n_bins,eps), negated
 * unless side; edge=side?w_min:w_max; q0=round(edge/scale); scale=q0!=0?edge/q0:scale;
 * bias=q0==0?0:edge; code=clip(round((w-bias)/scale),0,n_bins) with the UNROUNDED fp32 scale/bias;
 * scales/biases stored in T.  Packing: code j of a row in word j/8 at bits 4*(j%8) (bits=4).
 */
void orc_quantize(const float *w, int N, int K, int group_size, int bits, int dtype,
                  uint32_t *wq, void *scales, void *biases) {
    const int G = K / group_size;
    const size_t row_words = (size_t)K * bits / 32;
    const float n_bins = (float)((1 << bits) - 1), eps = 1e-7f;
    memset(wq, 0, (size_t)N * row_words * sizeof(uint32_t));
    for (int n = 0; n < N; ++n)
        for (int g = 0; g < G; ++g) {
            const float *wg = w + (size_t)n * K + (size_t)g * group_size;
            float w_max = wg[0], w_min = wg[0];
            for (int j = 1; j < group_size; ++j) {
                if (wg[j] > w_max) w_max = wg[j];
                if (wg[j] < w_min) w_min = wg[j];
            }
            int side = fabsf(w_min) > fabsf(w_max);
            float scale = fmaxf((w_max - w_min) / n_bins, eps);
            scale = side ? scale : -scale;
            float edge = side ? w_min : w_max;
            float q0 = rintf(edge / scale);
            int at_zero = (q0 == 0.0f);
            scale = at_zero ? scale : edge / q0;
            float bias = at_zero ? 0.0f : edge;
            stT(scales, (size_t)n * G + g, scale, dtype);
            stT(biases, (size_t)n * G + g, bias, dtype);
            for (int j = 0; j < group_size; ++j) {
                float c = rintf((wg[j] - bias) / scale);
                if (c < 0.0f) c = 0.0f;
                if (c > n_bins) c = n_bins;
                int k = g * group_size + j;
                orc_put_code(wq + (size_t)n * row_words, (size_t)k, bits, (uint32_t)c);
            }
        }
}

/* w_hat = scale*q + bias, materialised in T (mx.dequantize output dtype = scales dtype). */
void orc_dequantize(const uint32_t *wq, const void *scales, const void *biases, int N, int K,
                    int group_size, int bits, int dtype, float *out) {
    const int G = K / group_size;
    const size_t row_words = (size_t)K * bits / 32;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            float q = (float)orc_code(wq + (size_t)n * row_words, (size_t)k, bits);
            float s = ldT(scales, (size_t)n * G + k / group_size, dtype);
            float b = ldT(biases, (size_t)n * G + k / group_size, dtype);
            out[(size_t)n * K + k] = rnd(s * q + b, dtype);
        }
}

/* ------------------------------------------------------------------ A.2 mx.quantized_matmul(transpose=True)
 * Reference call sites: nn.QuantizedLinear installed by nn.quantize (models/utils.py:111), invoked at
 * models/llama/language.py:83 (q,k,v), :108 (o), :127 (gate,up,down), :207/:209 (lm_head / tied as_linear).
 * y[m,n] = T( sum_k x[m,k] * (scale[n,k/G]*q[n,k] + bias[n,k/G]) ), fp32 accumulate in k order.
 * `lin_bias` (nn.QuantizedLinear's optional bias, attention_bias/mlp_bias) is added afterwards in T.
 */
void orc_quantized_matmul_t(const float *x, int M, const uint32_t *wq, const void *scales,
                            const void *biases, int N, int K, int group_size, int bits,
                            int dtype, const void *lin_bias, float *y) {
    const int G = K / group_size;
    const size_t row_words = (size_t)K * bits / 32;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const uint32_t *wrow = wq + (size_t)n * row_words;
        for (int m = 0; m < M; ++m) {
            const float *xr = x + (size_t)m * K;
            float acc = 0.0f;
            for (int g = 0; g < G; ++g) {
                const float s = ldT(scales, (size_t)n * G + g, dtype);
                const float b = ldT(biases, (size_t)n * G + g, dtype);
                const float *xs = xr + g * group_size;
                for (int p = 0; p < group_size; ++p) {
                    float wv = s * (float)orc_code(wrow, (size_t)g * group_size + p, bits) + b;
                    acc += xs[p] * wv;
                }
            }
            float out = rnd(acc, dtype);
            if (lin_bias) out = rnd(out + ldT(lin_bias, n, dtype), dtype);
            y[(size_t)m * N + n] = out;
        }
    }
}

/* nn.Linear: x @ W.T (+ b), W [N,K] in T  (models/llama/language.py:83,108,127,209 when the
 * checkpoint has no "quantization" entry: models/utils.py:96-97). */
void orc_linear(const float *x, int M, const void *w, int N, int K, int dtype, const void *lin_bias,
                float *y) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc += x[(size_t)m * K + k] * ldT(w, (size_t)n * K + k, dtype);
            float out = rnd(acc, dtype);
            if (lin_bias) out = rnd(out + ldT(lin_bias, n, dtype), dtype);
            y[(size_t)m * N + n] = out;
        }
}

/* nn.Embedding / nn.QuantizedEmbedding (models/llama/language.py:176): row gather, dequantised in T. */
void orc_embedding(const int32_t *ids, int L, const void *w, const void *scales, const void *biases,
                   int quantized, int H, int group_size, int bits, int dtype, float *out) {
    for (int l = 0; l < L; ++l) {
        size_t row = (size_t)ids[l];
        if (!quantized) {
            for (int k = 0; k < H; ++k) out[(size_t)l * H + k] = ldT(w, row * H + k, dtype);
        } else {
            const int G = H / group_size;
            const uint32_t *wq = (const uint32_t *)w + row * ((size_t)H * bits / 32);
            orc_dequantize(wq, (const char *)scales + row * G * (dtype == PIE_F32 ? 4 : 2),
                           (const char *)biases + row * G * (dtype == PIE_F32 ? 4 : 2), 1, H,
                           group_size, bits, dtype, out + (size_t)l * H);
        }
    }
}

/* ------------------------------------------------------------------ A.3 mx.fast.rms_norm
 * Call sites: nn.RMSNorm at models/llama/language.py:137-141 (input/post-attention norms), :168 (final).
 * y = w * T( x32 * rsqrt(mean(x32^2) + eps) ), product rounded to T again.
 */
void orc_rms_norm(const float *x, int rows, int H, const void *w, float eps, int dtype, float *y) {
    for (int r = 0; r < rows; ++r) {
        const float *xr = x + (size_t)r * H;
        float ss = 0.0f;
        for (int k = 0; k < H; ++k) ss += xr[k] * xr[k];
        float inv = 1.0f / sqrtf(ss / (float)H + eps);
        for (int k = 0; k < H; ++k) {
            float nx = rnd(xr[k] * inv, dtype);
            y[(size_t)r * H + k] = rnd(ldT(w, k, dtype) * nx, dtype);
        }
    }
}

/* ------------------------------------------------------------------ A.4 mx.fast.rope (traditional=False)
 * Call site: models/llama/utils.py:42-50 <- models/llama/language.py:91-92 (offset = cache.offset).
 * x [heads, L, D]; position p = offset + l; theta_i = p * (1/freqs[i]); rotate-half pairs (i, i+D/2).
 */
/* traditional != 0: mx.fast.rope(traditional=True), ModelArgs.rope_traditional (language.py:27,69): the rotated pairs are
 * the interleaved (2i, 2i+1) instead of (i, i+D/2); same angles. */
void orc_rope_ex(const float *x, int heads, int L, int D, const float *freqs, int offset, int dtype,
                 int traditional, float *y) {
    const int half = D / 2;
    for (int h = 0; h < heads; ++h)
        for (int l = 0; l < L; ++l) {
            const float *xr = x + ((size_t)h * L + l) * D;
            float *yr = y + ((size_t)h * L + l) * D;
            float p = (float)(offset + l);
            for (int i = 0; i < half; ++i) {
                float theta = p * (1.0f / freqs[i]);
                float c = cosf(theta), s = sinf(theta);
                const int i0 = traditional ? 2 * i : i, i1 = traditional ? 2 * i + 1 : i + half;
                float a = xr[i0], b = xr[i1];
                yr[i0] = rnd(a * c - b * s, dtype);
                yr[i1] = rnd(a * s + b * c, dtype);
            }
        }
}
void orc_rope(const float *x, int heads, int L, int D, const float *freqs, int offset, int dtype,
              float *y) {
    orc_rope_ex(x, heads, L, D, freqs, offset, dtype, 0, y);
}

/* Llama3RoPE.__init__ (models/llama/utils.py:22-39): scaled frequency table, fp32 like mx.arange math.
 * With factor = low = high = 1 (rope_scaling=None, models/llama/language.py:59-63) this is base^(2i/D). */
void orc_llama3_rope_freqs(int D, float base, float max_len, float global_len, float factor,
                           float low_freq_factor, float high_freq_factor, float *freqs) {
    const float two_pi = 6.283185307179586f;
    float low_wl = global_len / low_freq_factor, high_wl = global_len / high_freq_factor;
    for (int i = 0; i < D / 2; ++i) {
        float f = powf(base, (float)(2 * i) / (float)D);
        float wl = two_pi * f;
        float f1 = wl > low_wl ? f * factor : f;
        int medium = (wl > high_wl) && (wl < low_wl);
        if (medium) {
            float smooth = (max_len / wl - low_freq_factor) / (high_freq_factor - low_freq_factor);
            f1 = f1 / ((1.0f - smooth) / factor + smooth);
        }
        freqs[i] = f1;
    }
}

/* ------------------------------------------------------------------ A.5 mx.fast.scaled_dot_product_attention
 * Call site: models/base.py:111-113 <- models/llama/language.py:98-105.
 * q [Hq, L, D], k/v [Hkv, cap, D] of which the first T positions are valid (the strided view
 * cache/kv_cache/reusable.py:142 returns), mask additive [L, T] in T or NULL, out [Hq, L, D].
 * fused=1: MLX fused-kernel contract (fp32 scores, fp32 softmax, fp32 PV, one rounding at the end).
 * fused=0: MLX's unfused fallback graph (each primitive materialised in T): T(q*scale), T(QK^T),
 *          T(+mask), T(softmax, computed in fp32), T(PV).  Provided to bound the distance between
 *          the two published behaviours; the HIP kernel implements fused=1.
 */
void orc_sdpa(const float *q, const float *k, const float *v, int Hq, int Hkv, int L, int T, int cap,
              int D, float scale, const float *mask, int dtype, int fused, float *out) {
    const int rep = Hq / Hkv;
#pragma omp parallel for schedule(static)
    for (int h = 0; h < Hq; ++h) {
        const float *kh = k + (size_t)(h / rep) * cap * D;
        const float *vh = v + (size_t)(h / rep) * cap * D;
        float *s = (float *)malloc(sizeof(float) * (size_t)T);
        float *qs = (float *)malloc(sizeof(float) * (size_t)D);
        for (int l = 0; l < L; ++l) {
            const float *qr = q + ((size_t)h * L + l) * D;
            for (int d = 0; d < D; ++d) {
                float v0 = scale * qr[d];
                qs[d] = fused ? v0 : rnd(rnd(scale, dtype) * qr[d], dtype);
            }
            float mx = -INFINITY;
            for (int t = 0; t < T; ++t) {
                float acc = 0.0f;
                for (int d = 0; d < D; ++d) acc += qs[d] * kh[(size_t)t * D + d];
                if (!fused) acc = rnd(acc, dtype);
                if (mask) {
                    acc += mask[(size_t)l * T + t];
                    if (!fused) acc = rnd(acc, dtype);
                }
                s[t] = acc;
                if (acc > mx) mx = acc;
            }
            float den = 0.0f;
            for (int t = 0; t < T; ++t) {
                s[t] = expf(s[t] - mx);
                den += s[t];
            }
            float *o = out + ((size_t)h * L + l) * D;
            if (fused) {
                for (int d = 0; d < D; ++d) {
                    float acc = 0.0f;
                    for (int t = 0; t < T; ++t) acc += s[t] * vh[(size_t)t * D + d];
                    o[d] = rnd(acc / den, dtype);
                }
            } else {
                for (int t = 0; t < T; ++t) s[t] = rnd(s[t] / den, dtype);
                for (int d = 0; d < D; ++d) {
                    float acc = 0.0f;
                    for (int t = 0; t < T; ++t) acc += s[t] * vh[(size_t)t * D + d];
                    o[d] = rnd(acc, dtype);
                }
            }
        }
        free(s);
        free(qs);
    }
}

/* create_causal_mask / create_attention_mask (models/base.py:18-53): additive (l<r)*-1e9 cast to T,
 * rows l = offset..offset+L-1, cols r = 0..offset+L-1.  Only built when L > 1. */
void orc_causal_mask(int L, int offset, int dtype, float *mask) {
    const int T = offset + L;
    for (int l = 0; l < L; ++l)
        for (int r = 0; r < T; ++r) mask[(size_t)l * T + r] = rnd((offset + l) < r ? -1e9f : 0.0f, dtype);
}

/* nn.silu(a) * b (models/llama/language.py:127): silu = a*sigmoid(a) is one (compiled) op, the
 * multiply a second op -> two roundings. */
void orc_silu_mul(const float *a, const float *b, size_t n, int dtype, float *y) {
    for (size_t i = 0; i < n; ++i) {
        float s = rnd(a[i] / (1.0f + expf(-a[i])), dtype);
        y[i] = rnd(s * b[i], dtype);
    }
}
void orc_add(const float *a, const float *b, size_t n, int dtype, float *y) {
    for (size_t i = 0; i < n; ++i) y[i] = rnd(a[i] + b[i], dtype);
}

/* Tail of _inference (engine/inference_engine.py:268-271) + greedy sampler (samplers/__init__.py:37-38):
 * logprobs = f32(logits) - logsumexp(f32(logits)); argmax returns the FIRST maximal index. */
int orc_logprobs_argmax(const float *logits, int V, float *logprobs) {
    float mx = -INFINITY;
    for (int i = 0; i < V; ++i) if (logits[i] > mx) mx = logits[i];
    float den = 0.0f;
    for (int i = 0; i < V; ++i) den += expf(logits[i] - mx);
    float lse = mx + logf(den);
    int best = 0;
    for (int i = 0; i < V; ++i) {
        logprobs[i] = logits[i] - lse;
        if (logprobs[i] > logprobs[best]) best = i;
    }
    return best;
}

/* ------------------------------------------------------------------ the model graph
 * models/llama/language.py (whole file) over per-layer KV buffers laid out like
 * cache/kv_cache/reusable.py ([n_kv, cap, D], write at [offset, offset+L), attend over [0, offset+L)).
 */
typedef struct {
    const void *w;      /* uint32 codes [N,K/8] when quantized, else T [N,K] */
    const void *scales; /* T [N,K/G] or NULL */
    const void *biases; /* T [N,K/G] or NULL */
    const void *lin_bias; /* T [N] or NULL (attention_bias / mlp_bias) */
} orc_linear_t;

typedef struct {
    int dtype, hidden, n_layers, n_heads, n_kv_heads, head_dim, inter, vocab;
    int group_size, bits, quantized, tie_word_embeddings;
    float eps;
    const float *rope_freqs;          /* [head_dim/2] */
    const void *const *attn_norm;     /* [n_layers] -> T[hidden] */
    const void *const *mlp_norm;      /* [n_layers] */
    const orc_linear_t *q, *k, *v, *o, *gate, *up, *down; /* [n_layers] each */
    orc_linear_t embed;               /* [vocab, hidden] */
    const void *final_norm;
    orc_linear_t lm_head;             /* unused when tied */
    int rope_traditional;             /* ModelArgs.rope_traditional (language.py:27,69) */
} orc_llama_t;

/* mx.quantized_matmul has two regimes in MLX (mlx/backend/metal/quantized.cpp + kernels/quantized.h, dependency pinned
 * only as "mlx" in pyproject.toml:29): few rows -> qmv kernels, the exact fp32 affine sum of orc_quantized_matmul_t;
 * many rows (prompt processing) -> qmm kernels, whose block loader DEQUANTISES the weights to T (dequantize<T>) and feeds
 * a T x T -> fp32 MMA.  g_qmm_min_rows is the row count from which the second form is used (MLX's own switch-over is
 * device dependent, 6..32 rows; 6 mirrors the product's prefill threshold; 0 = always the exact form). */
static int g_qmm_min_rows = 6;
void orc_set_qmm_min_rows(int n) { g_qmm_min_rows = n < 0 ? 0 : n; }
int orc_get_qmm_min_rows(void) { return g_qmm_min_rows; }

/* y = T( x @ mx.dequantize(w).T ) with fp32 accumulation: the qmm regime. */
void orc_quantized_matmul_dequant(const float *x, int M, const uint32_t *wq, const void *scales, const void *biases,
                                  int N, int K, int group_size, int bits, int dtype, const void *lin_bias, float *y) {
    const int G = K / group_size;
    const size_t row_words = (size_t)K * bits / 32;
#pragma omp parallel
    {
        float *wrow = malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            for (int k = 0; k < K; ++k) {
                float q = (float)orc_code(wq + (size_t)n * row_words, (size_t)k, bits);
                float s = ldT(scales, (size_t)n * G + k / group_size, dtype);
                float b = ldT(biases, (size_t)n * G + k / group_size, dtype);
                wrow[k] = rnd(s * q + b, dtype);
            }
            for (int m = 0; m < M; ++m) {
                const float *xr = x + (size_t)m * K;
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc += xr[k] * wrow[k];
                float out = rnd(acc, dtype);
                if (lin_bias) out = rnd(out + ldT(lin_bias, n, dtype), dtype);
                y[(size_t)m * N + n] = out;
            }
        }
        free(wrow);
    }
}

static void lin(const orc_llama_t *m, const orc_linear_t *p, const float *x, int M, int N, int K,
                float *y) {
    if (m->quantized && p->scales && g_qmm_min_rows > 0 && M >= g_qmm_min_rows)
        orc_quantized_matmul_dequant(x, M, (const uint32_t *)p->w, p->scales, p->biases, N, K, m->group_size,
                                     m->bits, m->dtype, p->lin_bias, y);
    else if (m->quantized && p->scales)
        orc_quantized_matmul_t(x, M, (const uint32_t *)p->w, p->scales, p->biases, N, K, m->group_size,
                               m->bits, m->dtype, p->lin_bias, y);
    else
        orc_linear(x, M, p->w, N, K, m->dtype, p->lin_bias, y);
}

/* [L, heads*D] -> [heads, L, D]  (reshape(B,L,h,-1).transpose(0,2,1,3), language.py:86-88) */
static void to_heads(const float *x, int L, int heads, int D, float *y) {
    for (int l = 0; l < L; ++l)
        for (int h = 0; h < heads; ++h)
            memcpy(y + ((size_t)h * L + l) * D, x + ((size_t)l * heads + h) * D, sizeof(float) * D);
}
static void from_heads(const float *x, int L, int heads, int D, float *y) {
    for (int l = 0; l < L; ++l)
        for (int h = 0; h < heads; ++h)
            memcpy(y + ((size_t)l * heads + h) * D, x + ((size_t)h * L + l) * D, sizeof(float) * D);
}

/* Model.__call__ (language.py:199-210) for batch 1.
 * ids [L]; kcache/vcache: n_layers pointers to float [n_kv, cap, D]; offset = cache.offset before the call.
 * logits: [L, vocab] (last_only=0, the reference's behaviour, language.py:205-209) or [vocab] for the
 * final position only (last_only=1: same values for that row; used by the timing leg).
 * sdpa_fused selects the attention contract (see orc_sdpa).  hidden_out (optional) [L, hidden] receives
 * the residual stream after the last block (before the final norm) for layer-wise checks.
 * Returns 0, or -1 on allocation failure / capacity overflow.
 */
int orc_llama_forward_ex(const orc_llama_t *m, const int32_t *ids, const float *embeds, int L, float *const *kcache,
                         float *const *vcache, int cap, int offset, int last_only, int sdpa_fused,
                         float *logits, float *hidden_out);

int orc_llama_forward(const orc_llama_t *m, const int32_t *ids, int L, float *const *kcache,
                      float *const *vcache, int cap, int offset, int last_only, int sdpa_fused,
                      float *logits, float *hidden_out) {
    return orc_llama_forward_ex(m, ids, NULL, L, kcache, vcache, cap, offset, last_only, sdpa_fused, logits, hidden_out);
}

/* embeds != NULL: [L, hidden] input embeddings used instead of embed_tokens(ids) -- `h = inputs_embeds`,
 * models/intern/language.py:155-158 (the VLM ensemble's merged text + image features, intern/ensemble.py:106-108). */
int orc_llama_forward_ex(const orc_llama_t *m, const int32_t *ids, const float *embeds, int L, float *const *kcache,
                         float *const *vcache, int cap, int offset, int last_only, int sdpa_fused,
                         float *logits, float *hidden_out) {
    const int H = m->hidden, D = m->head_dim, nh = m->n_heads, nkv = m->n_kv_heads, I = m->inter;
    const int dt = m->dtype, T = offset + L;
    if (T > cap) return -1;
    const float scale = 1.0f / sqrtf((float)D); /* head_dim**-0.5, language.py:41 */
    size_t big = (size_t)L * (size_t)(I > nh * D ? I : nh * D);
    float *h = malloc(sizeof(float) * (size_t)L * H), *xn = malloc(sizeof(float) * (size_t)L * H);
    float *q = malloc(sizeof(float) * (size_t)L * nh * D), *qh = malloc(sizeof(float) * (size_t)L * nh * D);
    float *kk = malloc(sizeof(float) * (size_t)L * nkv * D), *vv = malloc(sizeof(float) * (size_t)L * nkv * D);
    float *kh = malloc(sizeof(float) * (size_t)L * nkv * D), *vh = malloc(sizeof(float) * (size_t)L * nkv * D);
    float *t1 = malloc(sizeof(float) * big), *t2 = malloc(sizeof(float) * big), *r = malloc(sizeof(float) * (size_t)L * H);
    float *mask = NULL;
    if (!h || !xn || !q || !qh || !kk || !vv || !kh || !vh || !t1 || !t2 || !r) return -1;

    if (embeds) {
        for (size_t i = 0; i < (size_t)L * H; ++i) h[i] = rnd(embeds[i], dt);
    } else {
        orc_embedding(ids, L, m->embed.w, m->embed.scales, m->embed.biases, m->quantized && m->embed.scales != NULL,
                      H, m->group_size, m->bits, dt, h); /* language.py:176 */
    }
    if (L > 1) { /* language.py:178-179 -> base.py:37-53 */
        mask = malloc(sizeof(float) * (size_t)L * T);
        if (!mask) return -1;
        orc_causal_mask(L, offset, dt, mask);
    }
    for (int li = 0; li < m->n_layers; ++li) {
        /* TransformerBlock.__call__, language.py:144-154 */
        orc_rms_norm(h, L, H, m->attn_norm[li], m->eps, dt, xn);
        /* Attention.__call__, language.py:75-108 */
        lin(m, &m->q[li], xn, L, nh * D, H, q);
        lin(m, &m->k[li], xn, L, nkv * D, H, kk);
        lin(m, &m->v[li], xn, L, nkv * D, H, vv);
        to_heads(q, L, nh, D, qh);
        to_heads(kk, L, nkv, D, kh);
        to_heads(vv, L, nkv, D, vh);
        orc_rope_ex(qh, nh, L, D, m->rope_freqs, offset, dt, m->rope_traditional, q);   /* q now [nh, L, D] */
        orc_rope_ex(kh, nkv, L, D, m->rope_freqs, offset, dt, m->rope_traditional, kk); /* kk now [nkv, L, D] */
        /* cache.update_and_fetch, reusable.py:134-142 */
        for (int g = 0; g < nkv; ++g)
            for (int l = 0; l < L; ++l) {
                memcpy(kcache[li] + ((size_t)g * cap + offset + l) * D, kk + ((size_t)g * L + l) * D, sizeof(float) * D);
                memcpy(vcache[li] + ((size_t)g * cap + offset + l) * D, vh + ((size_t)g * L + l) * D, sizeof(float) * D);
            }
        orc_sdpa(q, kcache[li], vcache[li], nh, nkv, L, T, cap, D, scale, mask, dt, sdpa_fused, qh);
        from_heads(qh, L, nh, D, t1); /* transpose(0,2,1,3).reshape(B,L,-1), language.py:107 */
        lin(m, &m->o[li], t1, L, H, nh * D, r);
        orc_add(h, r, (size_t)L * H, dt, h); /* h = x + r, language.py:151 */
        orc_rms_norm(h, L, H, m->mlp_norm[li], m->eps, dt, xn);
        /* MLP.__call__, language.py:126-127 */
        lin(m, &m->gate[li], xn, L, I, H, t1);
        lin(m, &m->up[li], xn, L, I, H, t2);
        orc_silu_mul(t1, t2, (size_t)L * I, dt, t1);
        lin(m, &m->down[li], t1, L, H, I, r);
        orc_add(h, r, (size_t)L * H, dt, h); /* out = h + r, language.py:153 */
    }
    if (hidden_out) memcpy(hidden_out, h, sizeof(float) * (size_t)L * H);
    orc_rms_norm(h, L, H, m->final_norm, m->eps, dt, xn); /* language.py:187 */
    const orc_linear_t *head = m->tie_word_embeddings ? &m->embed : &m->lm_head; /* language.py:206-209 */
    if (last_only)
        lin(m, head, xn + (size_t)(L - 1) * H, 1, m->vocab, H, logits);
    else
        lin(m, head, xn, L, m->vocab, H, logits);
    free(h); free(xn); free(q); free(qh); free(kk); free(vv); free(kh); free(vh); free(t1); free(t2); free(r);
    free(mask);
    return 0;
}
