/*
 * pie_hip.h -- C ABI of libpie_hip.so: the MI355X (gfx950) implementation of PIE's decode hot path.
 *
 * The reference (TheProxyCompany/proxy-inference-engine @ 2025-05-09) has no FFI for this path: its seam
 * is the set of MLX op call sites inside `proxy_inference_engine` (all tensor math is `import mlx.core as
 * mx`, engine/inference_engine.py:6) plus the one-symbol native module `pie_core` (hello(),
 * src/pie_core/src/bindings.cpp:6-9).  Each entry point below replaces exactly one of those call sites (or a
 * fused run of them) and cites it.  Paths are relative to /root/reference/src/proxy_inference_engine/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch/HIP C++ types in signatures (hipStream_t travels as void*).
 *   - All data pointers are DEVICE pointers owned by the caller.  The library never allocates or frees
 *     caller tensors and keeps no reference after return, except pie_decoder_* which records the weight /
 *     KV pointers it is given (the caller keeps those alive until pie_decoder_destroy).
 *   - Every launch is asynchronous on `stream`; nothing synchronises the device.
 *   - Return value: 0 = PIE_OK, negative = PIE_E_*.  Nothing throws across the ABI.
 *     pie_last_error() returns a thread-local description of the last failure on the calling thread.
 *   - `dtype`: activation / parameter float type, PIE_BF16 or PIE_F16 (16-bit storage, fp32 accumulate).
 *   - Threading: thread-compatible (no global mutable state besides the thread-local error string); one
 *     host thread drives one decoder, like the reference's single non re-entrant InferenceEngine
 *     (server/app.py:23,35).
 */
#ifndef PIE_HIP_H
#define PIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PIE_OK = 0, PIE_E_ARG = -1, PIE_E_SHAPE = -2, PIE_E_ALIGN = -3, PIE_E_HIP = -4, PIE_E_STATE = -5, PIE_E_ARCH = -6, PIE_E_RANGE = -7,
       PIE_EXHAUSTED = 1 /* pie_page_alloc: no free page (the reference's std::nullopt); not an error */ };
enum { PIE_BF16 = 1, PIE_F16 = 2, PIE_I8 = 3 /* KV page storage only: int8 rows + per-head fp16 scales (page pool, pie_paged_*_i8) */ };

/* pie_core.hello()  (src/pie_core/src/bindings.cpp:8; asserted by tests/python/test_basic.py:16). */
const char *pie_hello(void);
const char *pie_version(void);
const char *pie_last_error(void);
/* Fills name (<= name_len bytes, NUL-terminated gcnArchName), CU count, bytes of device memory. */
int pie_device_info(char *name, int name_len, int *n_cus, size_t *hbm_bytes);

/* Test / tuning switches (no reference counterpart).  One process-wide table; value PIE_KNOB_DEFAULT restores the built-in default.  Every knob is
 * exercised by a test (tests/test_gpu_decode.py, test_gpu_ops.py, test_gpu_paged.py compare the forms they select); none changes a result beyond
 * what its comment says.  Nothing in the library reads the environment. */
enum {
    PIE_KNOB_PREFILL_MIN = 0,        /* prompts shorter than this run as iterated decode steps (MLX's qmv regime); default 6, min 2 */
    PIE_KNOB_PREFILL_CHUNK = 1,      /* rows per prompt chunk; default 4096 (16..8192) */
    PIE_KNOB_PREFILL_RESIDENT = 2,   /* GiB budget for resident dequantised layer matrices; default: half of the free HBM; 0 = none */
    PIE_KNOB_SMALL_M = 3,            /* rows up to which int4 Linears that k_w4r_gemm does not take use the few-row kernel; default 32; 0 = the T copy + library GEMM */
    PIE_KNOB_W4L_SLABS = 4,          /* 0: K-split products of the int4 GEMMs reduced by their own launch instead of by their consumers (bit-equal) */
    PIE_KNOB_PREFILL_ATTN_VALU = 5,  /* 1: the VALU prompt attention instead of the MFMA flash kernel (the tests' cross-check) */
    PIE_KNOB_PREFILL_QT = 6,         /* 1 / 2: one / two 32-row query tiles per prompt-attention workgroup */
    PIE_KNOB_ATTN_MERGE_MAX_CAP = 7, /* cache capacity up to which o_proj merges the split-KV partials (read at pie_decoder_create); default 1024 */
    PIE_KNOB_ATTN_WARM_MAX_MB = 8,   /* the attention launch's idle CUs warm the Infinity Cache with at most this many MB of o_proj's weights; 0 = off (read per step enqueue / graph capture) */
    PIE_KNOB_W4R = 9,                /* 0: int4 Linears of 6..256 rows on the round-2 kernels (k_w4m_gemm, k_w4l2_gemm) instead of the weight-streaming k_w4r_gemm (the tests' cross-check) */
    PIE_KNOB_FUSE_ATTN = 10,         /* 0: the step's attention as its own launch instead of behind the q|k|v launch's XCD-local seam (32 / 8 / 128 heads; bit-identical; read per step enqueue / graph capture) */
    PIE_KNOB_COUNT = 11
};
#define PIE_KNOB_DEFAULT (-1)
int pie_set_knob(int knob, int value);
int pie_get_knob(int knob); /* the value set, or PIE_KNOB_DEFAULT */

/* ---------------------------------------------------------------- mx.quantize / mx.dequantize (K10)
 * Call sites: cache/kv_cache/cache.py:144-147, quantized.py:91-96; defines the checkpoint triplet
 * models/utils.py:96-111 consumes.  w [N,K] T -> codes uint32 [N,K/8], scales/biases T [N,K/64].
 * Bit-exact with the published algorithm: integer codes, T-rounded affine params. */
int pie_quantize_w4g64(const void *w, int N, int K, int dtype, uint32_t *codes, void *scales, void *biases, void *stream);
int pie_dequantize_w4g64(const uint32_t *codes, const void *scales, const void *biases, int N, int K, int dtype,
                         void *w_out, void *stream);

/* ---------------------------------------------------------------- weight streaming layout ("W4S")
 * Load-time repack of one MLX-quantised Linear (weight/scales/biases triplet, models/utils.py:96-111) into
 * the layout the GEMV streams: units of 2304 B = one ROW PAIR x one 2048-wide K slice
 * (2 x [64 lanes x 16 B] nibble-reordered codes + [64 lanes x 4 B] {scale,bias}); unit index = pair*n_slices + slice.
 * row_map (device int32 [N_out], may be NULL = identity) gives, per packed row, the source row of
 * `codes`; N_out must be even.  This is how q|k|v are concatenated with RoPE partners adjacent and
 * gate|up interleaved.  pie_w4s_bytes() is the size of `packed`. */
size_t pie_w4s_bytes(int N_out, int K);
int pie_repack_w4g64(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K,
                     const int32_t *row_map, int N_out, void *packed, void *stream);

/* ---------------------------------------------------------------- mx.quantized_matmul(x, w, scales, biases,
 * transpose=True, group_size=64, bits=4) via nn.QuantizedLinear.__call__ (K1).
 * Call sites: models/llama/language.py:83 (q,k,v), :108 (o), :127 (gate,up,down), :207/:209 (lm_head).
 * x [M,K] T, packed = W4S of a [N,K] weight, lin_bias T [N] or NULL (nn.QuantizedLinear bias), y [M,N] T.
 * Row-by-row exact-fp32 arithmetic (MLX's qmv regime) for every M; the weights are streamed once per up to 5 rows of x
 * (k_w4s_gemv_rows), each row bit-identical to that row multiplied alone. */
int pie_qgemv_w4g64(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y,
                    int dtype, void *stream);

/* ---------------------------------------------------------------- MLX int8 group-64 checkpoints (config "quantization":
 * {"group_size": 64, "bits": 8}): weight uint32 [N, K/4] (byte i of word w = code 4w+i), scales/biases T [N, K/64].
 * Same call sites and kernel as the int4 path on "W8S" units of 4352 B (row pair x 2048-wide K slice, four 16-byte code
 * pieces per lane).  pie_quantize_g64 (bits = 2 | 4 | 6 | 8) / pie_dequantize_g64 / pie_embedding_g64 (bits = 4 | 8) are the bits-generic forms
 * of the w4g64 entry points above. */
size_t pie_w8s_bytes(int N_out, int K);
int pie_repack_w8g64(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map,
                     int N_out, void *packed, void *stream);
int pie_qgemv_w8g64(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream);
int pie_quantize_g64(const void *w, int N, int K, int bits, int dtype, uint32_t *codes, void *scales, void *biases, void *stream);
int pie_dequantize_g64(const uint32_t *codes, const void *scales, const void *biases, int N, int K, int bits, int dtype, void *w_out,
                       void *stream);
int pie_embedding_g64(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H,
                      int bits, int dtype, void *out, void *stream);

/* ---------------------------------------------------------------- MLX int2 group-64 checkpoints (config "quantization": {"group_size": 64 | 128,
 * "bits": 2}; models/utils.py:96-111 forwards any bits nn.quantize takes): weight uint32 [N, K/16] (code k of a word at bits [2k, 2k+2)),
 * scales / biases T [N, K/64].  Same call sites and streaming kernel on "W2S" units of 1280 B (row pair x 2048-wide K slice, ONE 16-byte code
 * piece per lane + its {scale | bias << 16}): 0.3125 B per weight in HBM, the checkpoint's own figure (round 4 streamed these as 4-bit codes,
 * 0.5625 B).  The embedding TABLE of such a checkpoint is handed to the decoder as 4-bit codes (one row per step); W2S is a Linear format. */
size_t pie_w2s_bytes(int N_out, int K);
int pie_repack_w2g64(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream);
int pie_qgemv_w2g64(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream);

/* ---------------------------------------------------------------- MLX int6 group-64 checkpoints (config "quantization": {"group_size": 64 | 128,
 * "bits": 6}): weight uint32 [N, 3K/16] -- MLX's little-endian bit stream, code k of a row at bits [6k, 6k+6) -- scales / biases T [N, K/64].
 * "W6S" units of 3328 B (row pair x 2048-wide K slice): the low nibbles as the W4S unit's two code pieces, the high two bits as the W2S unit's one
 * piece, then {scale | bias << 16} per lane: 0.8125 B per weight in HBM, the checkpoint's own figure (round 4 streamed these as bytes, 1.0625 B).
 * A group's dot product = the W4S dot of the low plane + 16 x the W2S dot of the high plane.  The embedding table is handed over as 8-bit codes. */
size_t pie_w6s_bytes(int N_out, int K);
int pie_repack_w6g64(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream);
int pie_qgemv_w6g64(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream);

/* ---------------------------------------------------------------- MLX group-32 checkpoints (config "quantization": {"group_size": 32,
 * "bits": 4 | 8}; nn.quantize takes any group_size in {32, 64, 128}, models/utils.py:96-111): weight uint32 [N, K*bits/32], scales / biases
 * T [N, K/32].  Same call sites and streaming kernel as the group-64 paths on "W4S32" units of 2560 B / "W8S32" units of 4608 B: the W4S / W8S
 * unit with TWO {scale | bias << 16} words per lane (its code pieces are two consecutive 32-wide groups).  Decode steps and prompts below 6
 * rows multiply in the exact-fp32 regime (one pass over the weights per row); longer prompts dequantise to T and call the library GEMM. */
size_t pie_w4s32_bytes(int N_out, int K);
int pie_repack_w4g32(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream);
int pie_qgemv_w4g32(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream);
size_t pie_w8s32_bytes(int N_out, int K);
int pie_repack_w8g32(const uint32_t *codes, const void *scales, const void *biases, int N_src, int K, const int32_t *row_map, int N_out,
                     void *packed, void *stream);
int pie_qgemv_w8g32(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream);
int pie_embedding_g32(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H, int bits, int dtype, void *out,
                      void *stream);

/* ---------------------------------------------------------------- dense checkpoints (no "quantization" entry in
 * config.json, models/utils.py:96-97): nn.Linear / nn.Embedding with 16-bit weights.  Same streaming kernel as the int4
 * path on "W16S" units of 2048 B = one ROW PAIR x one 512-wide K slice (2 x [64 lanes x 16 B]); row_map as for
 * pie_repack_w4g64.  pie_gemv_dense: y[M,N] = x[M,K] @ W.T (+ bias), fp32 accumulate, one rounding
 * (call sites models/llama/language.py:83,108,127,209).  pie_embedding_dense: row gather (language.py:176). */
size_t pie_w16s_bytes(int N_out, int K);
int pie_repack_dense(const void *w, int N_src, int K, const int32_t *row_map, int N_out, void *packed, void *stream);
int pie_gemv_dense(const void *x, int M, const void *packed, int N, int K, const void *lin_bias, void *y, int dtype, void *stream);
int pie_embedding_dense(const int32_t *ids, int L, const void *table, int V, int H, int dtype, void *out, void *stream);

/* The same product WITHOUT the final rounding: y fp32 [M,N] = the fp32 row sums.  For the row-parallel Linears (o_proj,
 * down_proj) of a tensor-parallel shard, whose partial sums are all-reduced over the ranks before the one rounding to T
 * (proxy_inference_engine_amd/tp.py; the reference has no parallelism). */
int pie_qgemv_w4g64_f32(const void *x, int M, const void *packed, int N, int K, float *y, int dtype, void *stream);

/* nn.QuantizedEmbedding.__call__ (models/llama/language.py:176): dequantise gathered rows of the
 * MLX-layout table.  ids device int32 [L] -> out [L,H] T. */
int pie_embedding_w4g64(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases,
                        int V, int H, int dtype, void *out, void *stream);

/* ---------------------------------------------------------------- mx.fast.rms_norm(x, w, eps) (K3)
 * Call sites: nn.RMSNorm at models/llama/language.py:137-141, :168.  x,y [rows,H] T, w [H] T. */
int pie_rms_norm(const void *x, const void *w, float eps, int rows, int H, int dtype, void *y, void *stream);

/* ---------------------------------------------------------------- mx.fast.rope(x, D, traditional=False,
 * base=None, scale=1.0, offset, freqs) (K4).  Call site: models/llama/utils.py:42-50.
 * x,y [heads,L,D] T; freqs device fp32 [D/2]; position of row l = offset + l. */
int pie_rope(const void *x, int heads, int L, int D, const float *freqs, int offset, int dtype, void *y, void *stream);
/* the same with mx.fast.rope's `traditional` flag: non-zero rotates the interleaved pairs (2i, 2i+1) */
int pie_rope_ex(const void *x, int heads, int L, int D, const float *freqs, int offset, int traditional, int dtype, void *y,
                void *stream);

/* ---------------------------------------------------------------- mx.fast.scaled_dot_product_attention (K2)
 * Call site: models/base.py:111-113 <- models/llama/language.py:98-105.  Decode form (L = 1, mask = None):
 * q [Hq,1,D] T; k,v [Hkv,cap,D] T of which the first T positions are attended (the strided view
 * cache/kv_cache/reusable.py:142 returns); out [Hq,1,D] T.  GQA: q-head h uses kv-head h / (Hq/Hkv).
 * fp32 scores / softmax / PV, one rounding at the end (MLX fused-kernel contract).
 * workspace: device scratch of pie_sdpa_decode_workspace_bytes(Hq, D) bytes. */
size_t pie_sdpa_decode_workspace_bytes(int Hq, int D);
int pie_sdpa_decode(const void *q, const void *k, const void *v, int Hq, int Hkv, int T, int cap, int D, float scale,
                    int dtype, void *out, void *workspace, void *stream);

/* The same call site at L > 1 with the causal mask of models/base.py:37-53 (query row l at absolute position offset + l sees
 * keys 0 .. offset + l): q, out [L, Hq, D] T (the [B, L, h, D] order of language.py:86 before its transpose); k, v
 * [Hkv, cap, D] with rows [0, offset + L) valid.  MFMA flash kernel; P kept as hi + lo halves of T (~fp32 contract). */
int pie_sdpa_prefill(const void *q, const void *k, const void *v, int Hq, int Hkv, int L, int offset, int cap, int D,
                     float scale, int dtype, void *out, void *stream);

/* nn.silu(a) * b (models/llama/language.py:127) and the residual adds (:151,:153) (K6, K7). */
int pie_silu_mul(const void *a, const void *b, size_t n, int dtype, void *y, void *stream);
int pie_add(const void *a, const void *b, size_t n, int dtype, void *y, void *stream);

/* ---------------------------------------------------------------- logits tail (K9)
 * engine/inference_engine.py:268-271 + samplers/__init__.py:37-38: logprobs = f32(logits) - logsumexp,
 * token = first argmax.  logits [V] T, logprobs fp32 [V], token device int32 [1]. */
int pie_logprobs_argmax(const void *logits, int V, int dtype, float *logprobs, int32_t *token, void *stream);
/* Measurement aid (no reference counterpart; SURVEY.md 8d "fraction of a measured device-copy bandwidth"): a bare streaming read of `bytes`
 * shaped like the weight GEMV's stream (one 8-wave workgroup per CU, non-temporal 16-byte loads, nothing computed).  bench.py times it for
 * roofline.stream_peak. */
int pie_stream_read(const void *p, size_t bytes, void *stream);

/* The stochastic branches of make_sampler (samplers/__init__.py:39-46) over fp32 log-probabilities [rows, V], one kernel, no sort:
 * every branch scales by 1 / temp, filters, and draws argmax(x + Gumbel) like mx.random.categorical.
 *   mode PIE_SAMPLE_CATEGORICAL  categorical_sampling (categorical.py:6-8)
 *        PIE_SAMPLE_TOP_K        top_k_sampling(k)                        (top_k.py:14-29; k in (0, V), else PIE_E_ARG like its ValueError)
 *        PIE_SAMPLE_TOP_P        top_p_sampling(p)                        (top_p.py:18-33; kept: ascending cumulative probability > 1 - p)
 *        PIE_SAMPLE_MIN_P        min_p_sampling(p, min_tokens_to_keep=k)  (min_p.py:30-60)
 * seed + counter: the random stream (Philox-4x32-10; the library's own, not MLX's).  counter = DEVICE uint64[2], zeroed by the caller when
 * it (re)seeds; the last workgroup advances it, so a captured graph keeps drawing fresh numbers.  workspace: pie_sample_workspace_bytes(rows, V)
 * bytes of device memory, ZEROED once by the caller and then reused from call to call (the kernels leave it ready for the next one).
 * tokens int32 [rows]; kept_count (nullable) int32 [rows] = number of drawable ids; kept_mask (nullable) uint8 [rows, V] = which ids the
 * filter keeps (tests compare it with the reference's sort-based definition).  2 launches (categorical, min-p) or 5 (top-k, top-p), a row
 * spread over V / 512 workgroups; V <= 524288. */
enum { PIE_SAMPLE_CATEGORICAL = 0, PIE_SAMPLE_TOP_K = 1, PIE_SAMPLE_TOP_P = 2, PIE_SAMPLE_MIN_P = 3 };
size_t pie_sample_workspace_bytes(int rows, int V);
int pie_sample(const float *logprobs, int rows, int V, int mode, double temp, double p, int k, unsigned long long seed, unsigned long long *counter,
               void *workspace, int32_t *tokens, int32_t *kept_count, unsigned char *kept_mask, void *stream);

/* ---------------------------------------------------------------- fused decode step
 * One forward of Model.__call__ (models/llama/language.py:199-210) for inputs[1,1] over per-layer
 * ReusableKVCache buffers (cache/kv_cache/reusable.py:96-142) followed by the tail of _inference
 * (engine/inference_engine.py:252-271) with the greedy sampler, as 5 launches per layer (6 beyond 1024 positions):
 *   rmsnorm+qkv GEMV+RoPE+cache append | split-KV attention | [combine] | split merge+o_proj+residual |
 *   rmsnorm+gate/up GEMV+SwiGLU | down_proj+residual ; then rmsnorm+lm_head ; log-softmax+argmax.
 * 4 per layer for models with 32 query / 8 kv heads of 128 and hidden <= 4096 (Llama-3-8B, Mistral-7B; any weight format) on a contiguous or T-page cache up to 1024 positions: the
 * attention then runs inside the q|k|v launch, behind an XCD-local seam (PIE_KNOB_FUSE_ATTN; bit-identical).
 * Position and token live in device memory so the captured hipGraph is replayable. */
typedef struct pie_decoder pie_decoder;

typedef struct {
    int dtype;          /* PIE_BF16 / PIE_F16 */
    int hidden, n_layers, n_heads, n_kv_heads, head_dim, inter, vocab;
    float rms_eps;
    int tie_word_embeddings; /* lm_head = embed_tokens.as_linear (language.py:206-207) */
    int kv_splits;      /* split-KV factor of the attention kernel (0 = default) */
    int weight_format;  /* PIE_W_INT4_G64 (MLX int4 group-64 triplets, W4S units) or PIE_W_DENSE (16-bit nn.Linear
                           weights, W16S units; embed_codes is then the T [vocab, hidden] table, scales/biases NULL) */
    int rope_traditional; /* ModelArgs.rope_traditional (language.py:27,69): rotate the interleaved pairs (2i, 2i+1); wqkv is
                             then the plain q|k|v concatenation (row_map NULL), not pie_qkv_row_map's order */
    int tp_rank, tp_world; /* tensor parallelism (0, 0 or 0, 1 = none).  With tp_world > 1 this decoder is ONE RANK's shard:
                             n_heads, n_kv_heads, inter and vocab are the LOCAL sizes (heads / world, intermediate / world, vocabulary
                             rows / world: proxy_inference_engine_amd/tp.py shard_config), hidden is the full width, the embedding
                             table holds all vocab * tp_world rows.  o_proj / down_proj produce fp32 partial sums that
                             pie_decoder_set_comm's communicator adds over the ranks before the one rounding + residual add. */
} pie_decoder_config;
enum { PIE_W_INT4_G64 = 0, PIE_W_DENSE = 1, PIE_W_INT8_G64 = 2 /* W8S units, embed_codes uint32 [vocab, hidden/4] */,
       PIE_W_INT4_G32 = 3 /* W4S32 units, embed scales / biases [vocab, hidden/32] */, PIE_W_INT8_G32 = 4 /* W8S32 units */,
       PIE_W_INT2_G64 = 5 /* W2S units (Linear matrices only: per-matrix fmt_* or the default with fmt_embed = PIE_W_INT4_G64 + 1) */,
       PIE_W_INT6_G64 = 6 /* W6S units (Linear matrices only; the embedding table as 8-bit codes, fmt_embed = PIE_W_INT8_G64 + 1) */ };

typedef struct {
    const void *attn_norm, *mlp_norm;   /* T [hidden] */
    const void *wqkv;                    /* W4S of [q;k;v] rows in RoPE-paired order (pie_qkv_row_map) */
    const void *wo;                      /* W4S [hidden, n_heads*head_dim] */
    const void *wgateup;                 /* W4S of interleaved (gate_i, up_i) rows */
    const void *wdown;                   /* W4S [hidden, inter] */
    /* optional Linear biases (attention_bias / mlp_bias, models/llama/language.py:42-53,117-126), T, NULL = none;
       bqkv in the packed q|k|v row order (pie_qkv_row_map), bgateup interleaved (pie_gateup_row_map) */
    const void *bqkv, *bo, *bgateup, *bdown;
    /* Per-matrix weight format: 0 = the decoder's weight_format, else PIE_W_* + 1.  The reference decides quantisation PER MODULE
       (models/utils.py:99-109: quantised iff the checkpoint holds "{path}.scales" and the input width is a multiple of 64), so a
       checkpoint may mix int4 and 16-bit Linears; the fused groups (q|k|v, gate|up) must be uniform inside. */
    int fmt_qkv, fmt_o, fmt_gateup, fmt_down;
} pie_layer_weights;

typedef struct {
    const uint32_t *embed_codes;         /* MLX layout [vocab, hidden/8] */
    const void *embed_scales, *embed_biases; /* T [vocab, hidden/64] */
    const void *final_norm;              /* T [hidden] */
    const void *lm_head;                 /* W4S [vocab, hidden] (of embed_tokens when tied) */
    const float *rope_freqs;             /* fp32 [head_dim/2], Llama3RoPE._freqs (models/llama/utils.py:39) */
    int fmt_embed, fmt_lm_head;          /* 0 = the decoder's weight_format, else PIE_W_* + 1 (see pie_layer_weights); a dense embedding:
                                            embed_codes is the T [vocab, hidden] table, scales / biases NULL */
} pie_global_weights;

/* Host helpers producing the row maps the decoder's fused epilogues assume (host int32 arrays). */
int pie_qkv_row_map(int n_heads, int n_kv_heads, int head_dim, int32_t *map /* [(n_heads+2*n_kv_heads)*head_dim] */);
int pie_gateup_row_map(int inter, int32_t *map /* [2*inter], source rows of cat(gate, up) */);

int pie_decoder_create(const pie_decoder_config *cfg, pie_decoder **out);
int pie_decoder_destroy(pie_decoder *d);
int pie_decoder_set_layer(pie_decoder *d, int layer, const pie_layer_weights *w);
int pie_decoder_set_globals(pie_decoder *d, const pie_global_weights *w);
/* Per-layer cache buffers [n_kv_heads, capacity, head_dim] T (ReusableKVCache.keys/values with B=1);
 * call again whenever a cache re-allocates (reusable.py:167-203).  k_ptrs/v_ptrs: HOST arrays of device pointers. */
int pie_decoder_set_kv(pie_decoder *d, const void *const *k_ptrs, const void *const *v_ptrs, int capacity, void *stream);
/* Paged alternative to pie_decoder_set_kv (the role of the reference's stub PagedKVCache,
 * cache/kv_cache/paged.py:1-14, over its PageAllocator): slabs = HOST array [n_layers] of device slab pointers (one
 * pie_page_pool geometry each, the same page ids valid in all of them), block_table = DEVICE int32 [max_blocks], caller-owned
 * and editable between steps (logical block j of the sequence -> page id).  Position p is stored in / read from page
 * block_table[p / 64], row p % 64; the caller must have filled the table for every position a step or prefill touches.
 * Capacity = max_blocks * 64.  Steps, prefill, graphs and outputs behave exactly as with contiguous buffers. */
int pie_decoder_set_paged_kv(pie_decoder *d, const void *const *slabs, size_t n_pages, const int32_t *block_table, int max_blocks,
                             void *stream);
/* One decode step for B sequences at once over the paged KV pool (continuous batching): the weights stream once for all rows;
 * row s is its own sequence -- input token tokens[s], attending context_lens[s] positions INCLUDING the new one (its position is
 * context_lens[s] - 1; 0 = idle slot), K / V appended to page block_tables[s][pos / 64] of every layer's slab (slabs: HOST array
 * [n_layers] of device pointers), attention over its own table row.  Outputs per row: logits T [B, vocab], logprobs fp32
 * [B, vocab], next_tokens int32 [B] (greedy).  What the reference's decode-state batch is meant to be (BatchDetails,
 * include/engine/batch_details.hpp:10-88; Scheduler and the paged attention kernel are skeletons there).  Independent of
 * pie_decoder_set_kv / set_state / bind_outputs; all device arrays are caller-owned.  flags: PIE_STEP_GRAPH replays a captured
 * hipGraph of the step while the caller passes the same buffers (their contents may change) -- captured on the second such call. */
int pie_decoder_step_batch(pie_decoder *d, const int32_t *tokens, const int32_t *context_lens, const void *const *slabs, size_t n_pages, size_t slab_bytes,
                           const int32_t *block_tables, int max_blocks, int B, void *logits, float *logprobs, int32_t *next_tokens,
                           int flags, void *stream);
/* Several FRESH prompts in one pass (the prefill-state half of BatchDetails, batch_details.hpp:10-88): their N rows are
 * concatenated -- ids [N]; per row: row_context_lens (position + 1), row_seq (which prompt, = its block-table row), seg_lo
 * (index of its prompt's first row), seg_hi (row index + 1: causal); per prompt: last_rows [S] (index of its last row).  K / V
 * go to each prompt's pages; attention reads this pass's own rows.  Outputs per prompt as in pie_decoder_step_batch.
 * All index arrays are device int32.  slab_bytes (here and in pie_decoder_step_batch): the size of EACH layer's slab; it must hold
 * n_pages pages of the ACTIVE page format (T pages, or int8 pages under PIE_OPT_KV_I8) -- a pool of the other format is refused
 * instead of being indexed with the wrong page stride. */
int pie_decoder_prefill_batch(pie_decoder *d, const int32_t *ids, const int32_t *row_context_lens, const int32_t *row_seq, const int32_t *seg_lo,
                              const int32_t *seg_hi, const int32_t *last_rows, int N, int S, const void *const *slabs, size_t n_pages, size_t slab_bytes,
                              const int32_t *block_tables, int max_blocks, void *logits, float *logprobs, int32_t *next_tokens, void *stream);
/* A MIXED batch: decode-state sequences and prompts in ONE pass over the weights (batch_details.hpp:10-88 holds both kinds in one
 * BatchDetails; the scheduler body that would form it is empty upstream).  The layout of pie_decoder_prefill_batch with n_decode decode
 * rows in front: row r < n_decode is sequence r (row_seq[r] == r, block-table row r) with its input token ids[r], row_context_lens[r] =
 * the positions it attends INCLUDING the new one, seg_lo[r] = r, seg_hi[r] = r + 1; the prompts' rows follow with row_seq >= n_decode.
 * out_rows [S]: the rows whose logits are wanted -- the n_decode decode rows first, then every prompt's last row.  The decode rows'
 * attention runs over their pages (pie_decoder_step_batch's kernel), a fresh prompt's rows' over this pass's own rows.
 * chunks_host (HOST int32 [n_chunks][4] = {first row, rows, cached positions, sequence}; n_chunks may be 0): prompts that CONTINUE a cached
 * prefix -- a chunk of a long prompt, or a suffix behind shared prefix pages (KVPage ref counts, page.hpp:55-68).  Their rows carry
 * row_context_lens = cached + i + 1 and trivial segments (seg_lo = r, seg_hi = r + 1); their attention is the single-prompt causal flash
 * kernel over the sequence's pages from that offset.  T pages only.  n_decode == 0 and n_chunks == 0 is pie_decoder_prefill_batch. */
int pie_decoder_step_mixed(pie_decoder *d, const int32_t *ids, const int32_t *row_context_lens, const int32_t *row_seq, const int32_t *seg_lo,
                           const int32_t *seg_hi, const int32_t *out_rows, int N, int S, int n_decode, const void *const *slabs, size_t n_pages,
                           size_t slab_bytes, const int32_t *block_tables, int max_blocks, void *logits, float *logprobs, int32_t *next_tokens,
                           int n_chunks, const int32_t *chunks_host, void *stream);
/* offset = cache.offset before the step (reusable.py:111); token < 0 keeps the device-side token (the
 * previous step's argmax). */
int pie_decoder_set_state(pie_decoder *d, int offset, int token, void *stream);
/* Runs one step.  flags: PIE_STEP_LOGITS computes lm_head + logprobs + argmax (else the step only fills
 * the KV caches: prompt tokens before the last); PIE_STEP_GRAPH replays a captured hipGraph of the step
 * (captured on first use per flag set).  After the step the device-side offset is incremented and, with
 * PIE_STEP_LOGITS, the device-side token is the greedy choice. */
enum { PIE_STEP_LOGITS = 1, PIE_STEP_GRAPH = 2 };
int pie_decoder_step(pie_decoder *d, int flags, void *stream);
/* Kernel nodes of the captured step graph (read back with hipGraphGetNodes when it was instantiated); -1 while no step with these flags has
 * been captured.  bench.py reports it as config.launches_per_step. */
int pie_decoder_graph_launches(const pie_decoder *d, int flags);
/* Prompt processing: Model.__call__(inputs[1, L]) from the current device-side offset, ids[0..L) device int32, all
 * launches queued back to back with no host round trip.  L >= 6 (pie_set_knob(PIE_KNOB_PREFILL_MIN, n)): batched, in chunks of
 * 4096 rows (PIE_KNOB_PREFILL_CHUNK) -- MLX's qmm regime: nn.QuantizedLinear at L > 1 dequantises to T before a T x T -> fp32 MMA.
 * int4 group-64 matrices run hand-written MFMA GEMMs straight from their 4-bit tiles (k_w4r_gemm up to 256 rows, k_w4l2_gemm beyond);
 * dense, int8 and group-32 matrices multiply a 16-bit copy in MFMA-ordered tiles (W16M; k_w16l_gemm -- no library GEMM since round 5;
 * in_features % 64 == 0), with HIP kernels for RMSNorm, RoPE + cache append, causal attention, SwiGLU and residuals.  Shorter prompts: iterated decode steps (the qmv regime).  logits_all == NULL: lm_head + tail only for the last token (the engine only
 * reads logits[:, -1, :], engine/inference_engine.py:254).  logits_all != NULL: T [L, vocab], lm_head on every
 * position like the reference's Model.__call__ (language.py:205-209). */
int pie_decoder_prefill(pie_decoder *d, const int32_t *ids, int L, void *logits_all, void *stream);
/* The same with the input embeddings given instead of token ids: embeds T [L, hidden] replaces embed_tokens(ids) --
 * `h = inputs_embeds` of the VLM text tower (models/intern/language.py:155-158), fed by the ensemble's merged text and image
 * features (models/intern/ensemble.py:33-91, :106-108).  Always the batched path, whatever L -- except on int8 pages (PIE_OPT_KV_I8 with a
 * paged cache), where the rows run as decode steps like a prompt of tokens does there. */
int pie_decoder_prefill_embeds(pie_decoder *d, const void *embeds, int L, void *logits_all, void *stream);
/* Output buffers of the step, allocated by the caller (device): logits T [vocab], logprobs fp32 [vocab],
 * token int32 [1] (the greedy choice), hidden T [hidden] (the residual stream; after a step it holds the
 * output of the last block, the input of the final norm).  history (optional, int32 [history_len]): the tail
 * kernel stores the greedy token chosen for position p at history[p], so the host can read the generated ids
 * later in one copy instead of synchronising every step (PromptCache.computed_ids, cache/prompt_cache.py:43-50).
 * Required before the first step. */
int pie_decoder_bind_outputs(pie_decoder *d, void *logits, float *logprobs, int32_t *token, void *hidden, int32_t *history,
                             int history_len);
/* Copies a device-resident token id into the decoder's device-side state (the input of the next
 * pie_decoder_step) without a host round trip; a no-op when `token_dev` is the bound token output. */
int pie_decoder_set_token_from(pie_decoder *d, const int32_t *token_dev, void *stream);
/* The step's launches by name.  pie_decoder_launch_kernel() enqueues ONE of them with exactly the arguments
 * the step uses (for per-kernel timing with events / rocprof; it does not advance the decode state, and
 * PIE_K_TAIL, which does, is refused).  pie_decoder_kernel_bytes() is that launch's algorithmic HBM traffic
 * (weights 0.5625 B/parameter + norm weights + KV rows; activations of a few KB are excluded).
 * Inside pie_decoder_step an int4 checkpoint has no PIE_K_EMBED launch: layer 0's PIE_K_QKV launch dequantises the token's row itself;
 * launched by name they remain two kernels with the same results (tests/test_gpu_decode.py::test_step_equals_its_kernels_launched_by_name). */
enum { PIE_K_EMBED = 0, PIE_K_QKV = 1, PIE_K_ATTN = 2, PIE_K_OPROJ = 3, PIE_K_GATEUP = 4, PIE_K_DOWN = 5, PIE_K_LMHEAD = 6, PIE_K_TAIL = 7 };
int pie_decoder_launch_kernel(pie_decoder *d, int which, int layer, void *stream);
size_t pie_decoder_kernel_bytes(const pie_decoder *d, int which, int T);
/* Algorithmic HBM bytes one decode step moves at context length T (SURVEY.md 8d formula). */
size_t pie_decoder_step_bytes(const pie_decoder *d, int T, int with_logits);
/* Decoder options.  PIE_OPT_KV_I8 = 1: the slabs handed to pie_decoder_step_batch / pie_decoder_prefill_batch hold int8 pages (PIE_I8 pools,
 * pie_paged_*_i8 below): new rows are quantised with their page's scales, the step's attention reads them back.  Changing an option drops the
 * captured graphs.  pie_decoder_status: synchronises the device and reports a give-up of a bounded wait of the tensor-parallel collectives in
 * *error (0 = none; sticky; that step's token is -1); bit 31: the fused q|k|v + attention launch (32 / 8 / 128 head geometry, PIE_KNOB_FUSE_ATTN) gave up
 * waiting for its kv-group (the producers of a group must get dispatched while its <= 128 attention workgroups spin: guaranteed for up to five such
 * launches in flight on a full device; the library fuses only while a process holds at most four decoders; a process that masks CUs sets
 * PIE_KNOB_FUSE_ATTN = 0).  Results after a give-up are not valid. */
enum { PIE_OPT_KV_I8 = 2 };
int pie_decoder_configure(pie_decoder *d, int option, int value);
int pie_decoder_status(pie_decoder *d, unsigned *error);

/* ---------------------------------------------------------------- tensor-parallel communicator (SURVEY.md 8 row e)
 * The reference has no multi-GPU path; BASELINE.json configs[4] (Llama-3-70B over the 8 GPUs of one node) needs
 * one.  A decode step has two all-reduces of ONE hidden vector per layer and a 3-number exchange in the tail: latency-bound, so
 * this is a one-shot push over IPC-mapped peer memory, not a ring: every rank stores its fp32 vector as 8-byte {value, epoch}
 * granules into its slot of every peer's receive area (one xGMI hop, the data is its own flag), then adds the slots of its own
 * area in rank order -- bit-identical sums on all ranks.  One process per GPU:
 *   pie_comm_create(rank, world, max_elems, &c)   allocates this rank's receive area (fine-grained device memory)
 *   pie_comm_export(c, handle64)                  64-byte hipIpcMemHandle_t of that area; the host gathers all ranks' handles
 *                                                 (torch.distributed all_gather, MPI, a file: any side channel)
 *   pie_comm_connect(c, handles)                  handles: world * 64 bytes in rank order (own entry ignored); maps the peers
 *   pie_allreduce_f32(c, data, n, stream)         in-place sum over the ranks, stream-ordered, no host synchronisation
 *   pie_decoder_set_comm(d, c)                    the decoder's steps (eager or captured in a hipGraph) use c; every rank must
 *                                                 enqueue the same sequence of steps.  c must outlive d.
 *   pie_comm_status(c, &err)                      synchronises; err != 0: a bounded wait (2 s) for a peer's data gave up
 *   pie_comm_destroy(c)
 * world == 1 is a valid (self-connected) communicator.  A peer may run at most one collective ahead of the slowest rank.
 * Inside the decoder the push half of each all-reduce rides in the row-parallel GEMV's epilogue; the launch behind it pulls, rounds and adds
 * the residual.
 * A second backend runs the same collectives through RCCL (north_star: "RCCL all-reduce over xGMI") -- ncclAllReduce / ncclAllGather on the
 * launch stream, capturable in the step's hipGraph; the comparator and fall-back of the one-shot form on a real node:
 *   pie_comm_rccl_unique_id(id128)                         on one rank; the host hands the 128 bytes to every rank
 *   pie_comm_create_rccl(rank, world, max_elems, id128, &c) collective over the world (ncclCommInitRank); the communicator is connected
 * Its sums are RCCL's (identical on every rank, summed in the ring's order, not in rank order). */
typedef struct pie_comm pie_comm;
enum { PIE_COMM_IPC = 0, PIE_COMM_RCCL = 1 };
int pie_comm_create(int rank, int world, size_t max_elems, pie_comm **out);
int pie_comm_rccl_unique_id(void *id128);
int pie_comm_create_rccl(int rank, int world, size_t max_elems, const void *id128, pie_comm **out);
int pie_comm_export(const pie_comm *c, void *handle64);
int pie_comm_connect(pie_comm *c, const void *handles);
int pie_allreduce_f32(pie_comm *c, float *data, size_t n, void *stream);
int pie_comm_status(pie_comm *c, unsigned *error);
int pie_comm_destroy(pie_comm *c);
int pie_decoder_set_comm(pie_decoder *d, pie_comm *c);

/* ---------------------------------------------------------------- KV page pool
 * Replaces pie_core's PageAllocator / KVPage: src/pie_core/include/engine/page_allocator.hpp:17-72,
 * include/engine/page.hpp:14-123, src/engine/page_allocator.cpp:8-157 (contract pinned by
 * tests/cpp/test_page_allocator.cpp, restated in tests/test_page_pool.py).  A page holds PIE_PAGE_TOKENS token slots
 * (page.hpp:14): a K block then a V block, each T [n_kv_heads, 64, head_dim] (the reference's logical shape is
 * [64, heads, head_dim], page.hpp:29-30; head-major storage keeps one head's rows of a page contiguous).  All pages live in ONE caller-owned
 * device slab of pie_page_pool_slab_bytes() bytes (page p at byte offset p * slab_bytes / num_pages); `slab` may be
 * NULL for bookkeeping only.  The free list is lock-free (tagged index stack) and LIFO, seeded so a fresh pool hands
 * out 0, 1, 2, ... (page_allocator.cpp:52-63).  Every entry point is thread-safe.
 *   create: PIE_E_ARG for num_pages == 0, n_kv_heads <= 0 or head_dim <= 0 (std::invalid_argument, page_allocator.cpp:21-29)
 *   alloc:  PIE_OK and *page_id with ref count 1 and num_tokens 0 (page_allocator.cpp:68-79), or PIE_EXHAUSTED
 *   free:   drops one reference; the page returns to the pool when the count reaches 0 (page_allocator.cpp:81-87)
 *   any id >= size: PIE_E_RANGE (std::out_of_range of check_page_id, page_allocator.cpp:110-117)
 * dtype PIE_BF16 / PIE_F16: the pages hold T rows, the layout the decoder's attention kernels consume.
 * dtype PIE_I8: the reference's own storage (page.hpp:25-32,109-117): int8 K block, int8 V block (each [n_kv_heads, 64, head_dim]), then
 * the fp16 per-head scales of K and of V ([n_kv_heads, 1] each: key_cache_scale_ / value_cache_scale_), padded to 256 bytes
 * (pie_page_i8_bytes); pie_page_scale_ptrs hands out the two scale vectors of a page.  See pie_paged_kv_append_i8 below. */
enum { PIE_PAGE_TOKENS = 64 };
typedef struct pie_page_pool pie_page_pool;
size_t pie_page_pool_slab_bytes(size_t num_pages, int n_kv_heads, int head_dim, int dtype);
int pie_page_pool_create(size_t num_pages, int n_kv_heads, int head_dim, int dtype, void *slab, pie_page_pool **out);
int pie_page_pool_destroy(pie_page_pool *pool);
size_t pie_page_pool_size(const pie_page_pool *pool);
size_t pie_page_pool_num_free(const pie_page_pool *pool);
int pie_page_alloc(pie_page_pool *pool, uint32_t *page_id);
int pie_page_free(pie_page_pool *pool, uint32_t page_id);
int pie_page_add_ref(pie_page_pool *pool, uint32_t page_id);
int pie_page_ref_count(const pie_page_pool *pool, uint32_t page_id, uint32_t *count);
int pie_page_num_tokens(const pie_page_pool *pool, uint32_t page_id, size_t *n);
int pie_page_set_num_tokens(pie_page_pool *pool, uint32_t page_id, size_t n);
int pie_page_ptrs(const pie_page_pool *pool, uint32_t page_id, void **k, void **v);
size_t pie_page_i8_bytes(int n_kv_heads, int head_dim);
int pie_page_scale_ptrs(const pie_page_pool *pool, uint32_t page_id, void **k_scale, void **v_scale);

/* Paged decode attention over a batch of sequences: what the reference's placeholder
 * Attention::invoke_paged_attention_kernel (src/pie_core/src/layers/attention.cpp:71-83) and its dummy Metal kernel
 * (src/kernels/paged_attention.metal:6-23) stand for, with the inputs BatchDetails names
 * (include/engine/batch_details.hpp:42-66): block_table int32 [B, max_blocks] = consolidated_block_table (logical block
 * j of sequence s -> page id), context_lens int32 [B] = positions each query attends INCLUDING its own (0 = idle slot,
 * output row zeroed).  q, out: T [B, n_heads, head_dim]; slab: one layer's page slab (page p at p * page bytes; inside a
 * page K then V, each T [n_kv_heads, 64, head_dim] -- head-major so one head's rows of a page are one 16 KB burst).
 * Same arithmetic contract as pie_sdpa_decode: results equal attention over the gathered contiguous rows.
 * Both tables live in device memory, so a captured graph can replay the launch while the host edits them. */
size_t pie_paged_attn_workspace_bytes(int B, int n_heads, int head_dim);
int pie_paged_attn_decode(const void *q, const void *slab, size_t n_pages, const int32_t *block_table, int max_blocks,
                          const int32_t *context_lens, int B, int n_heads, int n_kv_heads, int head_dim, float scale, int dtype,
                          void *out, void *workspace, void *stream);
/* Writes the new K / V rows (T [B, n_kv_heads, head_dim], after RoPE) of B sequences into their pages: sequence s at
 * position positions[s] (< 0 = idle slot) -> page block_table[s][positions[s] / 64], row positions[s] % 64. */
int pie_paged_kv_append(const void *k, const void *v, void *slab, size_t n_pages, const int32_t *block_table, int max_blocks,
                        const int32_t *positions, int B, int n_kv_heads, int head_dim, int dtype, void *stream);

/* int8 pages (PIE_I8 pools).  The reference declares the storage (page.hpp:25-32; "head-wise quant for now", :114) and neither a
 * quantiser nor a reader, so the arithmetic is this library's (restated in oracle/pie_oracle.py):
 *   store  q = clamp(rint(x / s), -127, 127)   x = the T element as fp32, s = the page's fp16 scale of that kv-head
 *   read   x' = fp32(q) * fp32(s), used in fp32 by the attention (softmax and P.V in fp32, one rounding of the output to T)
 * pie_page_i8_set_scales: writes the K / V scale vectors (fp16 [n_kv_heads], DEVICE; null = ones, the reference constructor's
 *   value) into the listed pages (page_ids: DEVICE int32 [n]; null = pages 0 .. n - 1).  A page's scales must not change under
 *   rows already stored in it.
 * pie_paged_kv_append_i8 / pie_paged_attn_decode_i8: as pie_paged_kv_append / pie_paged_attn_decode (k, v, q, out are T;
 *   `dtype` is T), on a slab of int8 pages.  Workspace: pie_paged_attn_workspace_bytes. */
int pie_page_i8_set_scales(void *slab, size_t n_pages, int n_kv_heads, int head_dim, const int32_t *page_ids, int n, const void *k_scales,
                           const void *v_scales, void *stream);
int pie_paged_kv_append_i8(const void *k, const void *v, void *slab, size_t n_pages, const int32_t *block_table, int max_blocks,
                           const int32_t *positions, int B, int n_kv_heads, int head_dim, int dtype, void *stream);
int pie_paged_attn_decode_i8(const void *q, const void *slab, size_t n_pages, const int32_t *block_table, int max_blocks,
                             const int32_t *context_lens, int B, int n_heads, int n_kv_heads, int head_dim, float scale, int dtype,
                             void *out, void *workspace, void *stream);

/* ---------------------------------------------------------------- vision tower ops
 * Call sites: models/intern/vision.py (Qwen2.5-VL vision tower: PatchEmbed :87-121, Attention :143-186, MLP :189-197,
 * PatchMerger :124-140).  RMSNorm, SiLU * up and the residual adds are pie_rms_norm / pie_silu_mul / pie_add.
 * pie_linear_w16m: nn.Linear on M rows -- y [M, N] = x [M, K] . w [N, K]^T (+ bias [N]), T in, fp32 accumulate, T out (rounded
 *   before the bias is added, like the text tower's Linear) on the hand-written 16-bit MFMA GEMM (csrc/w16_gemm.hpp; rounds 1-4
 *   called hipBLASLt here).  The weight is given as W16M tiles -- 32 output rows x 64 columns in MFMA operand order, zero-padded --
 *   built once per matrix: pie_w16m_bytes(N, K) bytes (256-byte aligned), filled by pie_repack_w16m from the row-major [N, K] weight.
 *   x: M rows of ldx elements (0 = packed) holding 64 * ceil(K / 64) used columns, ZEROS past K (the tower's two odd widths, 1176
 *   and 3420, are padded by the host); y: rows of ldy elements (0 = packed).  swiglu != 0: the weight's rows interleave (gate_i, up_i),
 *   bias likewise, and y [M, N / 2] = T(T(silu(g)) * u) with g, u the rounded, biased Linear outputs (MLP, vision.py:196-197: the values
 *   of the GEMM followed by pie_bias_silu_mul).  workspace: pie_linear_w16m_workspace(M, N, K) bytes of device scratch for the
 *   shapes that split K (few rows, narrow matrices; 0 for most and for swiglu).  Any N (16-byte stores when N and ldy are multiples
 *   of 8); swiglu needs N % 8 == 0.
 * pie_gelu: nn.GELU() exact form, fp32 inside, one rounding.
 * pie_vision_qkv_rope: qkv T [N, 3, H, D] (+ optional bias T [3 * H * D], the qkv Linear's) -> q T [N, H, DP], k, v T [H, N, DP] with rotate-half rotary on q and k
 *   (apply_rotary_pos_emb_vision, vision.py:55-70; cos / sin fp32 [N, D/2] = the row's angles) and head dims D..DP-1 zeroed
 *   (DP = 64 or 128, the attention kernel's head sizes).
 * pie_sdpa_segments: block-diagonal non-causal attention, the mask of vision.py:160-167: query row r attends keys
 *   [seg_lo[r], seg_hi[r]) (device int32 [N], non-decreasing, lo <= r < hi).  q, out T [N, H, D]; k, v T [H, N, D]. */
size_t pie_w16m_bytes(int N, int K);
int pie_repack_w16m(const void *w, int N, int K, int dtype, void *w16m, void *stream);
size_t pie_linear_w16m_workspace(int M, int N, int K);
int pie_linear_w16m(const void *x, int ldx, const void *w16m, const void *bias, int M, int N, int K, int dtype, void *y, int ldy, int swiglu,
                    void *workspace, size_t workspace_bytes, void *stream);
int pie_gelu(const void *x, size_t n, int dtype, void *y, void *stream);
int pie_vision_qkv_rope(const void *qkv, const void *bias, const float *cos_t, const float *sin_t, int N, int H, int D, int DP, int dtype,
                        void *q, void *k, void *v, void *stream);
/* Bias adds folded into the op that consumes the GEMM output (same values as pie_linear_w16m with a bias followed by the op):
 * pie_bias_silu_mul: y = silu(gate + bias_gate) * (up + bias_up), T [M, N] (MLP, vision.py:196-197); gate / up rows are `ld`
 *   elements apart (0 = N), so both may be column halves of one [M, 2N] GEMM output;
 * pie_add_bias: y = x + (r + bias), T [M, N] (the residual adds of vision.py:212-218). */
int pie_bias_silu_mul(const void *gate, const void *up, const void *bias_gate, const void *bias_up, int M, int N, int ld, int dtype, void *y,
                      void *stream);
/* pie_add_bias_rms_norm: the residual add and the RMSNorm after it in one pass: y = x + (r + bias), xn = rms_norm(y, norm_w, eps);
 * T [M, N], N % 8 == 0 (the values pie_add_bias followed by pie_rms_norm produce). */
int pie_add_bias_rms_norm(const void *x, const void *r, const void *bias, const void *norm_w, float eps, int M, int N, int dtype, void *y, void *xn,
                          void *stream);
int pie_add_bias(const void *x, const void *r, const void *bias, int M, int N, int dtype, void *y, void *stream);
int pie_sdpa_segments(const void *q, const void *k, const void *v, const int32_t *seg_lo, const int32_t *seg_hi, int N, int H, int D,
                      float scale, int dtype, void *out, void *stream);

/* ---------------------------------------------------------------- few-row int4 GEMM (prompts of 2..32 tokens)
 * mx.quantized_matmul in its many-row regime (weights dequantised to T, T x T products, fp32 accumulation;
 * models/llama/language.py:83,108,127) without a 16-bit copy of the weights: W4M tiles (32 output rows x 64 columns, 1152 B,
 * built on the device from the W4S stream of pie_repack_w4g64) are dequantised in registers into MFMA operand fragments.
 * N % 32 == 0, K % 64 == 0, 1 <= M <= 32.  x T [M, K], y T [M, N] (rows in the packed order of the W4S matrix). */
size_t pie_w4m_bytes(int N, int K);
int pie_repack_w4s_to_w4m(const void *w4s, int N, int K, void *w4m, void *stream);
int pie_qgemm_w4m(const void *x, const void *w4m, int M, int N, int K, int dtype, void *y, void *stream);


#ifdef __cplusplus
}
#endif
#endif /* PIE_HIP_H */
