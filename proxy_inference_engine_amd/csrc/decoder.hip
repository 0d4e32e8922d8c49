// decoder.hip -- native decode-step runtime: the fused launch sequence of one Model.__call__
// (models/llama/language.py:199-210) for inputs[1,1] + the tail of _inference
// (engine/inference_engine.py:252-271), replayable as a hipGraph because position, token and the KV
// buffer addresses are read from device memory.
//
// Launches per step: embed | per layer { rmsnorm+qkv+rope+append, split-KV attention, [combine: caches beyond 1024
// positions only], merge+o_proj+residual, rmsnorm+gate/up+swiglu, down+residual } | rmsnorm+lm_head(+wave stats) | finish.
// (The embedding rides in layer 0's q|k|v launch for int4 models; for the 32 / 8 / 128 head geometry the attention rides in every q|k|v launch,
// behind an XCD-local seam: fuse_attn() below.)
#include <atomic>
#include <cstring>
#include <new>
#include <vector>

#include "attention.hpp"
#include "tail.hpp"
#include "w4_gemv.hpp"
#include "decoder.hpp"

int embedding_launch(const int32_t *ids, int L, const uint32_t *codes, const void *scales, const void *biases, int V, int H, int dtype,
                     void *out, const float *freqs, const DecState *state, float *rope_cs, int half, hipStream_t st, int bits);

namespace {
thread_local std::string g_last_error;
#if defined(PIE_GEMV_PROF) && PIE_GEMV_PROF == 3
constexpr int PROF_QKV = 256, PROF_O = 288, PROF_GU = 320, PROF_DOWN = 352;  // 32 words each: the fine prologue stamps sit at +8..13
#else
constexpr int PROF_QKV = 16, PROF_O = 20, PROF_GU = 24, PROF_DOWN = 28;
#endif
}
namespace {
int g_knobs[PIE_KNOB_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
}
int pie_knob(int knob) { return knob >= 0 && knob < PIE_KNOB_COUNT ? g_knobs[knob] : PIE_KNOB_DEFAULT; }
namespace pie {
void set_error(const std::string &msg) { g_last_error = msg; }
int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
}  // namespace pie

__global__ void k_advance(DecState *s) { s->pos += 1; }
__global__ void k_set_state(DecState *s, int pos, int token, int cap) {
    if (pos >= 0) s->pos = pos;
    if (token >= 0) s->token = token;
    if (cap >= 0) s->cap = cap;
}

// Picks the split count / merge path for the current cache capacity; returns true when the launch sequence changed.
static bool plan_attention(pie_decoder *d) {
    const pie_decoder_config &c = d->cfg;
    int splits;
    if (c.kv_splits > 0) splits = c.kv_splits > ATTN_MAX_SPLITS ? ATTN_MAX_SPLITS : c.kv_splits;
    else if (d->kv_cap <= d->merge_max_cap) splits = GEMV_ATTN_SPLITS;
    else {
        splits = d->kv_cap / 64;  // >= 64 positions per workgroup, 16 per wave
        const int fill = 256 / c.n_kv_heads > 0 ? 256 / c.n_kv_heads : 1;
        if (splits > fill) splits = fill;
        if (splits > ATTN_MAX_SPLITS) splits = ATTN_MAX_SPLITS;
        if (splits < GEMV_ATTN_SPLITS) splits = GEMV_ATTN_SPLITS;
    }
    const bool combine = splits > GEMV_ATTN_SPLITS;
    const bool changed = splits != d->splits || combine != d->combine;
    d->splits = splits, d->combine = combine;
    return changed;
}

static int dev_alloc(void **p, size_t bytes) {
    PIE_HIP_TRY(hipMalloc(p, bytes));
    PIE_HIP_TRY(hipMemset(*p, 0, bytes));
    return PIE_OK;
}

static void drop_graphs(pie_decoder *d) {
    for (int i = 0; i < 2; ++i)
        if (d->graph[i]) {
            (void)hipGraphExecDestroy(d->graph[i]);
            d->graph[i] = nullptr;
            d->graph_kernels[i] = -1;
        }
}

// The attention launch's idle CUs warm at most this much of o_proj's stream: 9.4 MB (the 8B int4 matrix, all of it) is what the launch's ~5 us
// can carry; uncapped, the 33.5 MB of the dense 8B o_proj stretched the launch itself: dense 2.81 -> 3.00 ms per step, int8 (17.8 MB) 1.74 -> 1.81
// (cap 10: 2.81 / 1.75; off: 2.81 / 1.74; int4: 1.186-1.191 with, 1.190 without)
constexpr int ATTN_WARM_DEFAULT_MB = 10;
// Workgroups with equal blockIdx.x % 8 share an XCD (tools/pilot_probe, every grid size and history): what the fused q|k|v + attention launch
// relies on, so it is checked once per process on the real dispatcher (a partitioned device with one XCD passes trivially).
__global__ void k_xcc_probe(unsigned *out) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc;
}
static bool xcd_classes_hold() {
    static int cached = -1;  // (one device per process)
    if (cached >= 0) return cached == 1;
    unsigned *dev = nullptr, host[256];
    bool ok = hipMalloc((void **)&dev, sizeof(host)) == hipSuccess;
    for (int rep = 0; rep < 3 && ok; ++rep) {
        hipLaunchKernelGGL(k_xcc_probe, dim3(256), dim3(512), 0, 0, dev);
        ok = hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess;
        for (int b = 0; b < 256 && ok; ++b) ok = host[b] == host[b & 7];
    }
    if (dev) (void)hipFree(dev);
    (void)hipGetLastError();
    cached = ok ? 1 : 0;
    return ok;
}

// The step's attention inside the q|k|v launch (w4_gemv.hpp, FUSE): the 32 / 8 / 128 head geometry (kv-group = XCD class), any weight format,
// the short-cache plan on a contiguous or T-page cache (not int8 pages), one GPU.  Knob PIE_KNOB_FUSE_ATTN = 0 keeps the two launches (bit-identical; the tests' cross-check).
// The fused launch's attention workgroups spin until their kv-group has arrived.  They are the last-dispatched half of the grid, the other half leaves
// after arriving, so a launch holds at most 128 workgroup slots while it waits and the chip has 768 for this kernel: up to five such launches in flight
// cannot starve each other's producers.  A process that holds more than four live decoders gets the two-launch form (captured graphs follow at their next step).
static std::atomic<int> g_live_decoders{0};
static bool fuse_attn(const pie_decoder *d, int li) {
    const pie_decoder_config &c = d->cfg;
    if (pie_knob(PIE_KNOB_FUSE_ATTN) == 0 || !d->xcd_ok || !d->seam || g_live_decoders.load() > 4) return false;
    (void)li;  // (any weight format of the q|k|v matrix)
    return !d->tp() && !d->combine && !(d->kv_i8 && d->block_table) && c.n_heads == 32 && c.n_kv_heads == 8 && c.head_dim == 128 && c.hidden <= 4096 &&
           d->splits >= 1 && d->splits <= 4;
}

// What the step's attention launch is given (contiguous or T-page cache, not the int8 pages)
static AttnArgs attn_args(pie_decoder *d, int li) {
    const pie_decoder_config &c = d->cfg;
    const int H = c.hidden, D = c.head_dim, QD = c.n_heads * D;
    const pie_layer_weights &w = d->layers[li];
    AttnArgs a = {};
    a.q = d->qbuf, a.kv_table = d->kv_table, a.layer = li, a.n_layers = c.n_layers, a.state = d->state;
    a.Hq = c.n_heads, a.Hkv = c.n_kv_heads, a.splits = d->splits, a.scale = 1.0f / sqrtf((float)D);
    a.part_acc = d->part_acc, a.part_ml = d->part_ml, a.out = d->attn;
    a.block_table = d->block_table, a.n_pages = d->n_pages, a.bt_stride = 0;
    a.nt_kv = d->combine;  // long-context plan (capacity > 1024): the cache no longer survives in the Infinity Cache between steps
    a.prof = d->pf_sink;
    // the CUs this launch leaves idle warm the Infinity Cache with o_proj's weights (attention.hpp: +1.3 % on the step)
    if (!d->combine) {
        const int f = d->mat_fmt(w.wo);
        a.pf_rows = (256 - c.n_kv_heads * d->splits) / c.n_kv_heads;
        if (a.pf_rows < 0) a.pf_rows = 0;
        a.pf_ptr = (const char *)w.wo, a.pf_sink = d->pf_sink;
        a.pf_bytes = f == PIE_W_INT2_G64 ? pie_w2s_bytes(H, QD) : f == PIE_W_INT6_G64 ? pie_w6s_bytes(H, QD) : f == PIE_W_DENSE ? pie_w16s_bytes(H, QD) : (f == PIE_W_INT8_G64 ? pie_w8s_bytes(H, QD) : (f == PIE_W_INT4_G32 ? pie_w4s32_bytes(H, QD) : (f == PIE_W_INT8_G32 ? pie_w8s32_bytes(H, QD) : pie_w4s_bytes(H, QD))));
        const int cap_mb = pie_knob(PIE_KNOB_ATTN_WARM_MAX_MB) >= 0 ? pie_knob(PIE_KNOB_ATTN_WARM_MAX_MB) : ATTN_WARM_DEFAULT_MB;
        if (a.pf_bytes > (unsigned long long)cap_mb << 20) a.pf_bytes = (unsigned long long)cap_mb << 20;
        if (!a.pf_bytes) a.pf_rows = 0;
    }
    return a;
}

// One launch of the step's sequence (PIE_K_* of include/pie_hip.h); `li` is the layer for per-layer kernels.
int enqueue_kernel(pie_decoder *d, int which, int li, const int *token_ptr, u16 *logits_dst, hipStream_t st, bool embed_here) {
    const pie_decoder_config &c = d->cfg;
    const int H = c.hidden, D = c.head_dim, QD = c.n_heads * D, KVD = c.n_kv_heads * D;
    const pie_layer_weights &w = d->layers[li];
    auto gfmt = [d](const void *m) {  // streaming format of one matrix (per-module quantisation: models/utils.py:99-109)
        const int f = d->mat_fmt(m);
        if (f == PIE_W_INT2_G64) return (int)FMT_W2S;
        if (f == PIE_W_INT6_G64) return (int)FMT_W6S;
        return f == PIE_W_DENSE ? (int)FMT_W16S : (f == PIE_W_INT8_G64 ? FMT_W8S : (f == PIE_W_INT4_G32 ? FMT_W4S32 : (f == PIE_W_INT8_G32 ? FMT_W8S32 : FMT_W4S)));
    };
    const int efmt = d->mat_fmt(d->glob.embed_codes);
    const bool dense_embed = efmt == PIE_W_DENSE;
    switch (which) {
        case PIE_K_EMBED:  // h = embed_tokens(inputs)  (language.py:176)
            if (dense_embed) return pie_embedding_dense(token_ptr, 1, d->glob.embed_codes, d->embed_vocab(), H, c.dtype, d->h, st);
            return embedding_launch(token_ptr, 1, d->glob.embed_codes, d->glob.embed_scales, d->glob.embed_biases, d->embed_vocab(), H, c.dtype, d->h,
                                    d->glob.rope_freqs, d->state, d->rope_cs, D / 2, st, embed_bits(d));
        case PIE_K_QKV: {  // q,k,v = proj(input_layernorm(x)); rope(offset=cache.offset); cache.update_and_fetch  (language.py:83-95)
            GemvArgs a = {};
            a.fmt = gfmt(w.wqkv), a.w = (const char *)w.wqkv, a.K = H, a.N = QD + 2 * KVD;
            a.x = d->h, a.norm_w = (const u16 *)w.attn_norm, a.eps = c.rms_eps;
            a.freqs = d->glob.rope_freqs, a.rope_cs = dense_embed || d->row_is_h ? nullptr : d->rope_cs /* filled by the quantised embedding kernel */, a.state = d->state, a.q_out = d->qbuf, a.kv_table = d->kv_table;
            a.layer = li, a.n_layers = c.n_layers, a.n_heads = c.n_heads, a.n_kv_heads = c.n_kv_heads, a.head_dim = D;
            a.lin_bias = (const u16 *)w.bqkv, a.rope_traditional = c.rope_traditional;
            a.block_table = d->block_table, a.n_pages = d->n_pages;
            const bool i8 = d->kv_i8 && d->block_table;  // int8 pages: the T rows go to the staging page, then get quantised into the sequence's page
            if (i8) a.kv_table = d->kv_table_stage, a.block_table = d->zero_table, a.n_pages = 1;
            a.prof = reinterpret_cast<unsigned long long *>(d->pf_sink) + PROF_QKV;
            if (embed_here) {  // the step's embedding launch folded into this one (embed_in_qkv): x = the token's row, dequantised by every workgroup
                a.x = nullptr, a.rope_cs = nullptr, a.rope_cs_out = d->rope_cs, a.h_out = d->h, a.token = token_ptr;
                a.emb_codes = (const u32 *)d->glob.embed_codes, a.emb_scales = (const u16 *)d->glob.embed_scales, a.emb_biases = (const u16 *)d->glob.embed_biases;
                a.emb_vocab = d->embed_vocab();
            }
            if (fuse_attn(d, li)) a.fuse = 1, a.attn = attn_args(d, li), a.seam = d->seam;  // the step's attention behind an XCD-local seam of this launch
            const int rc = w4s_gemv_launch(c.dtype, embed_here ? PRO_EMBED : PRO_RMSNORM, EPI_ROPE_KV, a, 1, st);
            if (rc || !i8) return rc;
            const size_t blk = (size_t)c.n_kv_heads * PIE_PAGE_TOKENS * D;
            return paged_kv_append_i8_staged_launch(c.dtype, d->kv_stage, d->kv_stage + blk, const_cast<void *>(d->slab_host[li]), d->n_pages, d->block_table, d->max_blocks,
                                                    &d->state->pos, c.n_kv_heads, D, st);
        }
        case PIE_K_ATTN: {  // scaled_dot_product_attention over keys[..., :offset+1, :]  (language.py:98-105, base.py:111-113)
            if (d->kv_i8 && d->block_table) {  // int8 pages: the batch paths' kernel with one sequence whose length comes from the device-side state
                AttnArgs a = {};
                a.q = d->qbuf, a.slab = (const u16 *)d->slab_host[li], a.block_table = d->block_table, a.state = d->state, a.bt_stride = 0, a.n_pages = d->n_pages;
                a.rows = 1, a.Hq = c.n_heads, a.Hkv = c.n_kv_heads, a.scale = 1.0f / sqrtf((float)D);
                int splits = d->kv_cap / 128;  // 4-wave workgroups, >= 128 positions each
                a.splits = splits < 1 ? 1 : (splits > ATTN_MAX_SPLITS ? ATTN_MAX_SPLITS : splits);
                a.part_acc = d->part_acc, a.part_ml = d->part_ml, a.out = d->attn;
                return paged_attn_i8_launch(c.dtype, D, a, st);  // merges its splits itself (k_attn_combine): o_proj reads d->attn
            }
            if (fuse_attn(d, li)) return PIE_OK;  // ran behind the q|k|v launch's seam
            AttnArgs a = attn_args(d, li);
            return attn_decode_launch(c.dtype, D, a, d->combine, st);  // short caches: partials are merged by the o_proj prologue
        }
        case PIE_K_OPROJ: {  // h = x + o_proj(attn)  (language.py:108,151)
            GemvArgs a = {};
            a.fmt = gfmt(w.wo), a.w = (const char *)w.wo, a.K = QD, a.N = H, a.resid = d->h, a.lin_bias = (const u16 *)w.bo;
            // tensor-parallel shard (row-parallel Linear over the local heads): un-rounded fp32 partial, summed over the ranks,
            // THEN the Linear's one rounding and the residual add
            // the one-shot communicator: the push half of the all-reduce rides in this epilogue (EPI_TP_PUSH); RCCL: fp32 partial in memory
            const bool push = d->tp() && tp_comm_push_args(d->comm, &a.tp_peers, &a.tp_epoch, &a.tp_stride);
            const int epi = d->tp() ? (push ? EPI_TP_PUSH : EPI_PARTIAL_F32) : EPI_RESIDUAL;
            a.y32 = d->tp_part, a.tp_rank = c.tp_rank, a.tp_world = c.tp_world;
            const bool merged_attn = d->combine || (d->kv_i8 && d->block_table);  // the attention output is already one T vector
            if (merged_attn) a.x = d->attn;
            else a.part_acc = d->part_acc, a.part_ml = d->part_ml, a.splits = d->splits, a.state = d->state, a.head_dim = D;
            a.prof = reinterpret_cast<unsigned long long *>(d->pf_sink) + PROF_O;
            const int rc = w4s_gemv_launch(c.dtype, merged_attn ? PRO_NONE : PRO_ATTN, epi, a, 1, st);
            return rc || !d->tp() ? rc : tp_allreduce_launch(d->comm, c.dtype, d->tp_part, H, d->h, st, push);
        }
        case PIE_K_GATEUP: {  // silu(gate(post_attention_layernorm(h))) * up(...)  (language.py:127,152)
            GemvArgs a = {};
            a.fmt = gfmt(w.wgateup), a.w = (const char *)w.wgateup, a.K = H, a.N = 2 * c.inter, a.x = d->h, a.norm_w = (const u16 *)w.mlp_norm, a.eps = c.rms_eps;
            a.y = d->act, a.lin_bias = (const u16 *)w.bgateup;
            a.prof = reinterpret_cast<unsigned long long *>(d->pf_sink) + PROF_GU;
            return w4s_gemv_launch(c.dtype, PRO_RMSNORM, EPI_SWIGLU, a, 1, st);
        }
        case PIE_K_DOWN: {  // out = h + down_proj(...)  (language.py:127,153)
            GemvArgs a = {};
            a.fmt = gfmt(w.wdown), a.w = (const char *)w.wdown, a.K = c.inter, a.N = H, a.x = d->act, a.resid = d->h, a.lin_bias = (const u16 *)w.bdown;
            const bool push = d->tp() && tp_comm_push_args(d->comm, &a.tp_peers, &a.tp_epoch, &a.tp_stride);
            a.y32 = d->tp_part, a.tp_rank = c.tp_rank, a.tp_world = c.tp_world;
            a.prof = reinterpret_cast<unsigned long long *>(d->pf_sink) + PROF_DOWN;
            const int rc = w4s_gemv_launch(c.dtype, PRO_NONE, d->tp() ? (push ? EPI_TP_PUSH : EPI_PARTIAL_F32) : EPI_RESIDUAL, a, 1, st);
            return rc || !d->tp() ? rc : tp_allreduce_launch(d->comm, c.dtype, d->tp_part, H, d->h, st, push);
        }
        case PIE_K_LMHEAD: {  // lm_head(norm(h)) (language.py:187,206-209) with per-tile log-softmax partials
            GemvArgs a = {};
            a.fmt = gfmt(d->glob.lm_head), a.w = (const char *)d->glob.lm_head, a.K = H, a.N = c.vocab, a.x = d->h, a.norm_w = (const u16 *)d->glob.final_norm, a.eps = c.rms_eps;
            a.y = logits_dst, a.stats = d->stats;
            return w4s_gemv_launch(c.dtype, PRO_RMSNORM, EPI_LOGITS, a, 1, st);
        }
        case PIE_K_TAIL:  // log-softmax + greedy argmax, advances the device-side state (inference_engine.py:268-271)
            if (d->tp())  // vocabulary-parallel: (max, sum exp, argmax) of every shard -> global log-sum-exp and token
                return tp_tail_launch(d->comm, c.dtype, logits_dst, c.vocab, c.tp_rank * c.vocab, d->stats, d->n_stats, d->tp_part + H, d->logprobs, d->token_out,
                                      d->state, d->history, d->hist_cap, st);
            return logits_tail_launch(c.dtype, logits_dst, c.vocab, d->stats, d->n_stats, d->logprobs, d->token_out, d->state, d->history, d->hist_cap, st);
        default: return pie::fail(PIE_E_ARG, "pie_decoder: unknown kernel id");
    }
}

// The launch sequence of one step.  token_ptr: device int32 holding the input token id.
static int enqueue_step(pie_decoder *d, const int *token_ptr, bool with_logits, u16 *logits_dst, hipStream_t st) {
    // An int4 embedding table in front of an int4 q|k|v matrix: no embedding launch, layer 0's q|k|v launch dequantises the row itself
    // (4.4 us of kernel + a launch boundary per step; same bits: the same dequantisation, RMSNorm tree and RoPE table).
    const bool embed_in_qkv = !d->row_is_h && d->mat_fmt(d->glob.embed_codes) == PIE_W_INT4_G64 && d->mat_fmt(d->layers[0].wqkv) == PIE_W_INT4_G64 && !d->tp() && d->cfg.hidden <= 8 * 8 * GEMV_WAVES * 64;
    int rc = embed_in_qkv || d->row_is_h ? PIE_OK : enqueue_kernel(d, PIE_K_EMBED, 0, token_ptr, logits_dst, st);
    if (rc) return rc;
    for (int li = 0; li < d->cfg.n_layers; ++li)
        for (int k : {PIE_K_QKV, PIE_K_ATTN, PIE_K_OPROJ, PIE_K_GATEUP, PIE_K_DOWN})
            if ((rc = enqueue_kernel(d, k, li, token_ptr, logits_dst, st, embed_in_qkv && li == 0 && k == PIE_K_QKV))) return rc;
    if (!with_logits) {
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, st, d->state);
        PIE_LAUNCH_CHECK();
        return PIE_OK;
    }
    if ((rc = enqueue_kernel(d, PIE_K_LMHEAD, 0, token_ptr, logits_dst, st))) return rc;
    return enqueue_kernel(d, PIE_K_TAIL, 0, token_ptr, logits_dst, st);
}

extern "C" {

const char *pie_hello(void) { return "pie_core \xe2\x9c\x93"; }
#ifndef PIE_BUILD_HASH
#define PIE_BUILD_HASH "unhashed"
#endif
const char *pie_version(void) { return "pie_hip 0.3.0 (gfx950) " PIE_BUILD_HASH; }  // the hash of the sources: proxy_inference_engine_amd/build.py
const char *pie_last_error(void) { return g_last_error.c_str(); }

int pie_set_knob(int knob, int value) {
    PIE_REQUIRE(knob >= 0 && knob < PIE_KNOB_COUNT, PIE_E_ARG, "pie_set_knob: unknown knob");
    g_knobs[knob] = value < 0 ? PIE_KNOB_DEFAULT : value;
    return PIE_OK;
}
int pie_get_knob(int knob) { return pie_knob(knob); }

int pie_device_info(char *name, int name_len, int *n_cus, size_t *hbm_bytes) {
    int dev = 0;
    PIE_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    PIE_HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (n_cus) *n_cus = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return PIE_OK;
}

int pie_decoder_create(const pie_decoder_config *cfg, pie_decoder **out) {
    PIE_REQUIRE(cfg && out, PIE_E_ARG, "pie_decoder_create: null pointer");
    const pie_decoder_config &c = *cfg;
    PIE_REQUIRE(c.dtype == PIE_BF16 || c.dtype == PIE_F16, PIE_E_ARG, "pie_decoder_create: dtype must be PIE_BF16 or PIE_F16");
    PIE_REQUIRE(c.hidden > 0 && c.hidden % 64 == 0 && c.inter > 0 && c.inter % 64 == 0, PIE_E_SHAPE,
                "pie_decoder_create: hidden and intermediate sizes must be multiples of 64 (int4 group size)");
    PIE_REQUIRE(c.head_dim == 64 || c.head_dim == 128, PIE_E_SHAPE, "pie_decoder_create: head_dim must be 64 or 128");
    PIE_REQUIRE(c.n_layers > 0 && c.n_heads > 0 && c.n_kv_heads > 0 && c.n_heads % c.n_kv_heads == 0, PIE_E_SHAPE, "pie_decoder_create: bad head counts");
    PIE_REQUIRE(c.vocab > 0 && c.vocab % 2 == 0, PIE_E_SHAPE, "pie_decoder_create: vocab must be even");
    PIE_REQUIRE(c.hidden <= 32768 && c.inter <= 32768 && c.n_heads * c.head_dim <= 32768, PIE_E_SHAPE, "pie_decoder_create: K > 32768 not supported");
    PIE_REQUIRE(c.weight_format == PIE_W_INT4_G64 || c.weight_format == PIE_W_DENSE || c.weight_format == PIE_W_INT8_G64 || c.weight_format == PIE_W_INT4_G32 || c.weight_format == PIE_W_INT8_G32 || c.weight_format == PIE_W_INT2_G64 || c.weight_format == PIE_W_INT6_G64, PIE_E_ARG,
                "pie_decoder_create: unknown weight_format");
    PIE_REQUIRE(c.tp_world >= 0 && c.tp_world <= 8 && c.tp_rank >= 0 && c.tp_rank < (c.tp_world > 0 ? c.tp_world : 1), PIE_E_ARG,
                "pie_decoder_create: 0 <= tp_rank < tp_world <= 8");
    pie_decoder *d = new (std::nothrow) pie_decoder();
    PIE_REQUIRE(d, PIE_E_HIP, "pie_decoder_create: out of host memory");
    d->cfg = c;
    d->layers.resize(c.n_layers);
    d->layer_set.assign(c.n_layers, 0);
    if (pie_knob(PIE_KNOB_ATTN_MERGE_MAX_CAP) >= 0) d->merge_max_cap = pie_knob(PIE_KNOB_ATTN_MERGE_MAX_CAP);  // capacity up to which o_proj merges the splits
    int rc = PIE_OK;
    d->n_stats = w4s_gemv_waves(c.vocab, c.hidden);
    const int QD = c.n_heads * c.head_dim;
#define PIE_ALLOC(ptr, bytes) if ((rc = dev_alloc((void **)&(ptr), (bytes)))) { pie_decoder_destroy(d); return rc; }
    PIE_ALLOC(d->state, sizeof(DecState));
    PIE_ALLOC(d->kv_table, sizeof(unsigned long long) * 2 * c.n_layers);
    PIE_ALLOC(d->qbuf, 2 * (size_t)QD);
    PIE_ALLOC(d->attn, 2 * (size_t)QD);
    PIE_ALLOC(d->act, 2 * (size_t)c.inter);
    PIE_ALLOC(d->part_acc, 4 * (size_t)c.n_heads * ATTN_MAX_SPLITS * c.head_dim);
    PIE_ALLOC(d->part_ml, 4 * (size_t)c.n_heads * ATTN_MAX_SPLITS * 2);
    PIE_ALLOC(d->stats, sizeof(LogitStat) * (size_t)d->n_stats);
    PIE_ALLOC(d->rope_cs, sizeof(float) * (size_t)c.head_dim);
    PIE_ALLOC(d->pf_sink, 8192);  // the developer builds' stamps (-DPIE_ATTN_PROF: words 2..9; -DPIE_GEMV_PROF: 16 + 4 kind ..)
    PIE_ALLOC(d->seam, 2048);     // 8 x {counter, generation} 64 bytes apart, [256] the give-up flag
    d->xcd_ok = xcd_classes_hold();
    ++g_live_decoders;
    if (d->tp()) PIE_ALLOC(d->tp_part, sizeof(float) * ((size_t)c.hidden + 4));  // RCCL backend: the fp32 partial; [hidden]: the step's log-sum-exp
#undef PIE_ALLOC
    plan_attention(d);
    *out = d;
    return PIE_OK;
}

int pie_decoder_destroy(pie_decoder *d) {
    if (!d) return PIE_OK;
    if (d->seam) --g_live_decoders;  // (counted once the seam buffer exists: a create that failed earlier never was)
    drop_graphs(d);
    prefill_free(d);
    void *ptrs[] = {d->state, d->kv_table, d->qbuf, d->attn, d->act, d->part_acc, d->part_ml, d->stats, d->rope_cs, d->pf_sink, d->seam, d->tp_part, d->kv_stage, d->kv_table_stage,
                    d->zero_table};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete d;
    return PIE_OK;
}

int pie_decoder_set_layer(pie_decoder *d, int layer, const pie_layer_weights *w) {
    PIE_REQUIRE(d && w, PIE_E_ARG, "pie_decoder_set_layer: null pointer");
    PIE_REQUIRE(layer >= 0 && layer < d->cfg.n_layers, PIE_E_ARG, "pie_decoder_set_layer: layer out of range");
    PIE_REQUIRE(w->attn_norm && w->mlp_norm && w->wqkv && w->wo && w->wgateup && w->wdown, PIE_E_ARG, "pie_decoder_set_layer: null weight");
    PIE_REQUIRE(pie_aligned(w->wqkv, 256) && pie_aligned(w->wo, 256) && pie_aligned(w->wgateup, 256) && pie_aligned(w->wdown, 256) &&
                    pie_aligned(w->attn_norm, 16) && pie_aligned(w->mlp_norm, 16),
                PIE_E_ALIGN, "pie_decoder_set_layer: W4S buffers need 256-byte, norm weights 16-byte alignment");
    PIE_REQUIRE(!d->tp() || (!w->bo && !w->bdown), PIE_E_ARG,
                "pie_decoder_set_layer: o_proj / down_proj biases are not supported on a tensor-parallel shard (they would be added once per rank)");
    {
        const int f[4] = {w->fmt_qkv, w->fmt_o, w->fmt_gateup, w->fmt_down};
        const void *m[4] = {w->wqkv, w->wo, w->wgateup, w->wdown};
        for (int i = 0; i < 4; ++i) {
            PIE_REQUIRE(f[i] >= 0 && f[i] <= PIE_W_INT6_G64 + 1, PIE_E_ARG, "pie_decoder_set_layer: unknown per-matrix weight format");
            if (f[i]) d->fmt_map[m[i]] = f[i] - 1;
            else d->fmt_map.erase(m[i]);
        }
    }
    d->layers[layer] = *w;
    d->layer_set[layer] = 1;
    prefill_free(d);  // resident T copies of the previous weights are stale
    drop_graphs(d);
    return PIE_OK;
}

int pie_decoder_set_globals(pie_decoder *d, const pie_global_weights *w) {
    PIE_REQUIRE(d && w, PIE_E_ARG, "pie_decoder_set_globals: null pointer");
    PIE_REQUIRE(w->embed_codes && w->final_norm && w->lm_head && w->rope_freqs, PIE_E_ARG, "pie_decoder_set_globals: null weight");
    PIE_REQUIRE(w->fmt_embed >= 0 && w->fmt_embed <= PIE_W_INT8_G32 + 1 && w->fmt_lm_head >= 0 && w->fmt_lm_head <= PIE_W_INT6_G64 + 1, PIE_E_ARG,
                "pie_decoder_set_globals: unknown per-matrix weight format");
    PIE_REQUIRE((w->fmt_embed ? w->fmt_embed - 1 : d->cfg.weight_format) == PIE_W_DENSE || (w->embed_scales && w->embed_biases), PIE_E_ARG,
                "pie_decoder_set_globals: a quantised embedding needs scales and biases");
    PIE_REQUIRE((w->fmt_embed ? w->fmt_embed - 1 : d->cfg.weight_format) != PIE_W_INT2_G64 && (w->fmt_embed ? w->fmt_embed - 1 : d->cfg.weight_format) != PIE_W_INT6_G64, PIE_E_ARG,
                "pie_decoder_set_globals: the embedding table of a 2- / 6-bit checkpoint is handed over as 4- / 8-bit codes (fmt_embed = PIE_W_INT4_G64 + 1 / PIE_W_INT8_G64 + 1); W2S and W6S are Linear formats");
    if (w->fmt_embed) d->fmt_map[w->embed_codes] = w->fmt_embed - 1;
    else d->fmt_map.erase(w->embed_codes);
    if (w->fmt_lm_head) d->fmt_map[w->lm_head] = w->fmt_lm_head - 1;
    else if (w->lm_head != w->embed_codes) d->fmt_map.erase(w->lm_head);
    PIE_REQUIRE(pie_aligned(w->lm_head, 256) && pie_aligned(w->final_norm, 16) && pie_aligned(w->embed_codes, 16), PIE_E_ALIGN,
                "pie_decoder_set_globals: misaligned weight");
    d->glob = *w;
    d->glob_set = true;
    drop_graphs(d);
    return PIE_OK;
}

int pie_decoder_set_kv(pie_decoder *d, const void *const *k_ptrs, const void *const *v_ptrs, int capacity, void *stream) {
    PIE_REQUIRE(d && k_ptrs && v_ptrs, PIE_E_ARG, "pie_decoder_set_kv: null pointer");
    PIE_REQUIRE(capacity > 0, PIE_E_SHAPE, "pie_decoder_set_kv: capacity must be positive");
    const int L = d->cfg.n_layers;
    std::vector<unsigned long long> tab(2 * (size_t)L);
    for (int i = 0; i < L; ++i) {
        PIE_REQUIRE(k_ptrs[i] && v_ptrs[i] && pie_aligned(k_ptrs[i], 16) && pie_aligned(v_ptrs[i], 16), PIE_E_ALIGN,
                    "pie_decoder_set_kv: null or misaligned cache buffer");
        tab[i] = (unsigned long long)(uintptr_t)k_ptrs[i];
        tab[L + i] = (unsigned long long)(uintptr_t)v_ptrs[i];
    }
    hipStream_t st = (hipStream_t)stream;
    // pageable source: the copy is staged before the call returns, so `tab` may go out of scope
    PIE_HIP_TRY(hipMemcpyAsync(d->kv_table, tab.data(), tab.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(1), 0, st, d->state, -1, -1, capacity);
    PIE_LAUNCH_CHECK();
    d->kv_set = true;
    d->kv_cap = capacity;
    const bool was_paged = d->block_table != nullptr;
    d->block_table = nullptr, d->n_pages = 0;
    if (plan_attention(d) || was_paged) drop_graphs(d);  // the launch sequence changed: captured graphs are stale
    return PIE_OK;
}

int pie_decoder_set_paged_kv(pie_decoder *d, const void *const *slabs, size_t n_pages, const int32_t *block_table, int max_blocks,
                             void *stream) {
    PIE_REQUIRE(d && slabs && block_table, PIE_E_ARG, "pie_decoder_set_paged_kv: null pointer");
    PIE_REQUIRE(n_pages > 0 && n_pages < 0x7FFFFFFFu && max_blocks > 0 && max_blocks <= (1 << 24), PIE_E_SHAPE,
                "pie_decoder_set_paged_kv: n_pages and max_blocks must be positive");
    const int L = d->cfg.n_layers;
    const size_t v_off = (size_t)d->cfg.n_kv_heads * PIE_PAGE_TOKENS * d->cfg.head_dim * 2;  // bytes: K block, then V block
    std::vector<unsigned long long> tab(2 * (size_t)L);
    // int8 pages: the single-sequence step passes the slab bases as launch arguments (baked into a captured graph), so new slabs under an
    // unchanged table, pool size and width must drop the graphs too (T pages are reached through the device-side kv_table)
    bool slabs_changed = (int)d->slab_host.size() != L;
    for (int i = 0; i < L; ++i) {
        PIE_REQUIRE(slabs[i] && pie_aligned(slabs[i], 16), PIE_E_ALIGN, "pie_decoder_set_paged_kv: null or misaligned slab");
        tab[i] = (unsigned long long)(uintptr_t)slabs[i];
        tab[L + i] = tab[i] + v_off;
        slabs_changed = slabs_changed || d->slab_host[i] != slabs[i];
    }
    d->slab_host.assign(slabs, slabs + L);
    hipStream_t st = (hipStream_t)stream;
    if (d->kv_i8) {  // int8 pages (PIE_OPT_KV_I8): the staging page, the table that points every layer at it, and a block table of zeros
        if (!d->kv_stage) {
            PIE_HIP_TRY(hipMalloc((void **)&d->kv_stage, 2 * v_off));
            PIE_HIP_TRY(hipMemset(d->kv_stage, 0, 2 * v_off));
            PIE_HIP_TRY(hipMalloc((void **)&d->kv_table_stage, sizeof(unsigned long long) * 2 * L));
            std::vector<unsigned long long> stab(2 * (size_t)L);
            for (int i = 0; i < L; ++i) stab[i] = (unsigned long long)(uintptr_t)d->kv_stage, stab[L + i] = stab[i] + v_off;
            PIE_HIP_TRY(hipMemcpy(d->kv_table_stage, stab.data(), stab.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        }
        if (d->zero_blocks < max_blocks) {
            if (d->zero_table) (void)hipFree(d->zero_table);
            d->zero_table = nullptr, d->zero_blocks = 0;
            PIE_HIP_TRY(hipMalloc((void **)&d->zero_table, sizeof(int) * (size_t)max_blocks));
            PIE_HIP_TRY(hipMemset(d->zero_table, 0, sizeof(int) * (size_t)max_blocks));
            d->zero_blocks = max_blocks;
            drop_graphs(d);  // the table's address is a launch argument
        }
    }
    const bool blocks_changed = d->max_blocks != max_blocks;
    d->max_blocks = max_blocks;
    const int capacity = max_blocks * PIE_PAGE_TOKENS;
    PIE_HIP_TRY(hipMemcpyAsync(d->kv_table, tab.data(), tab.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(1), 0, st, d->state, -1, -1, capacity);
    PIE_LAUNCH_CHECK();
    d->kv_set = true;
    d->kv_cap = capacity;
    // kernel arguments are baked into captured graphs: a new table pointer or pool size invalidates them
    const bool changed = d->block_table != block_table || d->n_pages != (int)n_pages || blocks_changed || (d->kv_i8 && slabs_changed);
    d->block_table = block_table, d->n_pages = (int)n_pages;
    if (plan_attention(d) || changed) drop_graphs(d);
    return PIE_OK;
}

int pie_decoder_set_state(pie_decoder *d, int offset, int token, void *stream) {
    PIE_REQUIRE(d, PIE_E_ARG, "pie_decoder_set_state: null decoder");
    PIE_REQUIRE(offset >= 0, PIE_E_ARG, "pie_decoder_set_state: negative offset");
    PIE_REQUIRE(token < d->embed_vocab(), PIE_E_ARG, "pie_decoder_set_state: token id out of range");
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(1), 0, (hipStream_t)stream, d->state, offset, token, -1);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

static int ready(pie_decoder *d) {
    PIE_REQUIRE(d, PIE_E_ARG, "pie_decoder: null decoder");
    PIE_REQUIRE(d->glob_set && d->kv_set && d->out_set, PIE_E_STATE, "pie_decoder: set_globals, set_kv and bind_outputs must be called before stepping");
    for (char s : d->layer_set) PIE_REQUIRE(s, PIE_E_STATE, "pie_decoder: a layer has no weights (pie_decoder_set_layer)");
    PIE_REQUIRE(!d->tp() || d->comm, PIE_E_STATE, "pie_decoder: a tensor-parallel shard needs a communicator (pie_decoder_set_comm)");
    return PIE_OK;
}

int pie_decoder_step(pie_decoder *d, int flags, void *stream) {
    int rc = ready(d);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool with_logits = (flags & PIE_STEP_LOGITS) != 0;
    if (!(flags & PIE_STEP_GRAPH)) return enqueue_step(d, &d->state->token, with_logits, d->logits, st);
    const int gi = with_logits ? 1 : 0;
    if (d->graph[gi] && d->graph_fused[gi] && !fuse_attn(d, 0)) drop_graphs(d);  // fusion was withdrawn (knob, a third live decoder): capture the two-launch form
    if (!d->graph[gi]) {
        d->graph_fused[gi] = fuse_attn(d, 0);
        if (d->graph[gi]) {
            (void)hipGraphExecDestroy(d->graph[gi]);
            d->graph[gi] = nullptr;
        }
        // Capture on a private stream (the caller's may be the legacy default stream, which cannot be captured);
        // the instantiated graph is then launched on the caller's stream.
        hipGraph_t g = nullptr;
        hipStream_t cs = nullptr;
        PIE_HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(cs);
            return pie::fail(PIE_E_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
        }
        rc = enqueue_step(d, &d->state->token, with_logits, d->logits, cs);
        e = hipStreamEndCapture(cs, &g);
        (void)hipStreamDestroy(cs);
        if (rc) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        if (e != hipSuccess) return pie::fail(PIE_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        {  // what the graph really holds, for the benchmark's launches_per_step (not a formula)
            size_t n_nodes = 0;
            d->graph_kernels[gi] = -1;
            if (hipGraphGetNodes(g, nullptr, &n_nodes) == hipSuccess) {
                std::vector<hipGraphNode_t> nodes(n_nodes);
                int k = 0;
                if (n_nodes && hipGraphGetNodes(g, nodes.data(), &n_nodes) == hipSuccess) {
                    for (size_t i = 0; i < n_nodes; ++i) {
                        hipGraphNodeType t;
                        if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++k;
                    }
                    d->graph_kernels[gi] = k;
                }
            }
        }
        e = hipGraphInstantiate(&d->graph[gi], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return pie::fail(PIE_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    }
    PIE_HIP_TRY(hipGraphLaunch(d->graph[gi], st));
    return PIE_OK;
}

int pie_decoder_graph_launches(const pie_decoder *d, int flags) {
    if (!d) return -1;
    const int gi = (flags & PIE_STEP_LOGITS) ? 1 : 0;
    return d->graph[gi] ? d->graph_kernels[gi] : -1;
}

int pie_decoder_prefill(pie_decoder *d, const int32_t *ids, int L, void *logits_all, void *stream) {
    int rc = ready(d);
    if (rc) return rc;
    PIE_REQUIRE(ids && L > 0, PIE_E_ARG, "pie_decoder_prefill: need at least one token");
    hipStream_t st = (hipStream_t)stream;
    // a tensor-parallel shard feeds its prompt through the step kernels (2 all-reduces per layer and token); the many-row GEMM
    // path has no collective yet
    // int8 pages: the batched prompt path of ONE sequence reads and writes T pages; such prompts run as decode steps here (fresh prompts go
    // through pie_decoder_prefill_batch, which the Python host does)
    if (L >= prefill_min_rows() && !d->tp() && !(d->kv_i8 && d->block_table)) return prefill_batched(d, ids, nullptr, L, logits_all, st);  // MLX's qmm regime
    for (int l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        u16 *dst = logits_all ? (u16 *)logits_all + (size_t)l * d->cfg.vocab : d->logits;
        if ((rc = enqueue_step(d, ids + l, last || logits_all != nullptr, dst, st))) return rc;
    }
    if (logits_all)  // keep pie_decoder_outputs()'s logits pointer meaningful: last row
        PIE_HIP_TRY(hipMemcpyAsync(d->logits, (u16 *)logits_all + (size_t)(L - 1) * d->cfg.vocab, 2 * (size_t)d->cfg.vocab,
                                   hipMemcpyDeviceToDevice, st));
    return PIE_OK;
}

int pie_decoder_prefill_embeds(pie_decoder *d, const void *embeds, int L, void *logits_all, void *stream) {
    int rc = ready(d);
    if (rc) return rc;
    PIE_REQUIRE(embeds && L > 0, PIE_E_ARG, "pie_decoder_prefill_embeds: need at least one row");
    PIE_REQUIRE(pie_aligned(embeds, 16), PIE_E_ALIGN, "pie_decoder_prefill_embeds: 16-byte alignment required");
    PIE_REQUIRE(!d->tp(), PIE_E_STATE, "pie_decoder_prefill_embeds: not available on a tensor-parallel shard");
    hipStream_t st = (hipStream_t)stream;
    if (!(d->kv_i8 && d->block_table)) return prefill_batched(d, nullptr, embeds, L, logits_all, st);
    // int8 pages: the batched prompt path of one sequence reads and writes T pages, so -- like a prompt of tokens (pie_decoder_prefill) -- the rows
    // run as decode steps, each with its row copied into the residual stream instead of an embedding launch (no RoPE table either: the
    // q|k|v epilogue computes its own angles).  The greedy token the tail leaves in the device-side state is that of the last row.
    const size_t row_bytes = 2 * (size_t)d->cfg.hidden;
    d->row_is_h = true;
    for (int l = 0; l < L && !rc; ++l) {
        u16 *dst = logits_all ? (u16 *)logits_all + (size_t)l * d->cfg.vocab : d->logits;
        if (hipMemcpyAsync(d->h, (const char *)embeds + (size_t)l * row_bytes, row_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = pie::fail(PIE_E_HIP, "pie_decoder_prefill_embeds: copying a row failed");
        else rc = enqueue_step(d, &d->state->token, l == L - 1 || logits_all != nullptr, dst, st);
    }
    d->row_is_h = false;
    if (rc) return rc;
    if (logits_all)
        PIE_HIP_TRY(hipMemcpyAsync(d->logits, (u16 *)logits_all + (size_t)(L - 1) * d->cfg.vocab, 2 * (size_t)d->cfg.vocab, hipMemcpyDeviceToDevice, st));
    return PIE_OK;
}

int pie_decoder_bind_outputs(pie_decoder *d, void *logits, float *logprobs, int32_t *token, void *hidden, int32_t *history,
                             int history_len) {
    PIE_REQUIRE(d && logits && logprobs && token && hidden, PIE_E_ARG, "pie_decoder_bind_outputs: null pointer");
    PIE_REQUIRE(pie_aligned(logits, 16) && pie_aligned(hidden, 16) && pie_aligned(logprobs, 4) && pie_aligned(token, 4), PIE_E_ALIGN,
                "pie_decoder_bind_outputs: logits/hidden need 16-byte alignment");
    PIE_REQUIRE(history_len >= 0 && (history || history_len == 0), PIE_E_ARG, "pie_decoder_bind_outputs: bad history buffer");
    d->logits = (u16 *)logits, d->logprobs = logprobs, d->token_out = token, d->h = (u16 *)hidden;
    d->history = history, d->hist_cap = history_len;
    d->out_set = true;
    drop_graphs(d);
    return PIE_OK;
}

int pie_decoder_set_token_from(pie_decoder *d, const int32_t *token_dev, void *stream) {
    PIE_REQUIRE(d && token_dev, PIE_E_ARG, "pie_decoder_set_token_from: null pointer");
    if (token_dev == d->token_out) return PIE_OK;  // the tail kernel already stored it in the device-side state
    PIE_HIP_TRY(hipMemcpyAsync(&d->state->token, token_dev, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PIE_OK;
}

int pie_decoder_launch_kernel(pie_decoder *d, int which, int layer, void *stream) {
    int rc = ready(d);
    if (rc) return rc;
    PIE_REQUIRE(layer >= 0 && layer < d->cfg.n_layers, PIE_E_ARG, "pie_decoder_launch_kernel: layer out of range");
    PIE_REQUIRE(which != PIE_K_TAIL, PIE_E_ARG, "pie_decoder_launch_kernel: the tail advances the decode state; not launchable alone");
    return enqueue_kernel(d, which, layer, &d->state->token, d->logits, (hipStream_t)stream);
}

// Developer hook (not in the public header): the decoder's internal scratch vectors, for tools/step_bench's bisection.
void *pie_debug_buffer(pie_decoder *d, int which) {
    if (which == 7) return d->pf_sink;
    void *p[] = {d->qbuf, d->attn, d->act, d->part_acc, d->part_ml};
    return which >= 0 && which < 5 ? p[which] : nullptr;
}

int pie_decoder_configure(pie_decoder *d, int option, int value) {
    PIE_REQUIRE(d, PIE_E_ARG, "pie_decoder_configure: null decoder");
    PIE_REQUIRE(option == PIE_OPT_KV_I8, PIE_E_ARG, "pie_decoder_configure: unknown option");
    d->kv_i8 = value != 0;
    plan_attention(d);
    drop_graphs(d);
    return PIE_OK;
}

int pie_decoder_set_comm(pie_decoder *d, pie_comm *c) {
    PIE_REQUIRE(d && c, PIE_E_ARG, "pie_decoder_set_comm: null pointer");
    int rank = 0, world = 0;
    size_t max_elems = 0;
    PIE_REQUIRE(tp_comm_geometry(c, &rank, &world, &max_elems) == PIE_OK, PIE_E_STATE, "pie_decoder_set_comm: the communicator is not connected");
    PIE_REQUIRE(d->tp() && rank == d->cfg.tp_rank && world == d->cfg.tp_world, PIE_E_ARG,
                "pie_decoder_set_comm: rank / world differ from the decoder's tp_rank / tp_world");
    PIE_REQUIRE(max_elems >= (size_t)d->cfg.hidden, PIE_E_SHAPE, "pie_decoder_set_comm: the communicator's slots are shorter than the hidden size");
    d->comm = c;
    drop_graphs(d);
    return PIE_OK;
}

int pie_decoder_status(pie_decoder *d, unsigned *error) {
    PIE_REQUIRE(d && error, PIE_E_ARG, "pie_decoder_status: null pointer");
    PIE_HIP_TRY(hipDeviceSynchronize());
    *error = 0;
    const int rc = d->comm ? pie_comm_status(d->comm, error) : PIE_OK;
    if (!rc && d->seam) {  // the fused q|k|v + attention launch: a kv-group's workgroups were not co-resident within its bounded wait (w4_gemv.hpp, FUSE)
        unsigned gave_up = 0;
        PIE_HIP_TRY(hipMemcpy(&gave_up, d->seam + 256, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (gave_up) *error |= 0x80000000u;
    }
    return rc;
}

static size_t lin_bytes(const pie_decoder *d, const void *m, size_t n, size_t k) {  // algorithmic bytes of one Linear's weights
    const int f = d->mat_fmt(m);
    if (f == PIE_W_DENSE) return n * k * 2;
    if (f == PIE_W_INT2_G64) return n * k / 4 + 2 * (n * k / 64) * 2;  // two-bit codes + the group's 16-bit scale and bias
    if (f == PIE_W_INT6_G64) return n * k * 3 / 4 + 2 * (n * k / 64) * 2;  // six-bit codes likewise
    return n * k / (f == PIE_W_INT8_G64 || f == PIE_W_INT8_G32 ? 1 : 2) + 2 * (n * k / (f == PIE_W_INT4_G32 || f == PIE_W_INT8_G32 ? 32 : 64)) * 2;  // codes + 16-bit scale and bias per group of 64 (32)
}

size_t pie_decoder_kernel_bytes(const pie_decoder *d, int which, int T) {
    if (!d) return 0;
    const pie_decoder_config &c = d->cfg;
    const size_t H = c.hidden, I = c.inter, QD = (size_t)c.n_heads * c.head_dim, KVD = (size_t)c.n_kv_heads * c.head_dim;
    const pie_layer_weights &w = d->layers[0];  // per-kernel figure of layer 0 (layers of one checkpoint normally share their formats)
    switch (which) {
        case PIE_K_QKV: return lin_bytes(d, w.wqkv, QD + 2 * KVD, H) + H * 2 + 2 * KVD * 2;
        case PIE_K_ATTN: return 2 * KVD * 2 * (size_t)T;
        case PIE_K_OPROJ: return lin_bytes(d, w.wo, H, QD);
        case PIE_K_GATEUP: return lin_bytes(d, w.wgateup, 2 * I, H) + H * 2;
        case PIE_K_DOWN: return lin_bytes(d, w.wdown, H, I);
        case PIE_K_LMHEAD: return lin_bytes(d, d->glob.lm_head, c.vocab, H) + H * 2;
        case PIE_K_TAIL: return (size_t)c.vocab * 4;
        default: return 0;
    }
}

size_t pie_decoder_step_bytes(const pie_decoder *d, int T, int with_logits) {
    if (!d) return 0;
    const pie_decoder_config &c = d->cfg;
    const size_t H = c.hidden, I = c.inter, QD = (size_t)c.n_heads * c.head_dim, KVD = (size_t)c.n_kv_heads * c.head_dim;
    // int4 codes + 16-bit scale and bias per group of 64 = 0.5625 B / parameter  (SURVEY.md 8d); dense: 2 B / parameter
    size_t bytes = 0;
    for (const pie_layer_weights &w : d->layers)
        bytes += lin_bytes(d, w.wqkv, QD + 2 * KVD, H) + lin_bytes(d, w.wo, H, QD) + lin_bytes(d, w.wgateup, 2 * I, H) + lin_bytes(d, w.wdown, H, I) +
                 2 * H * 2 /* norm weights */ + 2 * KVD * 2 * (size_t)T /* KV read */ + 2 * KVD * 2 /* KV write */;
    if (with_logits) bytes += lin_bytes(d, d->glob.lm_head, c.vocab, H) + H * 2 + (size_t)c.vocab * 4 /* fp32 logprobs */;
    return bytes;
}

}  // extern "C"
