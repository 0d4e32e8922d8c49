// decoder.hpp -- the decoder object shared by decoder.hip (single-token step) and prefill.hip (batched prompt).
#pragma once
#include <unordered_map>
#include <vector>

#include "attention.hpp"
#include "tail.hpp"
#include "w4_gemv.hpp"

struct pie_decoder {
    pie_decoder_config cfg;
    std::vector<pie_layer_weights> layers;
    std::vector<char> layer_set;
    pie_global_weights glob;
    bool glob_set = false, kv_set = false;
    // device-side state and scratch (owned)
    DecState *state = nullptr;
    unsigned long long *kv_table = nullptr;  // [2*n_layers]
    // paged KV (pie_decoder_set_paged_kv): kv_table holds the layers' slab K / V bases, the caller-owned device block table
    // maps position p to page block_table[p / 64]; nullptr = contiguous per-layer buffers (pie_decoder_set_kv)
    const int *block_table = nullptr;
    int n_pages = 0;
    u16 *qbuf = nullptr, *attn = nullptr, *act = nullptr;
    float *part_acc = nullptr, *part_ml = nullptr, *rope_cs = nullptr;
    unsigned *pf_sink = nullptr;  // scratch for the developer builds' in-kernel stamps
    unsigned *seam = nullptr;     // the fused q|k|v + attention launch's per-XCD arrival counters / generations (w4_gemv.hpp, FUSE); zeroed once
    bool xcd_ok = false;          // the dispatcher places workgroups with equal blockIdx.x % 8 on one XCD (checked at creation)
    // caller-owned outputs (pie_decoder_bind_outputs)
    u16 *h = nullptr, *logits = nullptr;
    float *logprobs = nullptr;
    bool out_set = false;
    LogitStat *stats = nullptr;
    int *token_out = nullptr, *history = nullptr;
    int hist_cap = 0;
    int n_stats = 0, splits = GEMV_ATTN_SPLITS;
    // Attention plan, chosen from the cache capacity (host-known): short caches use <= 4 splits whose partials the o_proj
    // prologue merges (one launch less); long ones spread up to 32 splits per kv-head over the chip and merge them with
    // k_attn_combine -- the scoring loop is VALU work, 4 splits leave it on 32 CUs (83 us per layer at T = 8k, measured).
    bool combine = false;
    int merge_max_cap = 1024, kv_cap = 0;  // measured: merged wins at capacities 512 and 1024, the combine launch from 2048
    bool kv_i8 = false;  // PIE_OPT_KV_I8: the page slabs hold int8 pages (paged_i8.hip)
    bool row_is_h = false;  // the step's input row already sits in `h` (a row of caller-made embeddings): no embedding launch, no RoPE table
    // single-sequence step on int8 pages: the q|k|v GEMV's RoPE + append epilogue writes the new T rows into ONE staging page through a table of
    // staging pointers and an all-zero block table (the kernel is untouched), k_paged_kv_append_i8 quantises them into the sequence's page
    u16 *kv_stage = nullptr;                      // [2][n_kv, 64, D] T
    unsigned long long *kv_table_stage = nullptr;  // [2 * n_layers]: every layer -> the staging K / V block
    int *zero_table = nullptr;                     // [zero_blocks] zeros
    int zero_blocks = 0, max_blocks = 0;
    std::vector<const void *> slab_host;           // the layers' slab bases (host copy of kv_table's first half)
    hipGraphExec_t graph[2] = {nullptr, nullptr};  // [with_logits]
    int graph_kernels[2] = {-1, -1};                // kernel nodes of each captured graph (hipGraphGetNodes)
    bool graph_fused[2] = {false, false};           // the captured graph holds the fused q|k|v + attention launch (re-captured when fusion is withdrawn)
    struct PrefillScratch *prefill = nullptr;       // batched prompt processing (prefill.hip), allocated on first use
    // tensor parallelism (cfg.tp_world > 1): this decoder is one rank's shard; comm is caller-owned (pie_decoder_set_comm)
    pie_comm *comm = nullptr;
    float *tp_part = nullptr;  // [hidden] fp32 partial of the row-parallel Linears, [hidden] = log-sum-exp of the step
    bool tp() const { return cfg.tp_world >= 1; }  // tp_world = 1: the tensor-parallel code path on one rank (tests; the RCCL backend's only test on one card)
    // per-matrix weight format (PIE_W_*), keyed by the packed matrix pointer; matrices not listed use cfg.weight_format
    std::unordered_map<const void *, int> fmt_map;
    int mat_fmt(const void *packed) const {
        auto it = fmt_map.find(packed);
        return it == fmt_map.end() ? cfg.weight_format : it->second;
    }
    bool uniform_int4() const {
        if (cfg.weight_format != PIE_W_INT4_G64) return false;
        for (const auto &kv : fmt_map)
            if (kv.second != PIE_W_INT4_G64) return false;
        return true;
    }
    int embed_vocab() const { return tp() ? cfg.vocab * cfg.tp_world : cfg.vocab; }
};
// embedding_launch's `bits` for the decoder's embedding table: 4 / 8 (64-wide groups) or PIE_EMBED_W4G32
static inline int embed_bits(const pie_decoder *d) {
    const int f = d->mat_fmt(d->glob.embed_codes);
    return f == PIE_W_INT8_G64 ? 8 : (f == PIE_W_INT4_G32 ? PIE_EMBED_W4G32 : (f == PIE_W_INT8_G32 ? PIE_EMBED_W8G32 : 4));
}

// tp_comm.hip: sum over the ranks of data[n] (rank order), then h = T(h + T(sum)) when resid != nullptr
int tp_allreduce_launch(pie_comm *c, int dtype, float *data, int n, u16 *resid, hipStream_t st, bool pushed);
bool tp_comm_push_args(const pie_comm *c, unsigned long long *const **peers, const unsigned **epoch, unsigned *stride);
int tp_tail_launch(pie_comm *c, int dtype, const u16 *logits, int V_local, int vocab_offset, const LogitStat *stats, int n_stats, float *lse, float *logprobs,
                   int *token, DecState *state, int *history, int hist_cap, hipStream_t st);
int tp_comm_geometry(const pie_comm *c, int *rank, int *world, size_t *max_elems);

int paged_kv_append_i8_staged_launch(int dtype, const void *stage_k, const void *stage_v, void *slab, int n_pages, const int *block_table, int max_blocks,
                                     const int *position, int Hkv, int D, hipStream_t st);
// paged_i8.hip: split-KV decode attention over int8 pages (a.slab = the layer's int8 slab, a.ctx_len / a.block_table as for T pages)
int paged_attn_i8_launch(int dtype, int D, const AttnArgs &a, hipStream_t st);

// prefill.hip: batched prompt processing (L >= prefill_min_rows() tokens): per layer the W4S weights are dequantised to T
// and multiplied by hipBLASLt, with hand-written HIP kernels for RoPE + cache append, causal attention and SwiGLU.
int prefill_min_rows();
int prefill_batched(pie_decoder *d, const int32_t *ids, const void *embeds, int L, void *logits_all, hipStream_t st);
void prefill_free(pie_decoder *d);
int enqueue_kernel(pie_decoder *d, int which, int li, const int *token_ptr, u16 *logits_dst, hipStream_t st, bool embed_here = false);  // embed_here: PIE_K_QKV of layer 0 also embeds the token
