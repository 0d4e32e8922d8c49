// step_engine.hip -- the whole decode step as ONE persistent launch on the LDS-DMA weight ring (ring.hpp).
//
// Same arithmetic as the launch sequence of decoder.hip (one Model.__call__ for inputs[1,1], models/llama/language.py:199-210, +
// the lm_head of _inference, engine/inference_engine.py:252-271): the same RMSNorm sums, unit dot products, RoPE, attention
// (attn_head.hpp) and epilogue roundings, bit for bit; the log-softmax tail stays the launched k_logits_finish.
//
// Structure (MI355X_MICROARCH.md, price list: prefetch-credit, ldsdma-fill, allgather, engine-vs-launches):
//   * grid = one workgroup per CU (pinned by > 80 KB of LDS), 8 waves: 2 LOADER waves stream the W4S units of matrix after matrix
//     into per-consumer LDS rings by non-temporal LDS-DMA and never look at the activation chain; 6 CONSUMER waves run the dot
//     products out of LDS (ring_gemv.hpp).  Weights do not depend on activations: while the chip waits at a dependency edge the
//     rings fill (108 KiB = ~4 us of stream per CU), after it the consumers drain them at more than twice the stream rate.
//   * a dependency edge is an all-gather of 8-byte {value, tag} granules: every epilogue lane publishes its two output elements
//     with ONE write-through (sc1) store, every CU's consumer waves sweep the vector with sc1 loads until all tags carry the
//     phase's number, then stage it (fused RMSNorm) into the LDS image exactly as the launched GEMV's prologue does.  No flag, no
//     fence, no grid barrier; tags are unique per (launch, layer, edge), so nothing is re-initialised between launches.
//   * attention: one workgroup per q-head (attn_head.hpp) on CUs spread over the XCDs; it receives q and the new K/V row as
//     granules, has its old K/V rows in registers before they arrive, and publishes the finished head.
// Every wait is bounded (s_memrealtime, 1 s); a give-up sets EngSync::error (sticky) and the grid drains.
#include <vector>

#include "attn_head.hpp"
#include "decoder.hpp"
#include "ring_gemv.hpp"

namespace {

constexpr int ENG_THREADS = RING_WAVES * 64;
constexpr int ENG_MAXP = 6;  // gather passes (64 pieces of 8 elements) per consumer wave: K <= 6 * 6 * 512 = 18432
constexpr unsigned long long ENG_SPIN_TICKS = 100000000ull;  // s_memrealtime ticks (100 MHz): 1 s per wait
enum { K_QKV = 0, K_OPROJ = 1, K_GATEUP = 2, K_DOWN = 3, K_LMHEAD = 4 };
enum { E_QKV = 0, E_ATTN = 1, E_H1 = 2, E_ACT = 3, E_H2 = 4 };  // hand-off edges, by producer

struct EngLayer {
    const char *wqkv, *wo, *wgateup, *wdown;
    const u16 *attn_norm, *mlp_norm;
};
struct EngSync {
    unsigned seq;    // launches so far: makes the granule tags of this launch unique
    unsigned error;  // first give-up code, sticky (0 = none)
    unsigned pad[14];
};
struct EngArgs {
    const EngLayer *layers;
    int n_layers, H, I, n_heads, n_kv, V;
    float eps;
    const u32 *embed_codes;
    const u16 *embed_scales, *embed_biases, *final_norm;
    const char *lm_head;
    const float *freqs;
    DecState *state;
    const int *token_ptr;
    const unsigned long long *kv_table;
    u16 *h, *logits;
    LogitStat *stats;
    unsigned long long *gran;                  // granule buffers: buffer (edge, layer parity) at gran + (2 edge + parity) g_stride
    unsigned g_stride;                         // granules per buffer (the longest edge vector, rounded up)
    int with_logits, rope_traditional;
    EngSync *sync;
    unsigned lds_r0, lds_r1, lds_out, lds_ctl, lds_rope, lds_stage, lds_tab, lds_ring, lds_prof;  // byte offsets in dynamic LDS
    unsigned long long *prof;  // developer build (-DPIE_ENGINE_PROF): per-phase stamps of one workgroup
    int prof_block;
};

typedef __attribute__((ext_vector_type(4))) u32 u32x4_t;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) u32 gu32;
typedef __attribute__((address_space(1))) u16 gu16;
// plain stores through pointers that came out of a table: global, not flat (flat accesses count on lgkmcnt too)
__device__ __forceinline__ void gst32(void *p, u32 v) { *(gu32 *)(unsigned long long)p = v; }
__device__ __forceinline__ void gst16(void *p, u16 v) { *(gu16 *)(unsigned long long)p = v; }

// wave-uniform pointers re-loaded from memory become VGPRs to the compiler and put waterfall loops around buffer accesses: pin them
__device__ __forceinline__ const void *uniform_ptr(const void *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(uniform_ptr(p)), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ uint4 coh_ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {  // sc1: served by L2 / memory, never by this CU's L1
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// ONE aligned 8-byte write-through store: the tag travels with the value, so a reader that sees the tag sees the value (guide, Guideline 16 R2)
__device__ __forceinline__ void store_granule(unsigned long long *g, unsigned tag, unsigned value) {
    __hip_atomic_store((gu64 *)(unsigned long long)g, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Values the loop needs again and again, made opaque once: hipcc otherwise RE-LOADS kernel arguments from memory wherever it runs out
// of SGPRs (80 s_load sites in the first build), and a scalar load that misses while every CU streams costs microseconds
// (MI355X_MICROARCH.md, polling-cost row) -- the epilogues and the norm pass, a few dozen instructions each, took 1.3-1.6 us.
// An opaque value can only be kept or spilled to a VGPR lane, never re-fetched.
__device__ __forceinline__ int opq(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ unsigned opq(unsigned v) {
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ float opq(float v) {
    asm volatile("" : "+s"(v));
    return v;
}
template <class U>
__device__ __forceinline__ U *opq(U *p) {
    unsigned long long v = reinterpret_cast<unsigned long long>(p);
    asm volatile("" : "+s"(v));
    return reinterpret_cast<U *>(v);
}
// Per-layer pointer table in LDS (filled once per launch from the host-written tables): [layer][8] = wqkv, wo, wgateup, wdown,
// attn_norm, mlp_norm, K buffer, V buffer.  Read with one ds_read_b64: no scalar-memory access inside the layer loop.
enum { T_WQKV = 0, T_WO = 1, T_WGATEUP = 2, T_WDOWN = 3, T_ATTN_NORM = 4, T_MLP_NORM = 5, T_KBUF = 6, T_VBUF = 7 };
__device__ __forceinline__ const char *tab_ptr(unsigned tab_lds, int layer, int field) {
    unsigned long long v;
    const unsigned addr = tab_lds + (unsigned)(layer * 8 + field) * 8u;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}

// The CU that runs q-head h's attention: heads spread over the CUs in steps of n_cus / n_heads, offset so that consecutive heads
// land on different XCDs under round-robin placement (speed only; nothing depends on placement).
__host__ __device__ inline int eng_attn_cu(int h, int n_heads, int n_cus) {
    const int s = n_cus / n_heads, sp = s < 8 ? s : 8;
    return h * s + (h % sp);
}

// control words in LDS (byte offsets from lds_ctl): ring FULL[8] | FREE[8] | consumer rendezvous counter | RMSNorm partial sums [8].
// (Thinning the loaders to one fill in flight while their CU gathers -- the guide's gather-pass row -- measured slower here: 1.41 vs 1.37 ms per step.)
constexpr unsigned CTL_SYNC = RING_CTL_USER, CTL_RED = RING_CTL_USER + 16, CTL_BYTES = 128;

template <class T, int HD, int NSH>  // NSH: K slices of the hidden-size inputs (1 or 2)
__global__ void __launch_bounds__(ENG_THREADS, 1) k_step_engine(const EngArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_cus = gridDim.x, cu = blockIdx.x;
    const unsigned lds0 = lds_addr_of(smem);
    const unsigned ctl = lds0 + a.lds_ctl, ring = lds0 + a.lds_ring;
    const int n_layers = opq(a.n_layers), H = opq(a.H), I = opq(a.I), n_heads = opq(a.n_heads), n_kv = opq(a.n_kv), V = opq(a.V);
    const int with_logits = opq(a.with_logits), rope_trad = opq(a.rope_traditional);
    const float eps = opq(a.eps);
    unsigned long long *const gran = opq(a.gran);
    const unsigned g_stride = opq(a.g_stride), tab = lds0 + a.lds_tab, a_lds_ctl = opq(a.lds_ctl);
    const char *const lm_head = opq(a.lm_head);
    const int QD = n_heads * HD, KVD = n_kv * HD, NQ = QD + 2 * KVD;
    if (threadIdx.x < CTL_BYTES / 4) lds_st(ctl + 4 * threadIdx.x, 0u);
    for (int i = threadIdx.x; i < n_layers * 8; i += ENG_THREADS) {  // the per-layer pointer table (vector loads of host-written memory)
        const int li = i >> 3, f = i & 7;
        const unsigned long long v = f < 6 ? reinterpret_cast<const unsigned long long *>(a.layers)[li * 6 + f] : a.kv_table[(f - 6) * n_layers + li];
        *reinterpret_cast<unsigned long long *>(smem + a.lds_tab + (size_t)i * 8) = v;
    }
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + ENG_SPIN_TICKS;
    const int pos = opq(__builtin_amdgcn_readfirstlane(a.state->pos)), cap = opq(__builtin_amdgcn_readfirstlane(a.state->cap));
    const unsigned seq = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned *>(&a.sync->seq));
    const unsigned tag0 = opq(((seq & 0x1FFFFFu) << 11) + 1u);  // + 8 layer + edge; never 0 (the buffers are zeroed once, at creation)
    if (threadIdx.x < HD / 2) {  // cos / sin of pos / freqs[i] (llama/utils.py:42-50), as k_embedding_w4g64 computes them
        float *s_rope = reinterpret_cast<float *>(smem + a.lds_rope);
        const float theta = (float)pos * (1.0f / a.freqs[threadIdx.x]);
        float sn, cs;
        sincosf(theta, &sn, &cs);
        s_rope[2 * threadIdx.x] = cs, s_rope[2 * threadIdx.x + 1] = sn;
    }
#ifdef PIE_ENGINE_PROF
    for (int i = threadIdx.x; i < 132 * 8; i += ENG_THREADS) lds_st(lds0 + a.lds_prof + 4u * i, 0u);
#endif
    __syncthreads();  // the only workgroup barrier: before any LDS-DMA is in flight

    auto give_up = [&](unsigned code) {
        if (lane == 0) {
            unsigned expected = 0;
            __hip_atomic_compare_exchange_strong((gu32 *)(unsigned long long)&a.sync->error, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ================================================================== loader waves
    if (wave < RING_LOADERS) {
        const int n_mats = 4 * n_layers + (with_logits ? 1 : 0);
        auto next = [&](int w, int m, const char **p, unsigned *bytes) -> bool {  // stateless: consumer wave w's share of matrix m
            if (m >= n_mats) return false;
            const int li = m >> 2, kind = m < 4 * n_layers ? (m & 3) : K_LMHEAD;
            int n_pairs, K, gran_pairs = 1;
            const char *base;
            if (kind == K_LMHEAD) n_pairs = V >> 1, K = H, base = lm_head;
            else {
                base = tab_ptr(tab, li, kind);  // T_WQKV .. T_WDOWN = K_QKV .. K_DOWN
                if (kind == K_QKV) n_pairs = NQ >> 1, K = H;
                else if (kind == K_OPROJ) n_pairs = H >> 1, K = QD;
                else if (kind == K_GATEUP) n_pairs = I, K = H, gran_pairs = 2;
                else n_pairs = H >> 1, K = I;
            }
            const RingRun r = ring_run(n_pairs, gran_pairs, n_cus, cu, w);
            const int ns = w4s_slices(K);
            *p = base + (size_t)r.first * ns * W4S_UNIT_BYTES;
            *bytes = (unsigned)(r.count * ns * W4S_UNIT_BYTES);
            return true;
        };
        bool ok = true;
        if (wave == 0) ok = ring_loader<0>(ring, ctl, lane, next, deadline);
        if (RING_LOADERS > 1 && wave == 1) ok = ring_loader<(RING_LOADERS > 1 ? 1 : 0)>(ring, ctl, lane, next, deadline);
        if (!ok) give_up(0x10000u + wave);
        return;
    }

    // ================================================================== consumer waves
    const int cw = wave - RING_LOADERS;           // 0 .. RING_CONSUMERS-1
    const u16 *const final_norm = opq(a.final_norm);
    u16 *const logits = opq(a.logits);
    LogitStat *const stats = opq(a.stats);
    const int ctid = cw * 64 + lane;
    RingCursor cur = ring_cursor(ring, ctl, cw);
    char *img0 = smem + a.lds_r0, *img1 = smem + a.lds_r1;
    float *outp = reinterpret_cast<float *>(smem + a.lds_out) + cw * (2 * GEMV_MAX_RUN);
    const float *s_rope = reinterpret_cast<const float *>(smem + a.lds_rope);
    u16 *stage = reinterpret_cast<u16 *>(smem + a.lds_stage) + cw * 3 * HD;  // this wave's {q | k_new | v_new} rows
    unsigned sync_k = 0;
    bool alive = true;
    int prof_phase = 0;
    (void)prof_phase;  // counts phases for the developer build's stamps
    auto stamp = [&](int slot) {
#ifdef PIE_ENGINE_PROF
        // into LDS, dumped once at the end: a global store here would sit in vmcnt and be waited for by the next counted wait,
        // charging its write-through latency (~1 us while the chip streams) to whatever segment comes next
        if (a.prof && cw == 0 && prof_phase < 132) lds_st(lds0 + a.lds_prof + (unsigned)(prof_phase * 8 + slot) * 4u, (unsigned)__builtin_amdgcn_s_memrealtime());
#endif
    };
    // rendezvous of this CU's consumer waves (the loaders never take part): one LDS counter, monotonic
    auto cons_sync = [&]() {
        ++sync_k;
        lds_inc(ctl + CTL_SYNC, lane == 0 ? 1u : 0u);
        unsigned spins = 0;
        while ((int)(lds_ld_s(ctl + CTL_SYNC) - sync_k * RING_CONSUMERS) < 0) {
            if ((++spins & 1023u) == 0u && __builtin_amdgcn_s_memrealtime() > deadline) {
                alive = false;
                give_up(0x20000u + sync_k);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // this wave's row pairs of each kind of matrix
    const RingRun run_qkv = ring_run(NQ >> 1, 1, n_cus, cu, cw), run_h = ring_run(H >> 1, 1, n_cus, cu, cw), run_gu = ring_run(I, 2, n_cus, cu, cw),
                  run_v = ring_run(V >> 1, 1, n_cus, cu, cw);

    // ---- the activation gather: pieces (8 elements = 4 granules) p = 64 c + lane of pass c = cw + 6 i
    uint4 xv[ENG_MAXP];
    // x arrives as granules of edge buffer `g`; RMSNorm with `norm_w` (nullable) -> image `img` for a K-wide GEMV
    auto gather = [&](const unsigned long long *g, unsigned tag, int K, const u16 *norm_w, char *img, bool local_embed, int token, bool image_idle) {
        const int n_pieces = K >> 3, n_pass = (n_pieces + 63) >> 6;
        const GemvLds L = gemv_lds(K);
        uint4 nv[2];
        if (norm_w) {  // ordinary weights: requested before the sweep
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = cw + RING_CONSUMERS * i;
                if (c < n_pass) {
                    const int j = c * 64 + lane;
                    typedef __attribute__((address_space(1))) const u32x4_t gv4;
                    const u32x4_t v = ((gv4 *)(unsigned long long)norm_w)[j < n_pieces ? j : n_pieces - 1];
                    nv[i] = make_uint4(v.x, v.y, v.z, v.w);
                }
            }
        }
        if (local_embed) {  // h = embed_tokens(token) (language.py:176): every CU dequantises the row itself
            const u32 *row = a.embed_codes + (size_t)token * n_pieces;
            const u16 *srow = a.embed_scales + (size_t)token * (K >> 6), *brow = a.embed_biases + (size_t)token * (K >> 6);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = cw + RING_CONSUMERS * i;
                if (c < n_pass) {
                    const int j = c * 64 + lane, jc = j < n_pieces ? j : n_pieces - 1;
                    const u32 word = row[jc];
                    const float s = T::to_f32(srow[jc >> 3]), b = T::to_f32(brow[jc >> 3]);
                    u32 o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = __fadd_rn(__fmul_rn(s, (float)((word >> (8 * k)) & 0xFu)), b);
                        const float hi = __fadd_rn(__fmul_rn(s, (float)((word >> (8 * k + 4)) & 0xFu)), b);
                        o[k] = pack2<T>(lo, hi);
                    }
                    xv[i] = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = coh_rsrc(g, (unsigned)(K >> 1) * 8u);
            unsigned pending = 0;
#pragma unroll
            for (int i = 0; i < ENG_MAXP; ++i)
                if (cw + RING_CONSUMERS * i < n_pass) pending |= 1u << i;
            while (pending) {
                uint4 d0[ENG_MAXP], d1[ENG_MAXP];
#pragma unroll
                for (int i = 0; i < ENG_MAXP; ++i)
                    if (pending & (1u << i)) {  // wave-uniform
                        const int j = (cw + RING_CONSUMERS * i) * 64 + lane;
                        const unsigned off = (unsigned)(j < n_pieces ? j : n_pieces - 1) * 32u;
                        d0[i] = coh_ld16(rs, off), d1[i] = coh_ld16(rs, off + 16u);
                    }
#pragma unroll
                for (int i = 0; i < ENG_MAXP; ++i)
                    if (pending & (1u << i)) {
                        const bool ok = d0[i].y == tag && d0[i].w == tag && d1[i].y == tag && d1[i].w == tag;
                        if (__all(ok)) {
                            xv[i] = make_uint4(d0[i].x, d0[i].z, d1[i].x, d1[i].z);
                            pending &= ~(1u << i);
                        }
                    }
                if (pending) {
                    if (__builtin_amdgcn_s_memrealtime() > deadline) {
                        alive = false;
                        give_up(0x30000u + (tag & 0x7FFu));
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        if (norm_w) {  // mx.fast.rms_norm (language.py:137-141,168) with the launched prologue's summation tree: pass c = its wave c
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = cw + RING_CONSUMERS * i;
                if (c < 8) {
                    float q = 0.0f;
                    if (c < n_pass) {
                        const bool ok = c * 64 + lane < n_pieces;
                        const u32 v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float lo = lo_f32<T>(v[k]), hi = hi_f32<T>(v[k]);
                            q = fmaf(lo, lo, q);
                            q = fmaf(hi, hi, q);
                        }
                        q = 0.0f + (ok ? q : 0.0f);
                    }
                    q = half_wave_sum(q);
                    q = lane_value(q, 31) + lane_value(q, 63);
                    if (lane == 0) lds_st(ctl + CTL_RED + 4 * c, __builtin_bit_cast(unsigned, q));
                }
            }
        }
        stamp(5);
        // every wave of this CU is through the previous phase (its image is free) and the partial sums are in LDS; the down_proj image
        // has been idle since this CU's waves met in the gate|up gather (its last readers: the previous down phase, this layer's attention)
        if (!image_idle) cons_sync();
        stamp(6);
        if (!alive) return;
        float inv = 1.0f;
        if (norm_w) {
            // the 8 partial sums with ONE wait: an LDS round trip costs ~0.1 us while the DMA and five other waves use the LDS
            const float4 ra = *reinterpret_cast<const float4 *>(smem + a_lds_ctl + CTL_RED), rb = *reinterpret_cast<const float4 *>(smem + a_lds_ctl + CTL_RED + 16);
            const float r[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
            const float tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            inv = 1.0f / sqrtf(tot / (float)K + eps);
        }
        float *sxs = reinterpret_cast<float *>(img + L.off_sx);
#pragma unroll
        for (int i = 0; i < ENG_MAXP; ++i) {
            const int c = cw + RING_CONSUMERS * i;
            if (c < n_pass) {
                const int j = c * 64 + lane;
                const bool ok = j < n_pieces;
                uint4 x = xv[i];
                if (norm_w && i < 2) {
                    const u32 v[4] = {x.x, x.y, x.z, x.w}, gw[4] = {nv[i].x, nv[i].y, nv[i].z, nv[i].w};
                    u32 o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k] = pack2<T>(round_T<T>(lo_f32<T>(v[k]) * inv) * lo_f32<T>(gw[k]), round_T<T>(hi_f32<T>(v[k]) * inv) * hi_f32<T>(gw[k]));
                    x = make_uint4(o[0], o[1], o[2], o[3]);
                }
                float ps = ok ? sum8<T>(x) : 0.0f;
                ps = lanes8_sum(ps);
                if (ok) {
                    *reinterpret_cast<uint4 *>(img + ((size_t)(j & 7) * L.stride + (j >> 3)) * 16) = scale8<T>(x);
                    if ((j & 7) == 0) sxs[j >> 3] = ps;
                }
            }
        }
        stamp(7);
        cons_sync();  // the image is complete
    };

    // ---- layer 0's residual rows of this wave: the embedding (language.py:176), dequantised per element
    int token = __builtin_amdgcn_readfirstlane(*a.token_ptr);
    token = token < 0 ? 0 : (token >= V ? V - 1 : token);
    u32 resid = 0;  // the residual stream's rows (R, R + 1) of lane < run_h.count, R = 2 (run_h.first + lane)
    if (lane < run_h.count) {
        const int R = 2 * (run_h.first + lane);
        const u32 word = a.embed_codes[(size_t)token * (H >> 3) + (R >> 3)];
        const float s = T::to_f32(a.embed_scales[(size_t)token * (H >> 6) + (R >> 6)]), b = T::to_f32(a.embed_biases[(size_t)token * (H >> 6) + (R >> 6)]);
        const int n0 = R & 7;  // even
        resid = pack2<T>(__fadd_rn(__fmul_rn(s, (float)((word >> (4 * n0)) & 0xFu)), b), __fadd_rn(__fmul_rn(s, (float)((word >> (4 * n0 + 4)) & 0xFu)), b));
    }

    // ---- attention role of this CU
    const int a_s = n_cus / n_heads;
    const int a_head = cu / a_s;
    const bool attn_wg = a_head < n_heads && cu == eng_attn_cu(a_head, n_heads, n_cus);
    const int a_kvh = a_head / (n_heads / n_kv);

    const int n_ph = 4 * n_layers + (with_logits ? 1 : 0);
    for (int ph = 0; ph < n_ph && alive; ++ph) {
        const int li = ph >> 2, kind = ph < 4 * n_layers ? (ph & 3) : K_LMHEAD;
        const int par = li & 1;
        const unsigned tagL = tag0 + 8u * (unsigned)li;
        auto gbuf = [&](int edge, int parity) { return gran + (size_t)(2 * edge + parity) * g_stride; };
        stamp(0);
        // ------------------------------------------------------------ the phase's input vector -> LDS image
        int K = H, count = 0, first = 0;
        char *img = img0;
        const unsigned long long *gsrc = nullptr;
        unsigned gtag = 0;
        const u16 *norm_w = nullptr;
        if (kind == K_QKV) {  // input_layernorm (language.py:149); layer 0 reads the embedding row itself
            gsrc = gbuf(E_H2, par ^ 1), gtag = tagL - 8u + E_H2, norm_w = reinterpret_cast<const u16 *>(tab_ptr(tab, li, T_ATTN_NORM));
            first = run_qkv.first, count = run_qkv.count;
        } else if (kind == K_OPROJ) {
            K = QD, gsrc = gbuf(E_ATTN, par), gtag = tagL + E_ATTN;
            first = run_h.first, count = run_h.count;
        } else if (kind == K_GATEUP) {  // post_attention_layernorm (language.py:152)
            gsrc = gbuf(E_H1, par), gtag = tagL + E_H1, norm_w = reinterpret_cast<const u16 *>(tab_ptr(tab, li, T_MLP_NORM));
            first = run_gu.first, count = run_gu.count;
        } else if (kind == K_DOWN) {
            K = I, img = img1, gsrc = gbuf(E_ACT, par), gtag = tagL + E_ACT;
            first = run_h.first, count = run_h.count;
        } else {  // final norm (language.py:187)
            gsrc = gbuf(E_H2, (n_layers - 1) & 1), gtag = tag0 + 8u * (unsigned)(n_layers - 1) + E_H2, norm_w = final_norm;
            first = run_v.first, count = run_v.count;
        }
        gather(gsrc, gtag, K, norm_w, img, ph == 0, token, kind == K_DOWN);
        if (!alive) break;
        stamp(1);
        // ------------------------------------------------------------ the weight stream
        {
            const GemvLds L = gemv_lds(K);
            bool ok;
            if (kind == K_DOWN) ok = ring_consume<T, 0>(cur, smem, lds0, img, L, K, count, outp, lane, deadline);
            else ok = ring_consume<T, NSH>(cur, smem, lds0, img, L, K, count, outp, lane, deadline);
            if (!ok) {
                alive = false;
                give_up(0x40000u + (unsigned)ph);
                break;
            }
        }
        stamp(2);
        // ------------------------------------------------------------ epilogue: one lane per row pair, published as granules
        const bool live = lane < count;
        const int pair = first + lane, R = 2 * pair;
        float va = 0.0f, vb = 0.0f;
        if (live) {
            const float2 o = *reinterpret_cast<const float2 *>(outp + 2 * lane);
            va = o.x, vb = o.y;
        }
        if (kind == K_QKV) {  // RoPE (llama/utils.py:42-50, offset = cache.offset) + cache append (reusable.py:136-137)
            if (live) {
                const float ra = round_T<T>(va), rb = round_T<T>(vb);
                u16 *kdst = reinterpret_cast<u16 *>(const_cast<char *>(tab_ptr(tab, li, T_KBUF)));
                u16 *vdst = reinterpret_cast<u16 *>(const_cast<char *>(tab_ptr(tab, li, T_VBUF)));
                u32 val;
                if (R < QD + KVD) {
                    const int rr = R < QD ? R : R - QD;
                    const int head = rr / HD, ii = (rr % HD) >> 1;
                    const float cs = s_rope[2 * ii], sn = s_rope[2 * ii + 1];
                    const u16 o0 = T::from_f32(__fsub_rn(__fmul_rn(ra, cs), __fmul_rn(rb, sn))), o1 = T::from_f32(__fadd_rn(__fmul_rn(ra, sn), __fmul_rn(rb, cs)));
                    if (R >= QD) {
                        u16 *dst = kdst + ((size_t)head * cap + pos) * HD;
                        const int i0 = rope_trad ? 2 * ii : ii, i1 = rope_trad ? 2 * ii + 1 : ii + HD / 2;
                        gst16(dst + i0, o0), gst16(dst + i1, o1);
                    }
                    val = (u32)o0 | ((u32)o1 << 16);
                } else {
                    const int rr = R - QD - KVD;
                    val = pack2<T>(ra, rb);
                    gst32(vdst + ((size_t)(rr / HD) * cap + pos) * HD + rr % HD, val);
                }
                store_granule(gbuf(E_QKV, par) + pair, tagL + E_QKV, val);
            }
        } else if (kind == K_OPROJ || kind == K_DOWN) {  // h = x + r (language.py:151,153): Linear output rounded to T, then the add rounded to T
            if (live) {
                resid = pack2<T>(lo_f32<T>(resid) + round_T<T>(va), hi_f32<T>(resid) + round_T<T>(vb));
                store_granule(gbuf(kind == K_OPROJ ? E_H1 : E_H2, par) + pair, tagL + (kind == K_OPROJ ? E_H1 : E_H2), resid);
            }
        } else if (kind == K_GATEUP) {  // nn.silu(gate) * up (language.py:127); packed rows (2 i, 2 i + 1) = (gate_i, up_i)
            u32 act = 0;
            if (live) {
                const float gte = round_T<T>(va), up = round_T<T>(vb);
                const float slu = round_T<T>(gte / (1.0f + expf(-gte)));
                act = T::from_f32(slu * up);
            }
            const u32 nb = (u32)__shfl_down((int)act, 1, 64);  // pairs come in twos (ring_run granularity 2): even lanes publish two activations
            if (live && !(lane & 1)) store_granule(gbuf(E_ACT, par) + (pair >> 1), tagL + E_ACT, act | (nb << 16));
        } else {  // logits + per-wave log-softmax partials, as the launched EPI_LOGITS epilogue
            const float oa = round_T<T>(va), ob = round_T<T>(vb);
            if (live) gst32(logits + R, pack2<T>(oa, ob));
            const float mx = live ? fmaxf(oa, ob) : -INFINITY;
            const int ix = live ? (ob > oa ? R + 1 : R) : 0x7fffffff;
            const float wmax = wave_max(mx);
            int cand = (live && mx == wmax) ? ix : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
            float se = live ? expf(oa - wmax) + expf(ob - wmax) : 0.0f;
            se = wave_sum(se);
            if (lane == 0) {
                typedef __attribute__((address_space(1))) u32x4_t g_u32x4;
                const u32x4_t sv = {__builtin_bit_cast(u32, wmax), __builtin_bit_cast(u32, se), (u32)cand, 0u};
                *(g_u32x4 *)(unsigned long long)(stats + cu * RING_CONSUMERS + cw) = sv;
            }
        }
        stamp(3);
        // ------------------------------------------------------------ attention (workgroups with a q-head), between q|k|v and o_proj
        if (kind == K_QKV && attn_wg) {
            constexpr int LPT = HD / 8, TPW = 64 / LPT, NSTR = RING_CONSUMERS * TPW;
            float *s_m = reinterpret_cast<float *>(img1), *s_l = s_m + NSTR, *s_acc = s_l + NSTR;  // aliases the down_proj image: idle until this layer's down phase
            const int ts = lane / LPT, dc = lane % LPT;
            const u16 *kb = reinterpret_cast<const u16 *>(tab_ptr(tab, li, T_KBUF)) + (size_t)a_kvh * cap * HD + dc * 8;
            const u16 *vb2 = reinterpret_cast<const u16 *>(tab_ptr(tab, li, T_VBUF)) + (size_t)a_kvh * cap * HD + dc * 8;
            AttnHeadRing<AH_DEPTH> ringr;
            attn_head_preload<T, HD, RING_CONSUMERS, AH_DEPTH>(ringr, kb, vb2, pos, cw, ts);  // old rows: requested before q exists
            {   // q, the new K row and the new V row of this head: HD / 2 granules each, packed pairs.  EVERY wave fetches its own copy
                // into its own stage (192 granules: nothing) -- a shared stage cost one more rendezvous of the six waves (~0.35 us).
                const unsigned long long *gq = gbuf(E_QKV, par);
                const __amdgpu_buffer_rsrc_t rs = coh_rsrc(gq, (unsigned)(NQ >> 1) * 8u);
                const unsigned tg = tagL + E_QKV;
                const int l2 = lane % (HD / 2);
                unsigned base[3] = {(unsigned)(a_head * HD / 2 + l2), (unsigned)((QD + a_kvh * HD) / 2 + l2), (unsigned)((QD + KVD + a_kvh * HD) / 2 + l2)};
                for (;;) {
                    typedef __attribute__((ext_vector_type(2))) u32 u32x2_t;
                    u32x2_t v[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, base[k] * 8u, 0, 16);
                    const u32 t0 = v[0].y, t1 = v[1].y, t2 = v[2].y;
                    if (__all(t0 == tg && t1 == tg && t2 == tg)) {
                        if (lane < HD / 2) {
                            const u32 d0 = v[0].x, d1 = v[1].x, d2 = v[2].x;
                            const int i0 = rope_trad ? 2 * lane : lane, i1 = rope_trad ? 2 * lane + 1 : lane + HD / 2;
                            stage[i0] = (u16)d0, stage[i1] = (u16)(d0 >> 16);
                            stage[HD + i0] = (u16)d1, stage[HD + i1] = (u16)(d1 >> 16);
                            *reinterpret_cast<u32 *>(stage + 2 * HD + 2 * lane) = d2;
                        }
                        break;
                    }
                    if (__builtin_amdgcn_s_memrealtime() > deadline) {
                        alive = false;
                        give_up(0x50000u + (unsigned)li);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (!alive) break;
#if defined(PIE_ENGINE_PROF) && PIE_ENGINE_PROF == 2
            stamp(5);  // PROF=2: the attention's own segments overwrite the q|k|v gather's: q / k / v arrived | scored | rendezvous
#endif
            attn_head_score<T, HD, RING_CONSUMERS, AH_DEPTH>(ringr, kb, vb2, stage, pos, cw, lane, s_m, s_l, s_acc);
#if defined(PIE_ENGINE_PROF) && PIE_ENGINE_PROF == 2
            stamp(6);
#endif
            cons_sync();
#if defined(PIE_ENGINE_PROF) && PIE_ENGINE_PROF == 2
            stamp(7);
#endif
            if (!alive) break;
            if (ctid < HD / 2) store_granule(gbuf(E_ATTN, par) + a_head * (HD / 2) + ctid, tagL + E_ATTN, attn_head_finish<T, HD, RING_CONSUMERS>(s_m, s_l, s_acc, ctid));
            stamp(4);
        }
        ++prof_phase;
    }
#ifdef PIE_ENGINE_PROF
    if (a.prof && cw == 0)  // every workgroup's consumer wave 0: [cu][phase][8]
        for (int i = lane; i < 132 * 8; i += 64) a.prof[(size_t)cu * (132 * 8) + i] = lds_ld(lds0 + a.lds_prof + 4u * i);
#endif
    // the hidden state (bind_outputs' `hidden`), as the launch sequence leaves it
    if (alive && lane < run_h.count) gst32(a.h + 2 * (run_h.first + lane), resid);
    if (cu == 0 && cw == 0 && lane == 0) {
        *reinterpret_cast<unsigned *>(&a.sync->seq) = seq + 1u;  // plain store: read by the NEXT launch
        if (!with_logits) a.state->pos = pos + 1;              // a prompt token before the last: only the caches were filled
    }
}

}  // namespace

// ======================================================================== host side
struct EngineState {
    EngLayer *layers_dev = nullptr;
    EngSync *sync = nullptr;
    unsigned long long *gran = nullptr;
    size_t gran_count = 0;
    int n_cus = 0;
    bool enabled = true;
    unsigned long long *prof = nullptr;
};

void engine_invalidate(pie_decoder *d) {  // the layer table is rebuilt at the next step; options survive
    if (!d->engine) return;
    if (d->engine->layers_dev) (void)hipFree(d->engine->layers_dev);
    d->engine->layers_dev = nullptr;
}

void engine_free(pie_decoder *d) {
    if (!d->engine) return;
    engine_invalidate(d);
    if (d->engine->sync) (void)hipFree(d->engine->sync);
    if (d->engine->gran) (void)hipFree(d->engine->gran);
    if (d->engine->prof) (void)hipFree(d->engine->prof);
    delete d->engine;
    d->engine = nullptr;
}

static EngineState *engine_state(pie_decoder *d) {
    if (d->engine) return d->engine;
    EngineState *m = new (std::nothrow) EngineState();
    if (!m) return nullptr;
    const char *e = getenv("PIE_STEP_ENGINE");
    m->enabled = e && e[0] == '1';  // opt-in: the launch sequence measures faster (1.33 vs 1.23 ms per 8B step)
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) m->enabled = false;
    else m->n_cus = p.multiProcessorCount;
    d->engine = m;
    return m;
}

void engine_enable(pie_decoder *d, bool on) {
    if (EngineState *m = engine_state(d)) m->enabled = on && m->n_cus > 0;
}

// granules per hand-off buffer: the longest edge vector (q|k|v pairs, or inter / 2), in whole 64-byte lines
static unsigned engine_gran_stride(const pie_decoder_config &c) {
    const unsigned nq = (unsigned)(c.n_heads + 2 * c.n_kv_heads) * c.head_dim / 2, ni = (unsigned)c.inter / 2, nh = (unsigned)c.hidden / 2;
    unsigned n = nq > ni ? nq : ni;
    n = n > nh ? n : nh;
    return (n + 7u) & ~7u;
}

struct EngLdsPlan {
    unsigned r0, r1, out, ctl, rope, stage, tab, ring, prof, total;
};
static EngLdsPlan engine_lds(const pie_decoder_config &c) {
    const int QD = c.n_heads * c.head_dim;
    const int k0 = c.hidden > QD ? c.hidden : QD;
    EngLdsPlan p;
    unsigned off = 0;
    auto take = [&off](unsigned bytes) {
        const unsigned at = off;
        off += (bytes + 15u) & ~15u;
        return at;
    };
    p.r0 = take((unsigned)gemv_lds(k0).off_red);
    unsigned r1 = (unsigned)gemv_lds(c.inter).off_red;
    const unsigned attn = (unsigned)attn_head_lds_bytes(c.head_dim, RING_CONSUMERS);
    p.r1 = take(r1 > attn ? r1 : attn);
    p.out = take(RING_CONSUMERS * 2 * GEMV_MAX_RUN * 4);
    p.ctl = take(CTL_BYTES);
    p.rope = take((unsigned)c.head_dim * 4);
    p.stage = take(RING_CONSUMERS * 3u * (unsigned)c.head_dim * 2);
    p.tab = take(64u * (unsigned)c.n_layers);
    p.ring = take(RING_BYTES);
    p.prof = off;
#ifdef PIE_ENGINE_PROF
    p.prof = take(132 * 8 * 4);
#endif
    p.total = off;
    return p;
}

bool engine_enabled(pie_decoder *d) {
    EngineState *m = engine_state(d);
    return m && m->enabled;
}

// Static part: geometry, weight formats and the LDS budget.
bool engine_config_ok(pie_decoder *d) {
    const pie_decoder_config &c = d->cfg;
    EngineState *ms = engine_state(d);
    if (!ms || ms->n_cus <= 0) return false;
    if (!d->uniform_int4() || d->tp()) return false;
    for (size_t i = 0; i < d->layers.size(); ++i)
        if (d->layer_set[i] && (d->layers[i].bqkv || d->layers[i].bo || d->layers[i].bgateup || d->layers[i].bdown)) return false;
    const int QD = c.n_heads * c.head_dim, G = ms->n_cus;
    if (c.head_dim != 128 && c.head_dim != 64) return false;
    if (G < 8 || c.n_heads > G) return false;
    if (w4s_slices(c.hidden) > 2 || w4s_slices(QD) != w4s_slices(c.hidden)) return false;  // x of the hidden-size inputs lives in registers (ring_gemv.hpp)
    if (c.hidden > 4096 || QD > 4096) return false;                                         // one gather pass per launched staging wave (RMSNorm tree)
    if (c.inter > RING_CONSUMERS * ENG_MAXP * 512) return false;
    const int W = G * RING_CONSUMERS;
    auto run_ok = [W](int n_pairs) { return (n_pairs + W - 1) / W + 1 <= GEMV_MAX_RUN; };
    if (!run_ok((QD + 2 * c.n_kv_heads * c.head_dim) / 2) || !run_ok(c.hidden / 2) || !run_ok(c.inter) || !run_ok(c.vocab / 2)) return false;
    if (W > TAIL_MAX_STATS || d->n_stats < W) return false;
    if (c.n_layers > 250) return false;
    if (engine_lds(c).total > 160u * 1024u) return false;
    return true;
}

// Can this decoder's step run as the persistent launch?  (Everything else keeps the launch sequence of decoder.hip.)
bool engine_supported(pie_decoder *d, bool with_logits) {
    (void)with_logits;
    EngineState *ms = engine_state(d);
    if (!ms || !ms->enabled || !ms->layers_dev) return false;
    return d->head_plan && !d->block_table && engine_config_ok(d);
}

// Device-side tables of the persistent step; allocates, so it runs OUTSIDE stream capture (pie_decoder_step calls it first).
int engine_prepare(pie_decoder *d) {
    EngineState *m = engine_state(d);
    if (!m || !m->enabled || m->layers_dev) return PIE_OK;
    const pie_decoder_config &c = d->cfg;
    std::vector<EngLayer> h(c.n_layers);
    for (int i = 0; i < c.n_layers; ++i) {
        const pie_layer_weights &w = d->layers[i];
        h[i] = {(const char *)w.wqkv, (const char *)w.wo, (const char *)w.wgateup, (const char *)w.wdown, (const u16 *)w.attn_norm, (const u16 *)w.mlp_norm};
    }
    PIE_HIP_TRY(hipMalloc((void **)&m->layers_dev, sizeof(EngLayer) * c.n_layers));
    PIE_HIP_TRY(hipMemcpy(m->layers_dev, h.data(), sizeof(EngLayer) * c.n_layers, hipMemcpyHostToDevice));
    if (!m->sync) {
        PIE_HIP_TRY(hipMalloc((void **)&m->sync, sizeof(EngSync)));
        PIE_HIP_TRY(hipMemset(m->sync, 0, sizeof(EngSync)));
    }
    if (!m->gran) {
        m->gran_count = (size_t)10 * engine_gran_stride(c);  // 5 edges x 2 layer parities
        PIE_HIP_TRY(hipMalloc((void **)&m->gran, m->gran_count * 8));
        PIE_HIP_TRY(hipMemset(m->gran, 0, m->gran_count * 8));  // tag 0 = never published; the kernel's tags start at 1
    }
#ifdef PIE_ENGINE_PROF
    if (!m->prof) {
        PIE_HIP_TRY(hipMalloc((void **)&m->prof, (size_t)m->n_cus * 132 * 8 * 8));
        PIE_HIP_TRY(hipMemset(m->prof, 0, (size_t)m->n_cus * 132 * 8 * 8));
    }
#endif
    return PIE_OK;
}

template <class T>
static int engine_launch_t(const EngArgs &a, int hd, int nsh, int grid, unsigned lds, hipStream_t st) {
#define PIE_ENG_CASE(HD_, NSH_)                                                                                              \
    if (hd == HD_ && nsh == NSH_) {                                                                                         \
        static bool attr_set = false;                                                                                       \
        if (!attr_set) {                                                                                                    \
            PIE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_step_engine<T, HD_, NSH_>),                   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));               \
            attr_set = true;                                                                                                \
        }                                                                                                                   \
        hipLaunchKernelGGL((k_step_engine<T, HD_, NSH_>), dim3(grid), dim3(ENG_THREADS), lds, st, a);                       \
        PIE_LAUNCH_CHECK();                                                                                                 \
        return PIE_OK;                                                                                                      \
    }
    PIE_ENG_CASE(128, 2)
    PIE_ENG_CASE(128, 1)
    PIE_ENG_CASE(64, 2)
    PIE_ENG_CASE(64, 1)
#undef PIE_ENG_CASE
    return pie::fail(PIE_E_SHAPE, "persistent step: head geometry not instantiated");
}

int engine_step_enqueue(pie_decoder *d, const int *token_ptr, bool with_logits, u16 *logits_dst, hipStream_t st) {
    const pie_decoder_config &c = d->cfg;
    EngineState *m = d->engine;
    PIE_REQUIRE(m && m->layers_dev && m->sync && m->gran, PIE_E_STATE, "persistent step: engine_prepare() was not called");
    EngArgs a = {};
    a.layers = m->layers_dev, a.n_layers = c.n_layers, a.H = c.hidden, a.I = c.inter, a.n_heads = c.n_heads, a.n_kv = c.n_kv_heads, a.V = c.vocab;
    a.eps = c.rms_eps;
    a.embed_codes = d->glob.embed_codes, a.embed_scales = (const u16 *)d->glob.embed_scales, a.embed_biases = (const u16 *)d->glob.embed_biases;
    a.final_norm = (const u16 *)d->glob.final_norm, a.lm_head = (const char *)d->glob.lm_head, a.freqs = d->glob.rope_freqs;
    a.state = d->state, a.token_ptr = token_ptr, a.kv_table = d->kv_table;
    a.h = d->h, a.logits = logits_dst, a.stats = d->stats;
    a.gran = m->gran;
    a.g_stride = engine_gran_stride(c);
    PIE_REQUIRE((size_t)10 * a.g_stride <= m->gran_count, PIE_E_STATE, "persistent step: granule arena too small");
    a.with_logits = with_logits ? 1 : 0, a.rope_traditional = c.rope_traditional;
    a.sync = m->sync;
    a.prof = m->prof;
    {
        const char *e = getenv("PIE_ENGINE_PROF_BLOCK");
        a.prof_block = e ? atoi(e) : 0;
    }
    const EngLdsPlan p = engine_lds(c);
    a.lds_r0 = p.r0, a.lds_r1 = p.r1, a.lds_out = p.out, a.lds_ctl = p.ctl, a.lds_rope = p.rope, a.lds_stage = p.stage, a.lds_tab = p.tab, a.lds_ring = p.ring, a.lds_prof = p.prof;
    const int nsh = w4s_slices(c.hidden);
    int rc;
    if (c.dtype == PIE_BF16) rc = engine_launch_t<BF16>(a, c.head_dim, nsh, m->n_cus, p.total, st);
    else rc = engine_launch_t<F16>(a, c.head_dim, nsh, m->n_cus, p.total, st);
    if (rc || !with_logits) return rc;
    // log-softmax + greedy argmax (inference_engine.py:268-271): the launched tail, over one (max, sum exp, argmax) partial per consumer wave
    return logits_tail_launch(c.dtype, logits_dst, c.vocab, d->stats, m->n_cus * RING_CONSUMERS, d->logprobs, d->token_out, d->state, d->history, d->hist_cap, st,
                              &m->sync->error);  // sticky give-up word: the tail then reports token -1
}

void *engine_prof_ptr(pie_decoder *d) { return d->engine ? (void *)d->engine->prof : nullptr; }

int engine_status(pie_decoder *d, unsigned *err) {
    *err = 0;
    if (!d->engine || !d->engine->sync) return PIE_OK;
    PIE_HIP_TRY(hipMemcpy(err, &d->engine->sync->error, sizeof(unsigned), hipMemcpyDeviceToHost));
    return PIE_OK;
}
