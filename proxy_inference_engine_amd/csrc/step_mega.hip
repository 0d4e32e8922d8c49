// step_mega.hip -- the whole decode step as ONE persistent launch.
//
// Same arithmetic as the launch sequence of decoder.hip (one Model.__call__ for inputs[1,1], models/llama/language.py:199-210,
// + the tail of _inference, engine/inference_engine.py:252-271), bit for bit against that sequence with kv_splits = 1: the same
// W4S row-pair -> wave assignment, the same fp32 orders in RMSNorm, the unit dot products, attention and the log-softmax tail.
// What changes is the schedule:
//   * grid = one workgroup per CU: 8 STREAMING waves (the W4S GEMV of w4_gemv.hpp, weights through a register ring) + 4 IO
//     waves; the workgroups stay resident through all phases (embed | per layer: qkv, attention, o_proj, gate/up, down |
//     lm_head | tail);
//   * a phase's output vector travels as 4-byte GRANULES {16-bit tag = edge number, 16-bit value}: one naturally aligned
//     write-through (sc1) store each, so the data is its own flag -- producers never wait for a store acknowledgement and
//     there is no flag hop.  Round 2's first version (sc1 payload, drained, + a flag barrier polled by a sync wave) measured
//     7.6 us per hand-off: store ack ~2 us, workgroup rendezvous ~1, flag visible + poll ~2.7, gather + norm ~1.4
//     (profiles/r02_mega_v1_timeline_*.txt; the launch sequence pays ~3.9 per kernel boundary incl. its prologue);
//   * the IO waves own the consumer side of every hand-off: their vmcnt holds nothing but their polls, so a poll never queues
//     behind weight loads.  They probe one granule quad per wave until it turns up (no polling storm), sweep the vector with
//     16-byte sc1 loads until every tag matches, apply the fused RMSNorm in exactly the launch kernel's reduction order, and
//     publish the activation image into the OTHER of two LDS image buffers, then one s_barrier releases the streaming waves;
//   * weights do not depend on activations: after its epilogue a streaming wave issues the first D units of the NEXT phase's
//     weight stream before it waits for that barrier, so HBM keeps streaming while the hand-off is in flight;
//   * attention (caches up to MEGA_KV_BLOCKS row blocks per wave = 256 positions at head_dim 128): one workgroup per kv head,
//     the old K/V rows are loaded into registers BEFORE the q hand-off (they belong to earlier steps), the new row and q come
//     through LDS from the IO waves' sweep; the residual stream never leaves registers (the wave that produced a row pair of
//     h is the one that adds to it next).
// Every spin is bounded (s_memrealtime); a give-up sets MegaSync::error and the grid drains.
#include <vector>

#include "decoder.hpp"

namespace {

constexpr int MEGA_CONSUMERS = GEMV_WAVES;                 // streaming waves per workgroup
constexpr int MEGA_IO = 4;                                 // IO waves per workgroup
constexpr int MEGA_THREADS = (MEGA_CONSUMERS + MEGA_IO) * 64;
constexpr int MEGA_NT = MEGA_CONSUMERS * 64;               // staging threads of the launch kernel: piece j belongs to thread j % 512
constexpr int MEGA_IOT = MEGA_IO * 64;                     // IO lanes: lane t sweeps pieces t, t + 256, ... (8 values each)
constexpr int MEGA_KV_BLOCKS = 8;                          // K/V row blocks a streaming wave preloads for attention
constexpr unsigned long long MEGA_SPIN_LIMIT = 20000000ull;  // s_memrealtime ticks (100 MHz): 200 ms per wait
constexpr unsigned MEGA_LDS_MIN = 84 * 1024;               // > half of the 160 KiB: one workgroup per CU, whatever else fits

constexpr int MEGA_REPLICAS = 32;
constexpr int MEGA_MAX_WGS = 256;
// Device memory zeroed by a memset node before every launch: the flag barrier of the one hand-off that is not granules
// (lm_head -> tail: fp32 partials and the logits themselves), then the granule vectors.  Flag barrier: every workgroup owns one
// 4-byte flag per REPLICA and stores the barrier number into all copies with ONE wave instruction (lane r -> replica r); a
// workgroup polls replica blockIdx % 32 only, with one 16-byte load per lane (256 flags = 1 KiB): 8 pollers per line, no RMW.
struct MegaSync {
    unsigned flag[MEGA_REPLICAS][MEGA_MAX_WGS];
    unsigned error;  // first give-up code (0 = none)
    unsigned pad[31];
};

struct MegaLayer {
    const char *wqkv, *wo, *wgateup, *wdown;
    const u16 *attn_norm, *mlp_norm;
};

struct MegaArgs {
    const MegaLayer *layers;  // device array [n_layers]
    int n_layers, H, I, n_heads, n_kv, hd, V;
    float eps, attn_scale;
    const u32 *embed_codes;
    const u16 *embed_scales, *embed_biases, *final_norm;
    const char *lm_head;
    const float *freqs;
    DecState *state;
    const int *token_ptr;
    const unsigned long long *kv_table;
    u16 *h, *logits;
    float *logprobs;
    LogitStat *stats;
    int *token_out, *history;
    int hist_cap, with_logits, rope_traditional;
    MegaSync *sync;
    u32 *gh, *gq, *gk, *gv, *ga, *gact;  // granule vectors: hidden [H], q [QD], new k / v rows [KVD], attention out [QD], act [I]
    unsigned img_stride, off_outp, off_att, off_rope, off_ctl;  // dynamic LDS carve-up (bytes)
    unsigned long long *prof;  // developer build (-DPIE_MEGA_PROF): s_memrealtime stamps of one workgroup, [phase][16]
    int prof_block;
    int pace;  // s_sleep(8) repetitions between two prefetched units of a streaming wave
};

typedef __attribute__((ext_vector_type(4))) u32 u32x4_t;
typedef __attribute__((ext_vector_type(2))) u32 u32x2_t;

#define MEGA_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- write-through / L1-bypassing accesses for everything one workgroup hands to another inside the launch
// Every descriptor is built from a wave-uniform pointer; after stores the compiler can no longer prove that for pointers
// it re-loads from memory (layer table, kv_table) and would wrap each buffer access in a waterfall loop: pin them to SGPRs.
__device__ __forceinline__ const void *uniform_ptr(const void *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(uniform_ptr(p)), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ uint4 coh_ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);  // aux 16 = sc1
    return make_uint4(v.x, v.y, v.z, v.w);
}
// NB: never __builtin_bit_cast an ext-vector ELEMENT (v.y ...): hipcc (ROCm 7.2) reads element 0 for every component.  Copy the
// element into a scalar first (found the hard way: the split merge saw acc[0] four times).
__device__ __forceinline__ float coh_f32(u32 bits) { return __builtin_bit_cast(float, bits); }
__device__ __forceinline__ float4 coh_ld16f(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
    const u32 x = v.x, y = v.y, z = v.z, w = v.w;
    return make_float4(coh_f32(x), coh_f32(y), coh_f32(z), coh_f32(w));
}
__device__ __forceinline__ float2 coh_ld8f(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16);
    const u32 x = v.x, y = v.y;
    return make_float2(coh_f32(x), coh_f32(y));
}
__device__ __forceinline__ void coh_st16(__amdgpu_buffer_rsrc_t r, unsigned off, const uint4 &v) {
    u32x4_t x = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, 16);
}
// Global address space, explicitly: a pointer that was read from a table (KV buffers) is a FLAT pointer to the compiler, and flat
// stores retire out of order with respect to vmcnt -- the counted wait behind the epilogue stores would not cover them.
typedef __attribute__((address_space(1))) u32 gu32;
typedef __attribute__((address_space(1))) u16 gu16;
__device__ __forceinline__ u32 coh_ld4(const void *p) { return __hip_atomic_load((const gu32 *)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coh_st4(void *p, u32 v) { __hip_atomic_store((gu32 *)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coh_st2(void *p, u16 v) { __hip_atomic_store((gu16 *)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coh_stf(float *p, float v) { coh_st4(p, __builtin_bit_cast(u32, v)); }

// Host-written tables that no kernel modifies (layer pointers, kv_table): read through the constant address space, i.e. with
// scalar loads -- after the first store of the kernel the compiler no longer treats a global load as invariant, the pointer
// would live in VGPRs and every buffer access built on it would be wrapped in a waterfall loop.
template <class U>
__device__ __forceinline__ U const_load(const U *p) {  // scalar types only
    return *(const __attribute__((address_space(4))) U *)(unsigned long long)p;
}
__device__ __forceinline__ MegaLayer load_layer(const MegaLayer *p) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    MegaLayer l;
    l.wqkv = reinterpret_cast<const char *>(const_load(q + 0)), l.wo = reinterpret_cast<const char *>(const_load(q + 1));
    l.wgateup = reinterpret_cast<const char *>(const_load(q + 2)), l.wdown = reinterpret_cast<const char *>(const_load(q + 3));
    l.attn_norm = reinterpret_cast<const u16 *>(const_load(q + 4)), l.mlp_norm = reinterpret_cast<const u16 *>(const_load(q + 5));
    return l;
}


__device__ __forceinline__ u32 gran(unsigned tag, u16 v) { return (tag << 16) | v; }

enum { K_QKV = 0, K_OPROJ = 1, K_GATEUP = 2, K_DOWN = 3, K_LMHEAD = 4 };

// D: ring depth = units prefetched across a hand-off.  NPT: activation pieces (8 values) per staging thread of the launch
// kernel, K <= NPT * 4096; an IO lane sweeps up to 2 * NPT pieces.
template <class T, int D, int HD, int REP, int NPT>
__global__ void __launch_bounds__(MEGA_THREADS) k_step_mega(const MegaArgs a) {
    constexpr int UB = W4S_UNIT_BYTES;
    constexpr int MAXP = 2 * NPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_io = wave >= MEGA_CONSUMERS;
    const int iow = wave - MEGA_CONSUMERS, iot = iow * 64 + lane;  // IO waves only
    const int G = gridDim.x, W = G * MEGA_CONSUMERS;
    const int gw = blockIdx.x * MEGA_CONSUMERS + wave;             // streaming waves only
    const int tid = threadIdx.x;
    float *s_rope = reinterpret_cast<float *>(smem + a.off_rope);
    int *s_ctl = reinterpret_cast<int *>(smem + a.off_ctl);  // [0] give-up flag, [1] IO-wave rendezvous counter
    float *outp_all = reinterpret_cast<float *>(smem + a.off_outp);
    char *s_att = smem + a.off_att;

    const int H = a.H, QD = a.n_heads * HD, KVD = a.n_kv * HD;
    // written by earlier launches: plain loads; pinned to SGPRs so that everything derived from them stays wave-uniform
    const int pos = __builtin_amdgcn_readfirstlane(a.state->pos), cap = __builtin_amdgcn_readfirstlane(a.state->cap);
    int token = __builtin_amdgcn_readfirstlane(*a.token_ptr);
    token = token < 0 ? 0 : (token >= a.V ? a.V - 1 : token);

    int prof_phase = 0;
    auto stamp = [&](int slot) {
#ifdef PIE_MEGA_PROF
        if (a.prof && lane == 0 && (wave == 0 || wave == MEGA_CONSUMERS) && prof_phase < 512)
            a.prof[((size_t)blockIdx.x * 512 + prof_phase) * 16 + slot + (is_io ? 8 : 0)] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    auto give_up = [&](unsigned code) {
        if (lane == 0) {
            __hip_atomic_store(&a.sync->error, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ctl[0] = 1;
        }
    };
    // workgroup rendezvous; leaves the kernel (all waves together) once any wait of this workgroup has given up
#define MEGA_JOIN()                                                   \
    do {                                                              \
        MEGA_BAR();                                                   \
        if (__builtin_amdgcn_readfirstlane(s_ctl[0]) != 0) return;    \
    } while (0)

    // ================================================================== IO waves: the consumer side of every hand-off
    uint4 val[MAXP];  // swept pieces: 8 values each, packed like the launch kernel's xv[]
    int io_target = 0;
    auto io_sync = [&]() {  // rendezvous of the IO waves only (s_barrier would involve the streaming waves)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&s_ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        io_target += MEGA_IO;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < io_target) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > MEGA_SPIN_LIMIT) {
                give_up(0x10000u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    // Granule vector -> val[]: IO lane t takes pieces t + 256 m.  First ONE lane per wave probes one quad (a different producer
    // per wave and workgroup) until its tag turns up, then every lane sweeps its pieces until all 8 tags of each match.
    auto io_sweep = [&](const u32 *gbuf, int n_pieces, unsigned tag) {
        const int P = n_pieces >> 8;
        const __amdgpu_buffer_rsrc_t r = coh_rsrc(gbuf, (unsigned)n_pieces * 32);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        {   // probes: 16 lanes of each IO wave watch 16 different producers' quads; four probe loads stay in flight, a quarter of a
            // microsecond apart, and are checked in issue order, so that a slow early probe does not delay the detection
            const unsigned quads = (unsigned)n_pieces * 2;
            const unsigned pq = (((unsigned)(blockIdx.x * MEGA_IO + iow) * 16u + (unsigned)(lane & 15)) * 2654435761u >> 7) % quads;
            uint4 pr0 = coh_ld16(r, pq * 16), pr1, pr2, pr3;
            __builtin_amdgcn_s_sleep(4);
            pr1 = coh_ld16(r, pq * 16);
            __builtin_amdgcn_s_sleep(4);
            pr2 = coh_ld16(r, pq * 16);
            __builtin_amdgcn_s_sleep(4);
            pr3 = coh_ld16(r, pq * 16);
            for (;;) {
                if (__all((pr0.x >> 16) == tag)) break;
                pr0 = coh_ld16(r, pq * 16);
                __builtin_amdgcn_s_sleep(4);
                if (__all((pr1.x >> 16) == tag)) break;
                pr1 = coh_ld16(r, pq * 16);
                __builtin_amdgcn_s_sleep(4);
                if (__all((pr2.x >> 16) == tag)) break;
                pr2 = coh_ld16(r, pq * 16);
                __builtin_amdgcn_s_sleep(4);
                if (__all((pr3.x >> 16) == tag)) break;
                pr3 = coh_ld16(r, pq * 16);
                __builtin_amdgcn_s_sleep(4);
                if (__builtin_amdgcn_s_memrealtime() - t0 > MEGA_SPIN_LIMIT || __builtin_amdgcn_readfirstlane(s_ctl[0])) {
                    give_up(0x20000u | tag);
                    return;
                }
            }
        }
        stamp(3);
        unsigned pending = (1u << P) - 1u;
        for (;;) {
            uint4 g0[MAXP], g1[MAXP];
#pragma unroll
            for (int m = 0; m < MAXP; ++m)  // every load of the round first, then the checks: one latency per round, not one per piece
                if (m < P) {
                    const unsigned off = (unsigned)(iot + MEGA_IOT * m) * 32;
                    g0[m] = coh_ld16(r, off), g1[m] = coh_ld16(r, off + 16);
                }
#pragma unroll
            for (int m = 0; m < MAXP; ++m)
                if (m < P && ((pending >> m) & 1u)) {
                    const bool ok = (g0[m].x >> 16) == tag && (g0[m].y >> 16) == tag && (g0[m].z >> 16) == tag && (g0[m].w >> 16) == tag &&
                                    (g1[m].x >> 16) == tag && (g1[m].y >> 16) == tag && (g1[m].z >> 16) == tag && (g1[m].w >> 16) == tag;
                    if (ok) {
                        val[m] = make_uint4((g0[m].x & 0xffffu) | (g0[m].y << 16), (g0[m].z & 0xffffu) | (g0[m].w << 16), (g1[m].x & 0xffffu) | (g1[m].y << 16),
                                            (g1[m].z & 0xffffu) | (g1[m].w << 16));
                        pending &= ~(1u << m);
                    }
                }
            if (!__any(pending != 0)) return;
            if (__builtin_amdgcn_s_memrealtime() - t0 > MEGA_SPIN_LIMIT || __builtin_amdgcn_readfirstlane(s_ctl[0])) {
                give_up(0x30000u | tag);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    // val[] -> activation image `img` of a K-wide input: optional RMSNorm (nn.RMSNorm, language.py:137-141,168) in the launch
    // kernel's reduction order -- staging thread r of that kernel owns pieces r + 512 i, so IO lane t holds the pieces of its
    // threads t (even m) and t + 256 (odd m), i.e. of its waves iow and iow + 4 -- then scale, group sums, LDS image.
    uint4 nvv[MAXP];  // the norm weights of the pieces in val[], loaded BEFORE the sweep (ordinary weights: no hand-off)
    auto io_load_norm = [&](const u16 *norm_w, int K) {
        const int P = K >> 11;
#pragma unroll
        for (int m = 0; m < MAXP; ++m)
            if (m < P) {
                typedef __attribute__((address_space(1))) const u32x4_t gv4;
                const u32x4_t v = ((gv4 *)(unsigned long long)norm_w)[iot + MEGA_IOT * m];
                nvv[m] = make_uint4(v.x, v.y, v.z, v.w);
            }
    };
    auto io_publish = [&](int K, const u16 *norm_w, char *img) {
        const GemvLds L = gemv_lds(K);
        float *sxs = reinterpret_cast<float *>(img + L.off_sx);
        float *red = reinterpret_cast<float *>(img + L.off_red);
        const int P = K >> 11;  // pieces per IO lane = (K / 8) / 256
        if (norm_w) {
            float ssq[2] = {0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < MAXP; ++m)
                if (m < P) {
                    const u32 v[4] = {val[m].x, val[m].y, val[m].z, val[m].w};
                    float q = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = lo_f32<T>(v[k]), hi = hi_f32<T>(v[k]);
                        q = fmaf(lo, lo, q);
                        q = fmaf(hi, hi, q);
                    }
                    ssq[m & 1] += q;
                }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                float s = half_wave_sum(ssq[h2]);
                s = lane_value(s, 31) + lane_value(s, 63);
                if (lane == 0) red[iow + 4 * h2] = s;
            }
            io_sync();
            const float4 ra = *reinterpret_cast<const float4 *>(red), rb = *reinterpret_cast<const float4 *>(red + 4);
            const float tot = ((ra.x + ra.y) + (ra.z + ra.w)) + ((rb.x + rb.y) + (rb.z + rb.w));
            const float inv = 1.0f / sqrtf(tot / (float)K + a.eps);
#pragma unroll
            for (int m = 0; m < MAXP; ++m)
                if (m < P) {
                    const u32 v[4] = {val[m].x, val[m].y, val[m].z, val[m].w}, g[4] = {nvv[m].x, nvv[m].y, nvv[m].z, nvv[m].w};
                    u32 o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k] = pack2<T>(round_T<T>(lo_f32<T>(v[k]) * inv) * lo_f32<T>(g[k]), round_T<T>(hi_f32<T>(v[k]) * inv) * hi_f32<T>(g[k]));
                    val[m] = make_uint4(o[0], o[1], o[2], o[3]);
                }
        }
#pragma unroll
        for (int m = 0; m < MAXP; ++m)
            if (m < P) {
                const int j = iot + MEGA_IOT * m;
                float ps = sum8<T>(val[m]);
                ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0xB1, 0xF, 0xF, true);
                ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0x4E, 0xF, 0xF, true);
                ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0x141, 0xF, 0xF, true);
                *reinterpret_cast<uint4 *>(img + ((size_t)(j & 7) * L.stride + (j >> 3)) * 16) = scale8<T>(val[m]);
                if ((j & 7) == 0) sxs[j >> 3] = ps;
            }
    };

    // ================================================================== streaming waves: the weight stream
    uint4 c0[D], c1[D];
    u32 sb[D];
    __amdgpu_buffer_rsrc_t wrsrc = coh_rsrc(nullptr, 0);
    unsigned woff0 = 0, pstride32 = 0;
    int i_ns = 1, i_run = 0, i_chunks = 0, iss_sl = 0, iss_pl = 0;
    bool i_ragged = false;
    auto stream_open = [&](const char *w, int N, int K) {  // wave-uniform arguments
        const int n_pairs = N >> 1;
        i_ns = w4s_slices(K);
        wrsrc = coh_rsrc(w, (unsigned)((size_t)n_pairs * i_ns * UB));
        woff0 = (unsigned)((size_t)gw * i_ns * UB) + lane * 16;
        pstride32 = (unsigned)((size_t)W * i_ns * UB);
        i_run = (w && gw < n_pairs) ? (n_pairs - gw + W - 1) / W : 0;
        i_chunks = (K + 63) >> 6;
        i_ragged = (i_chunks & 31) != 0;
        iss_sl = 0, iss_pl = 0;
    };
    auto issue = [&](int d) {  // ring slot d <- the next unit of the open matrix (out-of-range offset = dropped load)
        unsigned off = iss_pl < i_run ? woff0 + (unsigned)iss_pl * pstride32 + (unsigned)iss_sl * UB : 0xFFFFF000u;
        if (i_ragged && iss_sl * 32 + (lane & 31) >= i_chunks) off = 0xFFFFF000u;
        if (++iss_sl == i_ns) iss_sl = 0, ++iss_pl;
        const u32x4_t v0 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 2);  // nt: every weight byte is read once per step
        const u32x4_t v1 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + 1024, 0, 2);
        c0[d] = make_uint4(v0.x, v0.y, v0.z, v0.w);
        c1[d] = make_uint4(v1.x, v1.y, v1.z, v1.w);
        sb[d] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, off + 2048 - lane * 12, 0, 2);
    };
    // The first D units of the next phase, in flight across the hand-off -- PACED: a CU's memory pipe moves ~27 GB/s, and issued as
    // one burst the 8 waves' D units (74 KB) queue ~3 us of traffic in front of the IO waves' polls (measured: hand-offs of 8-11 us).
    // One unit per wave every `pace` x 64 cycles keeps the pipe busy and the queue short.
    bool deferred = false;
    auto prefetch_next = [&](const char *w, int N, int K) {
        stream_open(w, N, K);
        if (a.pace < 0) {  // developer switch: no prefetch across the hand-off (the ring fills after the image is published)
            deferred = true;
            return;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            issue(d);
            if (d + 1 < D)
                for (int z = 0; z < a.pace; ++z) __builtin_amdgcn_s_sleep(8);
        }
    };
    float *outp = outp_all + (is_io ? 0 : wave) * (2 * GEMV_MAX_RUN);
    int c_run = 0;
    // one phase's units against the published image; the row sums of this wave's pairs are left in LDS (outp)
    auto stream = [&](int N, int K, const char *img) {
        const GemvLds L = gemv_lds(K);
        const float *sxs = reinterpret_cast<const float *>(img + L.off_sx);
        const int n_groups = K >> 6, ns = w4s_slices(K), n_pairs = N >> 1;
        c_run = gw < n_pairs ? (n_pairs - gw + W - 1) / W : 0;
        const int n_units = c_run * ns;
        float acc = 0.0f;
        int sl = 0, pl = 0;
        // The prefetched units were issued before the hand-off and have long landed; retiring everything HERE leaves only
        // weight loads pending inside the loop, so hipcc emits counted vmcnt waits there instead of vmcnt(0) at the loop head
        // (any store or scratch access still pending at the head makes it treat vmcnt as out of order).
        if (deferred) {
            deferred = false;
#pragma unroll
            for (int d = 0; d < D; ++d) issue(d);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int base = 0; base < n_units; base += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (base + d < n_units) {  // wave-uniform
                    const int g = sl * 32 + (lane & 31);
                    const bool gvalid = g < n_groups;
                    const int gc = gvalid ? g : n_groups - 1;
                    u32 xr[32];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(img + ((size_t)r * L.stride + gc) * 16);
                        xr[4 * r + 0] = v.x, xr[4 * r + 1] = v.y, xr[4 * r + 2] = v.z, xr[4 * r + 3] = v.w;
                    }
                    const float sx = sxs[gc];
                    const float dd = w4s_unit_dot<T>(c0[d], c1[d], xr);
                    const float scale = lo_f32<T>(sb[d]), bias = hi_f32<T>(sb[d]);
                    const float pr = fmaf(scale, dd * T::DSCALE - T::OFFSET * sx, bias * sx);
                    acc += gvalid ? pr : 0.0f;
                    if (++sl == ns) {
                        const float tot = half_wave_sum(acc);
                        if ((lane & 31) == 31) outp[2 * pl + (lane >> 5)] = tot;
                        acc = 0.0f, sl = 0, ++pl;
                    }
                }
                issue(d);  // unconditional: past the end of the matrix the offset is out of range and the load is dropped
            }
        }
    };

    // ================================================================== setup
    const int N_qkv = QD + 2 * KVD;
    const bool attn_wg = (int)blockIdx.x < a.n_kv;  // one attention workgroup per kv head
    const int kvh = blockIdx.x;                     // its kv head
    const float sl2 = a.attn_scale * ATTN_LOG2E;
    u32 res = 0;  // streaming lane l < run_h: the residual-stream pair (rows 2 (gw + l W), +1) -- it never leaves this register
    const int run_h = (!is_io && gw < (H >> 1)) ? ((H >> 1) - gw + W - 1) / W : 0;
    char *img0 = smem, *img1 = smem + a.img_stride;
    if (tid == 0) s_ctl[0] = 0, s_ctl[1] = 0;
    MegaLayer Lw = load_layer(a.layers);
    {   // h = embed_tokens(token) (language.py:176)
        const int words = H >> 3;
        const u32 *row = a.embed_codes + (size_t)token * words;
        const u16 *srow = a.embed_scales + (size_t)token * (H >> 6), *brow = a.embed_biases + (size_t)token * (H >> 6);
        if (is_io) {
            const int P = H >> 11;
#pragma unroll
            for (int m = 0; m < MAXP; ++m)
                if (m < P) {
                    const int j = iot + MEGA_IOT * m;
                    const u32 word = row[j];
                    const float s = T::to_f32(srow[j >> 3]), b = T::to_f32(brow[j >> 3]);
                    u32 o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = __fadd_rn(__fmul_rn(s, (float)((word >> (8 * k)) & 0xFu)), b);
                        const float hi = __fadd_rn(__fmul_rn(s, (float)((word >> (8 * k + 4)) & 0xFu)), b);
                        o[k] = pack2<T>(lo, hi);
                    }
                    val[m] = make_uint4(o[0], o[1], o[2], o[3]);
                }
        } else {
            if (tid < HD / 2) {  // cos / sin of pos / freqs[i] (llama/utils.py:42-50), as k_embedding_w4g64 computes them
                const float theta = (float)pos * (1.0f / a.freqs[tid]);
                float sn, cs;
                sincosf(theta, &sn, &cs);
                s_rope[2 * tid] = cs, s_rope[2 * tid + 1] = sn;
            }
            if (lane < run_h) {  // this lane's residual pair of the embedding row
                const int R = 2 * (gw + lane * W);
                const u32 word = row[R >> 3];
                const float s = T::to_f32(srow[R >> 6]), b = T::to_f32(brow[R >> 6]);
                const int sh = 4 * (R & 7);
                res = pack2<T>(__fadd_rn(__fmul_rn(s, (float)((word >> sh) & 0xFu)), b), __fadd_rn(__fmul_rn(s, (float)((word >> (sh + 4)) & 0xFu)), b));
            }
            prefetch_next(Lw.wqkv, N_qkv, H);
        }
    }
    MEGA_BAR();  // s_ctl initialised (the IO rendezvous below uses it)
    // Two programs in one kernel, one per role, meeting only at the workgroup barriers (the same sequence of MEGA_JOIN /
    // MEGA_BAR in both).  Sharing one loop made the register allocator carry both roles' state everywhere: 617 spills and
    // descriptors in VGPRs, i.e. waterfall loops inside the stream.
    if (is_io) {
        io_load_norm(Lw.attn_norm, H);
        io_publish(H, Lw.attn_norm, img0);
        int kind = K_QKV, li = 0, ph = 0;  // ph: GEMV phase counter (image buffer = ph & 1)
        unsigned edge = 0;                 // granule tag of the last hand-off written
        for (;;) {
            char *img = (ph & 1) ? img1 : img0, *img_next = (ph & 1) ? img0 : img1;
            MEGA_JOIN();  // the image of this phase is published
            stamp(0);
            const int N = kind == K_QKV ? N_qkv : kind == K_GATEUP ? 2 * a.I : kind == K_LMHEAD ? a.V : H;
            const int K = kind == K_OPROJ ? QD : kind == K_DOWN ? a.I : H;
            const unsigned tag = ++edge;  // the hand-off this phase produces
            // what follows this phase
            bool more = true;
            int nkind = kind + 1, nli = li;
            if (kind == K_DOWN) {
                if (li + 1 < a.n_layers) nkind = K_QKV, nli = li + 1;
                else if (a.with_logits) nkind = K_LMHEAD;
                else more = false;
            } else if (kind == K_LMHEAD) {
                more = false;
            }
            const MegaLayer Ln = nli != li ? load_layer(a.layers + nli) : Lw;
            // ------------------------------------------------------------ IO waves: the next phase's input
            if (kind == K_QKV) {
                if (attn_wg) {  // q heads of this kv head + its new k / v row -> LDS
                    u16 *s_q = reinterpret_cast<u16 *>(s_att);
                    constexpr int NQ = REP * HD / 4, NKV = HD / 4;  // quads of 4 granules
                    const __amdgpu_buffer_rsrc_t rq = coh_rsrc(a.gq + kvh * REP * HD, REP * HD * 4), rk = coh_rsrc(a.gk + kvh * HD, HD * 4),
                                                 rv = coh_rsrc(a.gv + kvh * HD, HD * 4);
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (int qd = iot; qd < NQ + 2 * NKV; qd += MEGA_IOT) {
                        for (;;) {
                            const uint4 g = qd < NQ ? coh_ld16(rq, (unsigned)qd * 16) : qd < NQ + NKV ? coh_ld16(rk, (unsigned)(qd - NQ) * 16) : coh_ld16(rv, (unsigned)(qd - NQ - NKV) * 16);
                            if ((g.x >> 16) == tag && (g.y >> 16) == tag && (g.z >> 16) == tag && (g.w >> 16) == tag) {
                                *reinterpret_cast<uint2 *>(s_q + qd * 4) = make_uint2((g.x & 0xffffu) | (g.y << 16), (g.z & 0xffffu) | (g.w << 16));
                                break;
                            }
                            if (__builtin_amdgcn_s_memrealtime() - t0 > MEGA_SPIN_LIMIT) {
                                if (true) { __hip_atomic_store(&a.sync->error, 0x40000u | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_ctl[0] = 1; }
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                }
                MEGA_JOIN();  // q staged
                if (attn_wg) MEGA_BAR();  // the attention merge rendezvous of the streaming waves
                ++edge;
                io_sweep(a.ga, QD >> 3, edge);
                stamp(1);
                io_publish(QD, nullptr, img_next);
            } else if (more) {
                if (nkind == K_DOWN) {
                    io_sweep(a.gact, a.I >> 3, tag);
                    stamp(1);
                    io_publish(a.I, nullptr, img_next);
                } else {
                    const u16 *nw = nkind == K_QKV ? Ln.attn_norm : nkind == K_GATEUP ? Lw.mlp_norm : a.final_norm;
                    io_load_norm(nw, H);
                    io_sweep(a.gh, H >> 3, tag);
                    stamp(1);
                    io_publish(H, nw, img_next);
                }
            }
            stamp(2);
            if (!more) break;
            kind = nkind, li = nli, Lw = Ln;
            ++ph, ++prof_phase;
        }
        if (!a.with_logits) return;
        MEGA_BAR();  // every streaming wave has drained its logits / partial stores
        if (wave == MEGA_CONSUMERS) {  // the flag barrier, by one IO wave
            if (lane < MEGA_REPLICAS) coh_st4(&a.sync->flag[lane][blockIdx.x], 1u);
            const __amdgpu_buffer_rsrc_t fr = coh_rsrc(&a.sync->flag[blockIdx.x % MEGA_REPLICAS][0], MEGA_MAX_WGS * 4);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const uint4 f = coh_ld16(fr, lane * 16);
                const int b0 = lane * 4;
                const bool ok = (b0 >= G || f.x >= 1u) && (b0 + 1 >= G || f.y >= 1u) && (b0 + 2 >= G || f.z >= 1u) && (b0 + 3 >= G || f.w >= 1u);
                if (__all(ok)) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > MEGA_SPIN_LIMIT || __any(coh_ld4(&a.sync->error) != 0)) {
                    give_up(0x50000u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        MEGA_JOIN();
        MEGA_BAR();
        MEGA_BAR();  // the tail's two reductions (streaming waves)
        return;
    }

    // ================================================================== streaming waves
    int kind = K_QKV, li = 0, ph = 0;  // ph: GEMV phase counter (image buffer = ph & 1)
    unsigned edge = 0;                 // granule tag of the last hand-off written
    for (;;) {
        char *img = (ph & 1) ? img1 : img0, *img_next = (ph & 1) ? img0 : img1;
        MEGA_JOIN();  // the image of this phase is published
        stamp(0);
        const int N = kind == K_QKV ? N_qkv : kind == K_GATEUP ? 2 * a.I : kind == K_LMHEAD ? a.V : H;
        const int K = kind == K_OPROJ ? QD : kind == K_DOWN ? a.I : H;
        const unsigned tag = ++edge;  // the hand-off this phase produces
        // what follows this phase
        bool more = true;
        int nkind = kind + 1, nli = li;
        if (kind == K_DOWN) {
            if (li + 1 < a.n_layers) nkind = K_QKV, nli = li + 1;
            else if (a.with_logits) nkind = K_LMHEAD;
            else more = false;
        } else if (kind == K_LMHEAD) {
            more = false;
        }
        const MegaLayer Ln = nli != li ? load_layer(a.layers + nli) : Lw;
        // ------------------------------------------------------------ stream + epilogue (one lane per row pair)
        stream(N, K, img);
        stamp(2);
        const bool live = lane < c_run;
        const int pair = gw + lane * W, R = 2 * pair;
        float va = 0.0f, vb = 0.0f;
        if (live) {
            const float2 o = *reinterpret_cast<const float2 *>(outp + 2 * lane);
            va = o.x, vb = o.y;
        }
        if (kind == K_QKV) {  // RoPE + cache append (language.py:83-95); q and the new k / v rows also travel as granules
            if (live) {
                const float ra = round_T<T>(va), rb = round_T<T>(vb);
                u16 *kdst = reinterpret_cast<u16 *>(const_load(a.kv_table + li));
                u16 *vdst = reinterpret_cast<u16 *>(const_load(a.kv_table + a.n_layers + li));
                if (R < QD + KVD) {
                    const int rr = R < QD ? R : R - QD;
                    const int head = rr / HD, ii = (rr % HD) >> 1;
                    const float cs = s_rope[2 * ii], sn = s_rope[2 * ii + 1];
                    const int i0 = a.rope_traditional ? 2 * ii : ii, i1 = a.rope_traditional ? 2 * ii + 1 : ii + HD / 2;
                    const u16 o0 = T::from_f32(__fsub_rn(__fmul_rn(ra, cs), __fmul_rn(rb, sn)));
                    const u16 o1 = T::from_f32(__fadd_rn(__fmul_rn(ra, sn), __fmul_rn(rb, cs)));
                    u32 *gdst = (R < QD ? a.gq : a.gk) + head * HD;
                    coh_st4(gdst + i0, gran(tag, o0));
                    coh_st4(gdst + i1, gran(tag, o1));
                    if (R >= QD) {
                        typedef __attribute__((address_space(1))) u16 gu16w;
                        gu16w *kd = (gu16w *)(unsigned long long)(kdst + ((size_t)head * cap + pos) * HD);
                        kd[i0] = o0, kd[i1] = o1;
                    }
                } else {
                    const int rr = R - QD - KVD;
                    const int head = rr / HD, dd2 = rr % HD;
                    const u32 pk = pack2<T>(ra, rb);
                    coh_st4(a.gv + rr, gran(tag, (u16)(pk & 0xffffu)));
                    coh_st4(a.gv + rr + 1, gran(tag, (u16)(pk >> 16)));
                    *(gu32 *)(unsigned long long)(vdst + ((size_t)head * cap + pos) * HD + dd2) = pk;
                }
            }
        } else if (kind == K_OPROJ || kind == K_DOWN) {  // h = x + r (language.py:151,153): Linear output rounded, then the add rounded
            if (live) {
                res = pack2<T>(lo_f32<T>(res) + round_T<T>(va), hi_f32<T>(res) + round_T<T>(vb));
                coh_st4(a.gh + R, gran(tag, (u16)(res & 0xffffu)));
                coh_st4(a.gh + R + 1, gran(tag, (u16)(res >> 16)));
                if (kind == K_DOWN && li + 1 == a.n_layers) *(gu32 *)(unsigned long long)(a.h + R) = res;  // the bound hidden-state output
            }
        } else if (kind == K_GATEUP) {  // nn.silu(gate) * up (language.py:127)
            if (live) {
                const float gte = round_T<T>(va), up = round_T<T>(vb);
                const float slu = round_T<T>(gte / (1.0f + expf(-gte)));
                coh_st4(a.gact + pair, gran(tag, T::from_f32(slu * up)));
            }
        } else {  // logits + per-wave log-softmax partials (language.py:206-209); the tail hand-off is the flag barrier
            const float oa = round_T<T>(va), ob = round_T<T>(vb);
            if (live) coh_st4(a.logits + R, pack2<T>(oa, ob));
            const float mx = live ? fmaxf(oa, ob) : -INFINITY;
            const int ix = live ? (ob > oa ? R + 1 : R) : 0x7fffffff;
            const float wmax = wave_max(mx);
            int cand = (live && mx == wmax) ? ix : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
            float se = live ? expf(oa - wmax) + expf(ob - wmax) : 0.0f;
            se = wave_sum(se);
            if (lane == 0) {
                const __amdgpu_buffer_rsrc_t sr = coh_rsrc(a.stats, (unsigned)W * 16);
                coh_st16(sr, (unsigned)gw * 16, make_uint4(__builtin_bit_cast(u32, wmax), __builtin_bit_cast(u32, se), (u32)cand, 0u));
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the flag barrier announces these stores
        }
        stamp(3);
        if (kind == K_QKV) {
            // -------------------------------------------------------- attention (language.py:98-105, base.py:111-113)
            if (!attn_wg) {
                prefetch_next(Lw.wo, H, QD);
                MEGA_JOIN();  // (q staged -- only in the attention workgroups)
            } else {
                constexpr int LPT = HD / 8, TPW = 64 / LPT, NSUB = MEGA_CONSUMERS, NB = MEGA_KV_BLOCKS;
                const int ts = lane / LPT, dc = lane % LPT;
                const int Ttot = pos + 1;
                const int first = wave * TPW;
                const int n_blk = first < Ttot ? (Ttot - first + NSUB * TPW - 1) / (NSUB * TPW) : 0;
                // rows of earlier steps: plain loads, issued before the q hand-off
                const unsigned kv_bytes = (unsigned)((size_t)a.n_kv * cap * HD * 2);
                const __amdgpu_buffer_rsrc_t kr = coh_rsrc(reinterpret_cast<const void *>(const_load(a.kv_table + li)), kv_bytes);
                const __amdgpu_buffer_rsrc_t vr = coh_rsrc(reinterpret_cast<const void *>(const_load(a.kv_table + a.n_layers + li)), kv_bytes);
                const unsigned hoff = (unsigned)(((size_t)kvh * cap * HD + dc * 8) * 2);
                uint4 kq[NB], vq[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (b < n_blk) {
                        int t = first + b * NSUB * TPW + ts;
                        t = t < pos ? t : (pos > 0 ? pos - 1 : 0);
                        const u32x4_t k4 = __builtin_amdgcn_raw_buffer_load_b128(kr, hoff + (unsigned)t * (HD * 2), 0, 0);
                        const u32x4_t v4 = __builtin_amdgcn_raw_buffer_load_b128(vr, hoff + (unsigned)t * (HD * 2), 0, 0);
                        kq[b] = make_uint4(k4.x, k4.y, k4.z, k4.w), vq[b] = make_uint4(v4.x, v4.y, v4.z, v4.w);
                    }
                MEGA_JOIN();  // q heads and the new k / v row of this kv head are in LDS (IO waves)
                const u16 *s_q = reinterpret_cast<const u16 *>(s_att);              // [REP][HD]
                const u16 *s_kn = s_q + REP * HD, *s_vn = s_kn + HD;               // [HD] each
                float *s_m = reinterpret_cast<float *>(s_att + (REP + 2) * HD * 2);  // [REP][NSUB]
                float *s_l = s_m + REP * NSUB, *s_acc = s_l + REP * NSUB;           // [REP][NSUB], [REP][NSUB][HD]
                u32 qr[REP][4];
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    const uint4 qv = *reinterpret_cast<const uint4 *>(s_q + h * HD + dc * 8);
                    qr[h][0] = qv.x, qr[h][1] = qv.y, qr[h][2] = qv.z, qr[h][3] = qv.w;
                }
                const uint4 knew = *reinterpret_cast<const uint4 *>(s_kn + dc * 8), vnew = *reinterpret_cast<const uint4 *>(s_vn + dc * 8);
                float m[REP], l[REP], acc[REP][8];
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    m[h] = ATTN_NEG, l[h] = 0.0f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[h][j] = 0.0f;
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (b < n_blk) {  // wave-uniform
                        const int t = first + b * NSUB * TPW + ts;
                        const bool valid = t < Ttot;
                        const uint4 kk = t == pos ? knew : kq[b], vv = t == pos ? vnew : vq[b];
                        const u32 kw[4] = {kk.x, kk.y, kk.z, kk.w};
                        float vf[8];
                        vf[0] = lo_f32<T>(vv.x), vf[1] = hi_f32<T>(vv.x), vf[2] = lo_f32<T>(vv.y), vf[3] = hi_f32<T>(vv.y);
                        vf[4] = lo_f32<T>(vv.z), vf[5] = hi_f32<T>(vv.z), vf[6] = lo_f32<T>(vv.w), vf[7] = hi_f32<T>(vv.w);
                        float sc[REP];
#pragma unroll
                        for (int h = 0; h < REP; ++h) {
                            sc[h] = 0.0f;
#pragma unroll
                            for (int j = 0; j < 4; ++j) sc[h] = T::dot2(qr[h][j], kw[j], sc[h]);
                        }
#pragma unroll
                        for (int h = 0; h < REP; ++h) sc[h] += __builtin_amdgcn_update_dpp(0.0f, sc[h], 0xB1, 0xF, 0xF, true);
#pragma unroll
                        for (int h = 0; h < REP; ++h) sc[h] += __builtin_amdgcn_update_dpp(0.0f, sc[h], 0x4E, 0xF, 0xF, true);
#pragma unroll
                        for (int h = 0; h < REP; ++h) sc[h] += __builtin_amdgcn_update_dpp(0.0f, sc[h], 0x141, 0xF, 0xF, true);
                        if (LPT == 16) {
#pragma unroll
                            for (int h = 0; h < REP; ++h) sc[h] += __builtin_amdgcn_update_dpp(0.0f, sc[h], 0x140, 0xF, 0xF, true);
                        }
                        bool grow = false;
#pragma unroll
                        for (int h = 0; h < REP; ++h) {
                            sc[h] = valid ? sc[h] * sl2 : ATTN_NEG;
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
                    }
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    float mw = m[h];
                    if (LPT == 8) mw = fmaxf(mw, ror8(mw));
                    mw = xor32_max(xor16_max(mw));
                    const float wg = attn_exp2(m[h] - mw);
                    m[h] = mw;
                    l[h] *= wg;
                    if (LPT == 8) l[h] += ror8(l[h]);
                    l[h] = xor32_sum(xor16_sum(l[h]));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        acc[h][j] *= wg;
                        if (LPT == 8) acc[h][j] += ror8(acc[h][j]);
                        acc[h][j] = xor32_sum(xor16_sum(acc[h][j]));
                    }
                }
                if (ts == 0) {
#pragma unroll
                    for (int h = 0; h < REP; ++h) {
                        if (dc == 0) s_m[h * NSUB + wave] = m[h], s_l[h * NSUB + wave] = l[h];
#pragma unroll
                        for (int j = 0; j < 8; ++j) s_acc[(h * NSUB + wave) * HD + dc * 8 + j] = acc[h][j];
                    }
                }
                MEGA_BAR();  // (the IO waves of this workgroup join it)
                for (int o = tid; o < REP * HD; o += MEGA_NT) {
                    const int h = o / HD, d = o % HD;
                    float M = ATTN_NEG;
#pragma unroll
                    for (int i = 0; i < NSUB; ++i) M = fmaxf(M, s_m[h * NSUB + i]);
                    float Lsum = 0.0f, A = 0.0f;
#pragma unroll
                    for (int i = 0; i < NSUB; ++i) {
                        const float w = attn_exp2(s_m[h * NSUB + i] - M);
                        Lsum = fmaf(w, s_l[h * NSUB + i], Lsum);
                        A = fmaf(w, s_acc[(h * NSUB + i) * HD + d], A);
                    }
                    // one split: the merge of the launch path (attn_merge_finish) degenerates to w = exp2(0) = 1, A / Lsum
                    const float w1 = attn_exp2(M - M);
                    const float L1 = fmaf(w1, Lsum, 0.0f), A1 = fmaf(w1, A, 0.0f);
                    coh_st4(a.ga + (kvh * REP + h) * HD + d, gran(tag + 1, T::from_f32(A1 / L1)));
                }
                prefetch_next(Lw.wo, H, QD);
            }
            ++edge;  // the attention output's tag
        } else if (more) {
            if (nkind == K_QKV) prefetch_next(Ln.wqkv, N_qkv, H);
            else if (nkind == K_GATEUP) prefetch_next(Lw.wgateup, 2 * a.I, H);
            else if (nkind == K_DOWN) prefetch_next(Lw.wdown, H, a.I);
            else prefetch_next(a.lm_head, a.V, H);
        }
        if (!more) break;
        kind = nkind, li = nli, Lw = Ln;
        ++ph, ++prof_phase;
    }

    if (!a.with_logits) {  // a prompt token before the last: only the caches were filled
        if (blockIdx.x == 0 && tid == 0) a.state->pos = pos + 1;
        return;
    }
    // ================================================================== tail: log-softmax + greedy argmax (inference_engine.py:268-271), as k_logits_finish
    MEGA_BAR();  // every streaming wave has drained its logits / partial stores
    MEGA_JOIN();
    float *s_max = reinterpret_cast<float *>(smem), *s_sum = s_max + 4;
    int *s_arg = reinterpret_cast<int *>(s_sum + 4);
    const int n_stats = W;
    float M = 0.0f, part_se = 0.0f;
    int tok = 0;
    LogitStat st[16];
    if (tid < 256) {
        const __amdgpu_buffer_rsrc_t sr = coh_rsrc(a.stats, (unsigned)W * 16);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid + 256 * k;
            const uint4 raw = coh_ld16(sr, (unsigned)(i < n_stats ? i : n_stats - 1) * 16);
            const u32 rx = raw.x, ry = raw.y;
            st[k].max = i < n_stats ? coh_f32(rx) : -INFINITY;
            st[k].sumexp = i < n_stats ? coh_f32(ry) : 0.0f;
            st[k].argmax = i < n_stats ? (int)raw.z : 0x7fffffff;
        }
        float mx = -INFINITY;
        int arg = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (st[k].max > mx || (st[k].max == mx && st[k].argmax < arg)) mx = st[k].max, arg = st[k].argmax;
        const float wmax = wave_max(mx);
        int cand = (mx == wmax) ? arg : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        if (lane == 0) s_max[wave] = wmax, s_arg[wave] = cand;
    }
    MEGA_BAR();
    if (tid < 256) {
        M = s_max[0], tok = s_arg[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_max[w] > M || (s_max[w] == M && s_arg[w] < tok)) M = s_max[w], tok = s_arg[w];
#pragma unroll
        for (int k = 0; k < 16; ++k) part_se += st[k].sumexp > 0.0f ? st[k].sumexp * expf(st[k].max - M) : 0.0f;
        part_se = wave_sum(part_se);
        if (lane == 0) s_sum[wave] = part_se;
    }
    MEGA_BAR();
    if (!is_io) {
        M = s_max[0], tok = s_arg[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_max[w] > M || (s_max[w] == M && s_arg[w] < tok)) M = s_max[w], tok = s_arg[w];
        const float lse = M + logf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
        const int V = a.V;
        const int slice = (((V + G - 1) / G) + 7) & ~7;
        const int begin = blockIdx.x * slice, end = min(V, begin + slice);
        const __amdgpu_buffer_rsrc_t lr = coh_rsrc(a.logits, (unsigned)V * 2);
        for (int i = begin + tid * 8; i < end; i += MEGA_NT * 8) {
            if (i + 8 <= end && (V & 7) == 0) {
                const uint4 v = coh_ld16(lr, (unsigned)i * 2);
                const float4 o0 = make_float4(lo_f32<T>(v.x) - lse, hi_f32<T>(v.x) - lse, lo_f32<T>(v.y) - lse, hi_f32<T>(v.y) - lse);
                const float4 o1 = make_float4(lo_f32<T>(v.z) - lse, hi_f32<T>(v.z) - lse, lo_f32<T>(v.w) - lse, hi_f32<T>(v.w) - lse);
                *reinterpret_cast<float4 *>(a.logprobs + i) = o0;
                *reinterpret_cast<float4 *>(a.logprobs + i + 4) = o1;
            } else {
                for (int k = i; k < min(i + 8, end); ++k) {
                    const u32 w2 = coh_ld4(a.logits + (k & ~1));
                    a.logprobs[k] = ((k & 1) ? hi_f32<T>(w2) : lo_f32<T>(w2)) - lse;
                }
            }
        }
        if (blockIdx.x == 0 && tid == 0) {
            *a.token_out = tok;
            const int next_pos = pos + 1;
            if (a.history && next_pos < a.hist_cap) a.history[next_pos] = tok;
            a.state->token = tok;
            a.state->pos = next_pos;
        }
    }
}

template <class T, int D, int NPT>
int mega_launch_t(const MegaArgs &a, int hd, int rep, int grid, unsigned lds, hipStream_t st) {
#define PIE_MEGA_CASE(HD_, REP_)                                                                                              \
    if (hd == HD_ && rep == REP_) {                                                                                          \
        static bool attr_set = false;                                                                                        \
        if (!attr_set) {                                                                                                     \
            PIE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_step_mega<T, D, HD_, REP_, NPT>),              \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));                \
            attr_set = true;                                                                                                 \
        }                                                                                                                    \
        hipLaunchKernelGGL((k_step_mega<T, D, HD_, REP_, NPT>), dim3(grid), dim3(MEGA_THREADS), lds, st, a);                 \
        PIE_LAUNCH_CHECK();                                                                                                  \
        return PIE_OK;                                                                                                       \
    }
    PIE_MEGA_CASE(128, 4)
    PIE_MEGA_CASE(128, 8)
    PIE_MEGA_CASE(64, 4)
#undef PIE_MEGA_CASE
    return pie::fail(PIE_E_SHAPE, "mega step: head geometry not instantiated");
}

}  // namespace

struct MegaState {
    MegaLayer *layers_dev = nullptr;
    char *dev = nullptr;  // MegaSync followed by the granule vectors (one allocation, one memset node per step)
    size_t dev_bytes = 0;
    int n_cus = 0;
    bool enabled = true;
    unsigned long long *prof = nullptr;
};

void mega_invalidate(pie_decoder *d) {  // the layer table is rebuilt at the next step; options survive
    if (!d->mega) return;
    if (d->mega->layers_dev) (void)hipFree(d->mega->layers_dev);
    if (d->mega->dev) (void)hipFree(d->mega->dev);
    d->mega->layers_dev = nullptr, d->mega->dev = nullptr;
}

void mega_free(pie_decoder *d) {
    if (!d->mega) return;
    mega_invalidate(d);
    delete d->mega;
    d->mega = nullptr;
}

static MegaState *mega_state(pie_decoder *d) {
    if (d->mega) return d->mega;
    MegaState *m = new (std::nothrow) MegaState();
    if (!m) return nullptr;
    const char *e = getenv("PIE_STEP_MEGA");
    m->enabled = e && e[0] == '1';  // opt-in (env or pie_decoder_configure) until it is the faster path: DESIGN.md 3
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) m->enabled = false;
    else m->n_cus = p.multiProcessorCount > MEGA_MAX_WGS ? MEGA_MAX_WGS : p.multiProcessorCount;
    d->mega = m;
    return m;
}

void mega_enable(pie_decoder *d, bool on) {
    if (MegaState *m = mega_state(d)) m->enabled = on && m->n_cus > 0;
}

// Can this decoder's step run as the persistent launch?  (Everything else keeps the launch sequence of decoder.hip.)
bool mega_supported(pie_decoder *d, bool with_logits) {
    const pie_decoder_config &c = d->cfg;
    MegaState *ms = mega_state(d);
    if (!ms || !ms->enabled || !ms->layers_dev) return false;
    if (c.weight_format != PIE_W_INT4_G64 || d->block_table) return false;
    for (const pie_layer_weights &w : d->layers)
        if (w.bqkv || w.bo || w.bgateup || w.bdown) return false;
    const int rep = c.n_heads / c.n_kv_heads, QD = c.n_heads * c.head_dim;
    if (!((c.head_dim == 128 && (rep == 4 || rep == 8)) || (c.head_dim == 64 && rep == 4))) return false;
    // attention keeps a kv head's rows in one workgroup's registers: capacity <= 8 waves x 8 blocks x (64 / (head_dim / 8)) rows
    if (d->kv_cap > MEGA_CONSUMERS * MEGA_KV_BLOCKS * (64 / (c.head_dim / 8))) return false;
    if (c.kv_splits > 1) return false;
    const int G = ms->n_cus;
    if (G < 8 || c.n_kv_heads > G) return false;
    // the IO lanes sweep whole 2048-value slabs; the fused norm is written for hidden <= 8192
    if (c.hidden % 2048 || c.inter % 2048 || QD % 2048 || c.hidden > 8192 || QD > 8192 || c.inter > 32768) return false;
    const int W = G * MEGA_CONSUMERS;
    auto run_ok = [W](int N) { return (N / 2 + W - 1) / W <= GEMV_MAX_RUN; };
    if (!run_ok(QD + 2 * c.n_kv_heads * c.head_dim) || !run_ok(c.hidden) || !run_ok(2 * c.inter) || (with_logits && !run_ok(c.vocab))) return false;
    if (with_logits && (W > TAIL_MAX_STATS || d->n_stats < W)) return false;  // stats buffer: one entry per streaming wave
    return true;
}

static size_t mega_gran_words(const pie_decoder_config &c) {
    const size_t QD = (size_t)c.n_heads * c.head_dim, KVD = (size_t)c.n_kv_heads * c.head_dim;
    return (size_t)c.hidden + QD + 2 * KVD + QD + (size_t)c.inter;
}

// Device-side tables of the persistent step; allocates, so it runs OUTSIDE stream capture (pie_decoder_step calls it first).
int mega_prepare(pie_decoder *d) {
    MegaState *m = mega_state(d);
    if (!m || !m->enabled || m->layers_dev) return PIE_OK;
    const pie_decoder_config &c = d->cfg;
    std::vector<MegaLayer> h(c.n_layers);
    for (int i = 0; i < c.n_layers; ++i) {
        const pie_layer_weights &w = d->layers[i];
        h[i] = {(const char *)w.wqkv, (const char *)w.wo, (const char *)w.wgateup, (const char *)w.wdown, (const u16 *)w.attn_norm, (const u16 *)w.mlp_norm};
    }
    PIE_HIP_TRY(hipMalloc((void **)&m->layers_dev, sizeof(MegaLayer) * c.n_layers));
    PIE_HIP_TRY(hipMemcpy(m->layers_dev, h.data(), sizeof(MegaLayer) * c.n_layers, hipMemcpyHostToDevice));
    m->dev_bytes = (sizeof(MegaSync) + mega_gran_words(c) * 4 + 15) & ~(size_t)15;
    PIE_HIP_TRY(hipMalloc((void **)&m->dev, m->dev_bytes));
    PIE_HIP_TRY(hipMemset(m->dev, 0, m->dev_bytes));
#ifdef PIE_MEGA_PROF
    if (!m->prof) {
        PIE_HIP_TRY(hipMalloc((void **)&m->prof, (size_t)256 * 512 * 16 * 8));
        PIE_HIP_TRY(hipMemset(m->prof, 0, (size_t)256 * 512 * 16 * 8));
    }
#endif
    return PIE_OK;
}

int mega_step_enqueue(pie_decoder *d, const int *token_ptr, bool with_logits, u16 *logits_dst, hipStream_t st) {
    const pie_decoder_config &c = d->cfg;
    MegaState *m = d->mega;
    PIE_REQUIRE(m && m->layers_dev && m->dev, PIE_E_STATE, "mega step: mega_prepare() was not called");
    const int QD = c.n_heads * c.head_dim, KVD = c.n_kv_heads * c.head_dim;
    MegaArgs a = {};
    a.layers = m->layers_dev, a.n_layers = c.n_layers, a.H = c.hidden, a.I = c.inter, a.n_heads = c.n_heads, a.n_kv = c.n_kv_heads, a.hd = c.head_dim, a.V = c.vocab;
    a.eps = c.rms_eps, a.attn_scale = 1.0f / sqrtf((float)c.head_dim);
    a.embed_codes = d->glob.embed_codes, a.embed_scales = (const u16 *)d->glob.embed_scales, a.embed_biases = (const u16 *)d->glob.embed_biases;
    a.final_norm = (const u16 *)d->glob.final_norm, a.lm_head = (const char *)d->glob.lm_head, a.freqs = d->glob.rope_freqs;
    a.state = d->state, a.token_ptr = token_ptr, a.kv_table = d->kv_table;
    a.h = d->h, a.logits = logits_dst, a.logprobs = d->logprobs;
    a.stats = d->stats, a.token_out = d->token_out, a.history = d->history, a.hist_cap = d->hist_cap;
    a.with_logits = with_logits ? 1 : 0, a.rope_traditional = c.rope_traditional;
    a.sync = reinterpret_cast<MegaSync *>(m->dev);
    u32 *g = reinterpret_cast<u32 *>(m->dev + sizeof(MegaSync));
    a.gh = g, g += c.hidden;
    a.gq = g, g += QD;
    a.gk = g, g += KVD;
    a.gv = g, g += KVD;
    a.ga = g, g += QD;
    a.gact = g;
    a.prof = m->prof;
    {
        const char *e = getenv("PIE_MEGA_PACE");
        a.pace = e ? atoi(e) : 3;
    }
    {
        const char *e = getenv("PIE_MEGA_PROF_BLOCK");
        a.prof_block = e ? atoi(e) : 0;
    }
    int kmax = c.hidden > c.inter ? c.hidden : c.inter;
    kmax = kmax > QD ? kmax : QD;
    const int rep = c.n_heads / c.n_kv_heads;
    a.img_stride = ((unsigned)gemv_lds(kmax).off_out + 15u) & ~15u;  // x slots | group sums | reduction scratch, per image
    unsigned lds = 2 * a.img_stride;
    a.off_outp = lds, lds += MEGA_CONSUMERS * 2 * GEMV_MAX_RUN * 4;
    a.off_att = lds, lds += (unsigned)((rep + 2) * c.head_dim * 2 + rep * MEGA_CONSUMERS * (c.head_dim + 2) * 4);
    lds = (lds + 15u) & ~15u;
    a.off_rope = lds, lds += (unsigned)c.head_dim * 4;
    a.off_ctl = lds, lds += 16;
    lds = lds > MEGA_LDS_MIN ? lds : MEGA_LDS_MIN;  // more than half a CU's LDS: exactly one workgroup per CU, all co-resident
    PIE_REQUIRE(lds <= 160u * 1024u, PIE_E_SHAPE, "mega step: activation images do not fit LDS");
    PIE_HIP_TRY(hipMemsetAsync(m->dev, 0, m->dev_bytes, st));
    const int G = m->n_cus;
    const bool big = kmax > 4 * 4096;  // more than 4 activation pieces per staging thread
    if (c.dtype == PIE_BF16) return big ? mega_launch_t<BF16, 4, 8>(a, c.head_dim, rep, G, lds, st) : mega_launch_t<BF16, 4, 4>(a, c.head_dim, rep, G, lds, st);
    return big ? mega_launch_t<F16, 4, 8>(a, c.head_dim, rep, G, lds, st) : mega_launch_t<F16, 4, 4>(a, c.head_dim, rep, G, lds, st);
}

void *mega_prof_ptr(pie_decoder *d) { return d->mega ? (void *)d->mega->prof : nullptr; }

int mega_status(pie_decoder *d, unsigned *err) {
    *err = 0;
    if (!d->mega || !d->mega->dev) return PIE_OK;
    PIE_HIP_TRY(hipMemcpy(err, &reinterpret_cast<MegaSync *>(d->mega->dev)->error, sizeof(unsigned), hipMemcpyDeviceToHost));
    return PIE_OK;
}
