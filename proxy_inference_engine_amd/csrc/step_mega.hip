// step_mega.hip -- the whole decode step as ONE persistent launch.
//
// Same arithmetic as the launch sequence of decoder.hip (one Model.__call__ for inputs[1,1], models/llama/language.py:199-210,
// + the tail of _inference, engine/inference_engine.py:252-271), bit for bit: the same W4S row-pair -> wave assignment,
// the same fp32 orders in RMSNorm, the unit dot products, the split-KV attention, its merge and the log-softmax tail.
// What changes is the schedule:
//   * grid = one workgroup per CU, 8 streaming waves + 1 sync wave; the workgroups stay resident through all phases
//     (embed | per layer: qkv, attention, o_proj, gate/up, down | lm_head | tail);
//   * a phase boundary is a grid-wide hand-off: every producing wave drains its write-through (sc1) stores, the workgroup
//     meets at an s_barrier, the SYNC wave adds to a sharded arrival counter and polls it with sc1 loads (its vmcnt holds
//     nothing else, so a poll never queues behind weight loads), a second s_barrier releases the streaming waves, which
//     then gather the activation vector with sc1 loads (MI355X guide, "Valid forms", first table row);
//   * weights do not depend on activations: after its epilogue stores each streaming wave issues the first D units of the
//     NEXT phase's weight stream and waits only for the stores (s_waitcnt vmcnt(3 D), counters retire in order), so HBM
//     keeps streaming while the grid synchronises -- the part a kernel boundary cannot overlap.
// Every spin is bounded (s_memrealtime); a give-up sets MegaSync::error and the grid drains.
//
// STATUS (round 2): bit-identical to the launch sequence and SLOWER -- 1.92 vs 1.26 ms per step on the 8B model -- so it is
// opt-in (PIE_STEP_MEGA=1 / pie_decoder_configure).  The stream phases run at HBM speed, but one grid-wide hand-off on CUs that
// keep streaming costs 7.6 us (store ack ~2, workgroup rendezvous ~1, flag visible + poll ~2.7, gather + norm ~1.4) against
// ~3.9 for a kernel boundary incl. the next kernel's prologue; a second version with data-tagged 4-byte granules swept by
// dedicated IO waves (commit "Persistent step v2") removed the ack and the flag hop and measured no better (2.0 ms): every hop
// through memory costs 3-5 us while the CU's own weight loads are in flight.  DESIGN.md 3 has the timelines.
#include <vector>

#include "decoder.hpp"

namespace {

constexpr int MEGA_CONSUMERS = GEMV_WAVES;                 // streaming waves per workgroup
constexpr int MEGA_THREADS = (MEGA_CONSUMERS + 1) * 64;    // + the sync wave
constexpr int MEGA_NT = MEGA_CONSUMERS * 64;               // staging threads
constexpr int MEGA_NPT_NORM = 2;                           // pieces per thread of a NORMALISED input (K = hidden <= 8192)
constexpr unsigned long long MEGA_SPIN_LIMIT = 20000000ull;  // s_memrealtime ticks (100 MHz): 200 ms per grid barrier
constexpr unsigned MEGA_LDS_MIN = 84 * 1024;               // > half of the 160 KiB: one workgroup per CU, whatever else fits

// Grid barrier state (device memory, zeroed by a memset node before every launch).  No atomics and no shared counters: a
// read-modify-write on one line serialises at ~12 ns per arrival and 256 pollers on a handful of lines take microseconds per
// poll round (measured: 6-7 us per barrier with 8 sharded counters).  Instead every workgroup owns one 4-byte flag per REPLICA
// and stores the barrier number into all MEGA_REPLICAS copies with ONE wave instruction (lane r -> replica r); a workgroup polls
// replica (blockIdx % MEGA_REPLICAS) only, with one 16-byte load per lane (256 flags = 1 KiB): 8 pollers per line, no RMW.
constexpr int MEGA_REPLICAS = 32;
constexpr int MEGA_MAX_WGS = 256;  // one 16-byte load per lane covers 4 x 64 flags
struct MegaSync {
    unsigned flag[MEGA_REPLICAS][MEGA_MAX_WGS];  // flag[r][b] = number of the last barrier workgroup b has entered
    unsigned error;                              // first give-up code (0 = none)
    unsigned pad[31];
};

struct MegaLayer {
    const char *wqkv, *wo, *wgateup, *wdown;
    const u16 *attn_norm, *mlp_norm;
};

struct MegaArgs {
    const MegaLayer *layers;  // device array [n_layers]
    int n_layers, H, I, n_heads, n_kv, hd, V;
    float eps;
    const u32 *embed_codes;
    const u16 *embed_scales, *embed_biases, *final_norm;
    const char *lm_head;
    const float *freqs;
    DecState *state;
    const int *token_ptr;
    const unsigned long long *kv_table;
    u16 *h, *qbuf, *act, *logits;
    float *part_acc, *part_ml, *logprobs;
    LogitStat *stats;
    int *token_out, *history;
    int hist_cap, with_logits, splits, rope_traditional;
    MegaSync *sync;
    unsigned lds_rope, lds_ctl;  // byte offsets of the RoPE table and the control words in dynamic LDS
    unsigned long long *prof;    // developer build (-DPIE_MEGA_PROF): s_memrealtime stamps of one workgroup, [phase][16]
    int prof_block;
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

// D: ring depth = units prefetched across a hand-off.  NPT: activation pieces (8 elements) per staging thread, K <= NPT * 4096.
template <class T, int D, int HD, int REP, int NPT>
__global__ void __launch_bounds__(MEGA_THREADS) k_step_mega(const MegaArgs a) {
    constexpr int UB = W4S_UNIT_BYTES;
    constexpr int NT = MEGA_NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_sync = wave == MEGA_CONSUMERS;
    const int G = gridDim.x, W = G * MEGA_CONSUMERS;
    const int gw = blockIdx.x * MEGA_CONSUMERS + wave;
    const int tid = threadIdx.x;
    float *s_rope = reinterpret_cast<float *>(smem + a.lds_rope);
    int *s_ctl = reinterpret_cast<int *>(smem + a.lds_ctl);  // every access sits next to a MEGA_BAR (memory clobber): no volatile, which would turn it into flat accesses

    const int H = a.H, QD = a.n_heads * HD, KVD = a.n_kv * HD;
    // written by earlier launches: plain loads; pinned to SGPRs so that everything derived from them stays wave-uniform
    const int pos = __builtin_amdgcn_readfirstlane(a.state->pos), cap = __builtin_amdgcn_readfirstlane(a.state->cap);
    int token = __builtin_amdgcn_readfirstlane(*a.token_ptr);
    token = token < 0 ? 0 : (token >= a.V ? a.V - 1 : token);

    // ------------------------------------------------------------------ the grid-wide hand-off
    int prof_phase = 0;
    auto stamp = [&](int slot) {
#ifdef PIE_MEGA_PROF
        if (a.prof && (int)blockIdx.x == a.prof_block && lane == 0 && (wave == 0 || is_sync) && prof_phase < 512)
            a.prof[prof_phase * 16 + slot + (is_sync ? 8 : 0)] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    unsigned epoch = 0;  // sync wave: barriers entered so far
    auto grid_sync = [&](bool wait) -> bool {
        MEGA_BAR();  // every streaming wave has drained its stores (s_waitcnt vmcnt before this call)
        stamp(4);
        if (is_sync) {
            const unsigned k = ++epoch;
            if (lane < MEGA_REPLICAS) coh_st4(&a.sync->flag[lane][blockIdx.x], k);
            if (wait) {
                const __amdgpu_buffer_rsrc_t fr = coh_rsrc(&a.sync->flag[blockIdx.x % MEGA_REPLICAS][0], MEGA_MAX_WGS * 4);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                int dead = 0;
                for (;;) {
                    const uint4 f = coh_ld16(fr, lane * 16);
                    const int b0 = lane * 4;
                    const bool ok = (b0 >= G || f.x >= k) && (b0 + 1 >= G || f.y >= k) && (b0 + 2 >= G || f.z >= k) && (b0 + 3 >= G || f.w >= k);
                    if (__all(ok)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > MEGA_SPIN_LIMIT || __any(coh_ld4(&a.sync->error) != 0)) {
                        dead = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (dead && lane == 0) __hip_atomic_store(&a.sync->error, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) s_ctl[0] = dead;
            }
        }
        stamp(5);
        MEGA_BAR();
        stamp(6);
        return __builtin_amdgcn_readfirstlane(s_ctl[0]) == 0;  // uniform by construction; say so, or every loop-carried descriptor turns divergent
    };
#define MEGA_SYNC(wait)             \
    do {                            \
        if (!grid_sync(wait)) return; \
    } while (0)

    // ------------------------------------------------------------------ the weight stream (streaming waves)
    uint4 c0[D], c1[D];
    u32 sb[D];
    // issue side: one matrix at a time, units in this wave's order
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
        i_run = (!is_sync && w && gw < n_pairs) ? (n_pairs - gw + W - 1) / W : 0;
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
    auto prefetch_next = [&](const char *w, int N, int K) {  // after the epilogue stores: D units of the next phase, wait for the stores only
        asm volatile("" ::: "memory");
        stream_open(w, N, K);
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
        // the builtin, not inline asm: hipcc's waitcnt pass must SEE that the stores have retired -- with stores and loads
        // both pending it treats vmcnt as out of order and drains the ring (vmcnt(0)) at the head of the streaming loop
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * D) & 15) | (((3 * D) >> 4) << 14));
        asm volatile("" ::: "memory");
    };

    // ------------------------------------------------------------------ one GEMV phase
    // x pieces arrive in xv[] (8 elements each, piece j = tid + i * NT); PRO_RMSNORM normalises them with nv[].
    uint4 xv[NPT], nv[MEGA_NPT_NORM];
    auto gather_x = [&](const u16 *x, int K) {  // sc1 loads of a vector another workgroup wrote in this launch
        const int n_pieces = K >> 3;
        const __amdgpu_buffer_rsrc_t r = coh_rsrc(x, (unsigned)K * 2);
#pragma unroll
        for (int i = 0; i < NPT; ++i)
            if (i * NT < n_pieces) {  // uniform
                const int j = tid + i * NT;
                xv[i] = coh_ld16(r, (unsigned)(j < n_pieces ? j : n_pieces - 1) * 16);
            }
    };
    auto load_norm_w = [&](const u16 *w, int K) {  // ordinary weights: loaded BEFORE the hand-off
        const int n_pieces = K >> 3;
#pragma unroll
        for (int i = 0; i < MEGA_NPT_NORM; ++i)
            if (i * NT < n_pieces) {
                const int j = tid + i * NT;
                typedef __attribute__((address_space(1))) const u32x4_t gv4;  // the pointer comes from the layer table: say it is global, not flat
                const u32x4_t v = ((gv4 *)(unsigned long long)w)[j < n_pieces ? j : n_pieces - 1];
                nv[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
    };

    // PRO: PRO_NONE / PRO_RMSNORM on xv[]; PRO_ATTN builds xv[] from the split-KV partials.  The row sums of this wave's
    // pairs are left in LDS (outp); returns after the stream, before the epilogue.
    float *outp = nullptr;
    int c_run = 0;
    auto gemv_body = [&](int pro, int N, int K) {
        const GemvLds L = gemv_lds(K);
        float *sxs = reinterpret_cast<float *>(smem + L.off_sx);
        float *red = reinterpret_cast<float *>(smem + L.off_red);
        outp = reinterpret_cast<float *>(smem + L.off_out) + (is_sync ? 0 : wave) * (2 * GEMV_MAX_RUN);
        const int n_pieces = K >> 3, n_groups = K >> 6, ns = w4s_slices(K), n_pairs = N >> 1;
        c_run = (!is_sync && gw < n_pairs) ? (n_pairs - gw + W - 1) / W : 0;
        const int n_units = c_run * ns;
        if (pro == PRO_RMSNORM) {
            if (!is_sync) {
                float ssq = 0.0f;
#pragma unroll
                for (int i = 0; i < MEGA_NPT_NORM; ++i)
                    if (i * NT < n_pieces) {
                        const bool ok = tid + i * NT < n_pieces;
                        const u32 v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
                        float q = 0.0f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float lo = lo_f32<T>(v[k]), hi = hi_f32<T>(v[k]);
                            q = fmaf(lo, lo, q);
                            q = fmaf(hi, hi, q);
                        }
                        ssq += ok ? q : 0.0f;
                    }
                ssq = half_wave_sum(ssq);
                ssq = lane_value(ssq, 31) + lane_value(ssq, 63);
                if (lane == 0) red[wave] = ssq;
            }
            MEGA_BAR();
            if (!is_sync) {
                const float4 ra = *reinterpret_cast<const float4 *>(red), rb = *reinterpret_cast<const float4 *>(red + 4);
                const float tot = ((ra.x + ra.y) + (ra.z + ra.w)) + ((rb.x + rb.y) + (rb.z + rb.w));
                const float inv = 1.0f / sqrtf(tot / (float)K + a.eps);
#pragma unroll
                for (int i = 0; i < MEGA_NPT_NORM; ++i)
                    if (i * NT < n_pieces) {
                        const u32 v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, g[4] = {nv[i].x, nv[i].y, nv[i].z, nv[i].w};
                        u32 o[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            o[k] = pack2<T>(round_T<T>(lo_f32<T>(v[k]) * inv) * lo_f32<T>(g[k]), round_T<T>(hi_f32<T>(v[k]) * inv) * hi_f32<T>(g[k]));
                        xv[i] = make_uint4(o[0], o[1], o[2], o[3]);
                    }
            }
        }
        if (!is_sync) {
#pragma unroll
            for (int i = 0; i < NPT; ++i)
                if (i * NT < n_pieces) {
                    const int j = tid + i * NT;
                    const bool ok = j < n_pieces;
                    float ps = ok ? sum8<T>(xv[i]) : 0.0f;
                    ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0xB1, 0xF, 0xF, true);
                    ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0x4E, 0xF, 0xF, true);
                    ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0x141, 0xF, 0xF, true);
                    if (ok) {
                        *reinterpret_cast<uint4 *>(smem + ((size_t)(j & 7) * L.stride + (j >> 3)) * 16) = scale8<T>(xv[i]);
                        if ((j & 7) == 0) sxs[j >> 3] = ps;
                    }
                }
        }
        MEGA_BAR();
        stamp(1);
        if (is_sync) return;
        float acc = 0.0f;
        int sl = 0, pl = 0;
        // The prefetched units were issued before the hand-off and have long landed; retiring everything HERE leaves only
        // weight loads pending inside the loop, so hipcc emits counted vmcnt waits there instead of vmcnt(0) at the loop head
        // (any store or scratch access still pending at the head makes it treat vmcnt as out of order).
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
                        const uint4 v = *reinterpret_cast<const uint4 *>(smem + ((size_t)r * L.stride + gc) * 16);
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

    // ------------------------------------------------------------------ setup: RoPE table, embedding, first prefetch
    if (!is_sync && tid < HD / 2) {  // cos / sin of pos / freqs[i] (llama/utils.py:42-50), as k_embedding_w4g64 computes them
        const float theta = (float)pos * (1.0f / a.freqs[tid]);
        float sn, cs;
        sincosf(theta, &sn, &cs);
        s_rope[2 * tid] = cs, s_rope[2 * tid + 1] = sn;
    }
    if (lane == 0 && is_sync) s_ctl[0] = 0;
    {   // h = embed_tokens(token) (language.py:176): every workgroup dequantises the row itself; workgroup 0 also stores it (the residual stream)
        const int words = H >> 3;
        const u32 *row = a.embed_codes + (size_t)token * words;
        const u16 *srow = a.embed_scales + (size_t)token * (H >> 6), *brow = a.embed_biases + (size_t)token * (H >> 6);
        const __amdgpu_buffer_rsrc_t hr = coh_rsrc(a.h, (unsigned)H * 2);
        if (!is_sync) {
#pragma unroll
            for (int i = 0; i < MEGA_NPT_NORM; ++i)
                if (i * NT < words) {
                    const int j = tid + i * NT, jc = j < words ? j : words - 1;
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
                    if (blockIdx.x == 0 && j < words) coh_st16(hr, (unsigned)j * 16, xv[i]);
                }
        }
    }
    const int N_qkv = QD + 2 * KVD;
    if (!is_sync) {
        const MegaLayer L0 = load_layer(a.layers);
        load_norm_w(L0.attn_norm, H);
        stream_open(L0.wqkv, N_qkv, H);
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
    }

    // attention geometry (host plan: splits <= GEMV_ATTN_SPLITS, merged by the o_proj prologue)
    const int n_attn_wg = a.n_kv * a.splits;
    const bool attn_wg = (int)blockIdx.x < n_attn_wg;
    const float sl2 = (1.0f / sqrtf((float)HD)) * ATTN_LOG2E;

    // The phases run through ONE copy of the streaming loop (a state machine over the phase kind): five inlined copies made
    // the compiler hoist five sets of loop invariants across the whole step and spill inside the hand-offs.
    enum { K_QKV = 0, K_OPROJ = 1, K_GATEUP = 2, K_DOWN = 3, K_LMHEAD = 4 };
    int kind = K_QKV, li = 0;
    MegaLayer Lw = load_layer(a.layers);
    u32 pre_u = 0;
    const int run_h = (!is_sync && gw < (H >> 1)) ? ((H >> 1) - gw + W - 1) / W : 0;  // this wave's row pairs of an [H, *] matrix
    for (;;) {
        // ---------------------------------------------------------------- prologue: the phase's input vector
        stamp(0);
        int N = N_qkv, K = H, pro = PRO_RMSNORM;
        if (kind == K_QKV) {  // input_layernorm + q|k|v + RoPE + cache append (language.py:83-95); layer 0 holds the embedding in xv
            if (li > 0 && !is_sync) gather_x(a.h, H);
        } else if (kind == K_OPROJ) {  // split merge + o_proj + residual (language.py:107-108,151)
            N = H, K = QD, pro = PRO_NONE;
            if (!is_sync) {
                const int n_pieces = QD >> 3, ppd = HD >> 3;
                const int active = attn_split(pos + 1, a.splits).active;
                const __amdgpu_buffer_rsrc_t pa = coh_rsrc(a.part_acc, (unsigned)((size_t)a.n_heads * a.splits * HD * 4));
                const __amdgpu_buffer_rsrc_t pm = coh_rsrc(a.part_ml, (unsigned)((size_t)a.n_heads * a.splits * 2 * 4));
#pragma unroll
                for (int i = 0; i < 2; ++i)  // QD <= 8192 (host-checked)
                    if (i * NT < n_pieces) {
                        int j = tid + i * NT;
                        j = j < n_pieces ? j : n_pieces - 1;
                        const int hh = j / ppd, d0 = (j % ppd) * 8;
                        AttnMergeRegs<GEMV_ATTN_SPLITS> mr;
#pragma unroll
                        for (int s = 0; s < GEMV_ATTN_SPLITS; ++s) {
                            const int jc = s < active ? s : active - 1;
                            const float2 ml = coh_ld8f(pm, (unsigned)((hh * a.splits + jc) * 2) * 4);
                            mr.mj[s] = ml.x, mr.lj[s] = ml.y;
                            const unsigned o = (unsigned)((hh * a.splits + jc) * HD + d0) * 4;
                            mr.a0[s] = coh_ld16f(pa, o), mr.a1[s] = coh_ld16f(pa, o + 16);
                        }
                        float o8[8];
                        attn_merge_finish<GEMV_ATTN_SPLITS>(mr, active, o8);
                        xv[i] = make_uint4(pack2<T>(o8[0], o8[1]), pack2<T>(o8[2], o8[3]), pack2<T>(o8[4], o8[5]), pack2<T>(o8[6], o8[7]));
                    }
                if (lane < run_h) pre_u = coh_ld4(a.h + 2 * (gw + lane * W));
            }
        } else if (kind == K_GATEUP) {  // post_attention_layernorm + gate|up + SwiGLU (language.py:127,152)
            N = 2 * a.I;
            if (!is_sync) gather_x(a.h, H);
        } else if (kind == K_DOWN) {  // down_proj + residual (language.py:127,153)
            N = H, K = a.I, pro = PRO_NONE;
            if (!is_sync) {
                gather_x(a.act, a.I);
                if (lane < run_h) pre_u = coh_ld4(a.h + 2 * (gw + lane * W));
            }
        } else {  // final norm + lm_head + per-wave log-softmax partials (language.py:187,206-209)
            N = a.V;
            if (!is_sync) gather_x(a.h, H);
        }
        gemv_body(pro, N, K);
        stamp(2);
        // ---------------------------------------------------------------- epilogue: one lane per row pair
        if (!is_sync) {
            const bool live = lane < c_run;
            const int pair = gw + lane * W, R = 2 * pair;
            float va = 0.0f, vb = 0.0f;
            if (live) {
                const float2 o = *reinterpret_cast<const float2 *>(outp + 2 * lane);
                va = o.x, vb = o.y;
            }
            if (kind == K_QKV) {
                if (live) {
                    const float ra = round_T<T>(va), rb = round_T<T>(vb);
                    u16 *kdst = reinterpret_cast<u16 *>(const_load(a.kv_table + li));
                    u16 *vdst = reinterpret_cast<u16 *>(const_load(a.kv_table + a.n_layers + li));
                    if (R < QD + KVD) {
                        const int rr = R < QD ? R : R - QD;
                        const int head = rr / HD, ii = (rr % HD) >> 1;
                        const float cs = s_rope[2 * ii], sn = s_rope[2 * ii + 1];
                        u16 *dst = R < QD ? a.qbuf + (size_t)head * HD : kdst + ((size_t)head * cap + pos) * HD;
                        const int i0 = a.rope_traditional ? 2 * ii : ii, i1 = a.rope_traditional ? 2 * ii + 1 : ii + HD / 2;
                        coh_st2(dst + i0, T::from_f32(__fsub_rn(__fmul_rn(ra, cs), __fmul_rn(rb, sn))));
                        coh_st2(dst + i1, T::from_f32(__fadd_rn(__fmul_rn(ra, sn), __fmul_rn(rb, cs))));
                    } else {
                        const int rr = R - QD - KVD;
                        const int head = rr / HD, dd2 = rr % HD;
                        coh_st4(vdst + ((size_t)head * cap + pos) * HD + dd2, pack2<T>(ra, rb));
                    }
                }
            } else if (kind == K_OPROJ || kind == K_DOWN) {  // h = x + r: Linear output rounded to T, then the add rounded to T
                if (live) coh_st4(a.h + R, pack2<T>(lo_f32<T>(pre_u) + round_T<T>(va), hi_f32<T>(pre_u) + round_T<T>(vb)));
            } else if (kind == K_GATEUP) {
                if (live) {
                    const float gte = round_T<T>(va), up = round_T<T>(vb);
                    const float slu = round_T<T>(gte / (1.0f + expf(-gte)));
                    coh_st2(a.act + pair, T::from_f32(slu * up));
                }
            } else {
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
            }
        }
        // ---------------------------------------------------------------- what streams next (weights do not wait for the hand-off)
        const int prev_kind = kind;
        bool more = true;
        if (kind == K_DOWN) {
            if (li + 1 < a.n_layers) ++li, kind = K_QKV, Lw = load_layer(a.layers + li);
            else if (a.with_logits) kind = K_LMHEAD;
            else more = false;
        } else if (kind == K_LMHEAD) {
            more = false;
        } else {
            ++kind;
        }
        if (!is_sync) {
            if (!more) {
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            } else if (kind == K_QKV) {
                load_norm_w(Lw.attn_norm, H);
                prefetch_next(Lw.wqkv, N_qkv, H);
            } else if (kind == K_OPROJ) {
                prefetch_next(Lw.wo, H, QD);
            } else if (kind == K_GATEUP) {
                load_norm_w(Lw.mlp_norm, H);
                prefetch_next(Lw.wgateup, 2 * a.I, H);
            } else if (kind == K_DOWN) {
                prefetch_next(Lw.wdown, H, a.I);
            } else {
                load_norm_w(a.final_norm, H);
                prefetch_next(a.lm_head, a.V, H);
            }
        }
        stamp(3);
        if (!more) break;
        // ---------------------------------------------------------------- the hand-off (after q|k|v: attention in between)
        if (prev_kind != K_QKV) {
            MEGA_SYNC(true);
            ++prof_phase;
            continue;
        }
        MEGA_SYNC(attn_wg);  // workgroups without an attention role only announce their q|k|v rows
        if (attn_wg && !is_sync) {
            constexpr int LPT = HD / 8, TPW = 64 / LPT, NSUB = attn_short_waves(REP), DA = 2;  // short caches only: two row blocks in flight are enough
            // the same wave count and stream layout as k_attn_decode's short-cache form (attention.hpp: attn_short_waves, WIDE), so the two paths
            // stay bit-identical: waves NSUB.. of the workgroup sit the scoring out
            constexpr bool WIDE = PIE_ATTN_WIDE && (size_t)REP * NSUB * TPW * HD * 4 <= 65536;
            constexpr int NSTR = WIDE ? NSUB * TPW : NSUB;
            float *s_m = reinterpret_cast<float *>(smem);              // [REP][NSTR]
            float *s_l = s_m + REP * NSTR;                             // [REP][NSTR]
            float *s_acc = s_l + REP * NSTR;                           // [REP][NSTR][HD]
            const int g = blockIdx.x % a.n_kv, split = blockIdx.x / a.n_kv;
            const int ts = lane / LPT, dc = lane % LPT;
            const int Ttot = pos + 1;
            const AttnSplit sp = attn_split(Ttot, a.splits);
            if (split < sp.active && wave < NSUB) {  // uniform per wave
                const int t_begin = split * sp.chunk, t_end = min(Ttot, t_begin + sp.chunk);
                const unsigned kv_bytes = (unsigned)((size_t)a.n_kv * cap * HD * 2);
                const __amdgpu_buffer_rsrc_t kr = coh_rsrc(reinterpret_cast<const void *>(const_load(a.kv_table + li)), kv_bytes);
                const __amdgpu_buffer_rsrc_t vr = coh_rsrc(reinterpret_cast<const void *>(const_load(a.kv_table + a.n_layers + li)), kv_bytes);
                const unsigned hoff = (unsigned)(((size_t)g * cap * HD + dc * 8) * 2);
                const int first = t_begin + wave * TPW;
                const int n_blk = first < t_end ? (t_end - first + NSUB * TPW - 1) / (NSUB * TPW) : 0;
                uint4 kq[DA], vq[DA];
                auto aissue = [&](int d, int b) {
                    int t = first + b * NSUB * TPW + ts;
                    t = t < t_end ? t : t_end - 1;
                    kq[d] = coh_ld16(kr, hoff + (unsigned)t * (HD * 2));
                    vq[d] = coh_ld16(vr, hoff + (unsigned)t * (HD * 2));
                };
#pragma unroll
                for (int d = 0; d < DA; ++d) aissue(d, d);
                u32 qr[REP][4];
                const __amdgpu_buffer_rsrc_t qrs = coh_rsrc(a.qbuf, (unsigned)QD * 2);
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    const uint4 qv = coh_ld16(qrs, (unsigned)(((g * REP + h) * HD + dc * 8) * 2));
                    qr[h][0] = qv.x, qr[h][1] = qv.y, qr[h][2] = qv.z, qr[h][3] = qv.w;
                }
                float m[REP], l[REP], acc[REP][8];
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    m[h] = ATTN_NEG, l[h] = 0.0f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[h][j] = 0.0f;
                }
                for (int base = 0; base < n_blk; base += DA) {
#pragma unroll
                    for (int d = 0; d < DA; ++d) {
                        const int b = base + d;
                        if (b < n_blk) {
                            const bool valid = first + b * NSUB * TPW + ts < t_end;
                            const u32 kw[4] = {kq[d].x, kq[d].y, kq[d].z, kq[d].w};
                            float vf[8];
                            vf[0] = lo_f32<T>(vq[d].x), vf[1] = hi_f32<T>(vq[d].x), vf[2] = lo_f32<T>(vq[d].y), vf[3] = hi_f32<T>(vq[d].y);
                            vf[4] = lo_f32<T>(vq[d].z), vf[5] = hi_f32<T>(vq[d].z), vf[6] = lo_f32<T>(vq[d].w), vf[7] = hi_f32<T>(vq[d].w);
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
                        aissue(d, b + DA);
                    }
                }
                if constexpr (WIDE) {
                    const int str = wave * TPW + ts;
#pragma unroll
                    for (int h = 0; h < REP; ++h) {
                        if (dc == 0) s_m[h * NSTR + str] = m[h], s_l[h * NSTR + str] = l[h];
                        *reinterpret_cast<float4 *>(&s_acc[(h * NSTR + str) * HD + dc * 8]) = make_float4(acc[h][0], acc[h][1], acc[h][2], acc[h][3]);
                        *reinterpret_cast<float4 *>(&s_acc[(h * NSTR + str) * HD + dc * 8 + 4]) = make_float4(acc[h][4], acc[h][5], acc[h][6], acc[h][7]);
                    }
                } else {
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
                        if (dc == 0) s_m[h * NSTR + wave] = m[h], s_l[h * NSTR + wave] = l[h];
#pragma unroll
                        for (int j = 0; j < 8; ++j) s_acc[(h * NSTR + wave) * HD + dc * 8 + j] = acc[h][j];
                    }
                }
                }
            }
        }
        if (attn_wg) {
            MEGA_BAR();
            if (!is_sync) {
                constexpr int TPW2 = 64 / (HD / 8);
                constexpr bool WIDE = PIE_ATTN_WIDE && (size_t)REP * attn_short_waves(REP) * TPW2 * HD * 4 <= 65536;
                constexpr int NSUB = WIDE ? attn_short_waves(REP) * TPW2 : attn_short_waves(REP);  // streams in LDS
                const float *s_m = reinterpret_cast<const float *>(smem), *s_l = s_m + REP * NSUB, *s_acc = s_l + REP * NSUB;
                const int g = blockIdx.x % a.n_kv, split = blockIdx.x / a.n_kv;
                if (split < attn_split(pos + 1, a.splits).active) {
                    for (int o = tid; o < REP * (HD / 2); o += NT) {  // (head, dim pair) per thread, as k_attn_decode's final pass
                        const int h = o / (HD / 2), d = (o % (HD / 2)) * 2;
                        float M = ATTN_NEG;
#pragma unroll
                        for (int i = 0; i < NSUB; ++i) M = fmaxf(M, s_m[h * NSUB + i]);
                        float Lsum = 0.0f, A0 = 0.0f, A1 = 0.0f;
#pragma unroll
                        for (int i = 0; i < NSUB; ++i) {
                            const float w = attn_exp2(s_m[h * NSUB + i] - M);
                            const float2 av = *reinterpret_cast<const float2 *>(&s_acc[(h * NSUB + i) * HD + d]);
                            Lsum = fmaf(w, s_l[h * NSUB + i], Lsum);
                            A0 = fmaf(w, av.x, A0), A1 = fmaf(w, av.y, A1);
                        }
                        const size_t hq = (size_t)g * REP + h;
                        coh_stf(a.part_acc + (hq * a.splits + split) * HD + d, A0);
                        coh_stf(a.part_acc + (hq * a.splits + split) * HD + d + 1, A1);
                        if (d == 0) {
                            coh_stf(a.part_ml + (hq * a.splits + split) * 2 + 0, M);
                            coh_stf(a.part_ml + (hq * a.splits + split) * 2 + 1, Lsum);
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
        }
        stamp(7);
        MEGA_SYNC(true);
        ++prof_phase;
    }

    if (!a.with_logits) {  // a prompt token before the last: only the caches were filled
        if (blockIdx.x == 0 && tid == 0) a.state->pos = pos + 1;
        return;
    }
    // ================================================================ tail: log-softmax + greedy argmax (inference_engine.py:268-271), as k_logits_finish
    MEGA_SYNC(true);
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
    if (!is_sync) {
        M = s_max[0], tok = s_arg[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_max[w] > M || (s_max[w] == M && s_arg[w] < tok)) M = s_max[w], tok = s_arg[w];
        const float lse = M + logf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
        const int V = a.V;
        const int slice = (((V + G - 1) / G) + 7) & ~7;
        const int begin = blockIdx.x * slice, end = min(V, begin + slice);
        const __amdgpu_buffer_rsrc_t lr = coh_rsrc(a.logits, (unsigned)V * 2);
        for (int i = begin + tid * 8; i < end; i += NT * 8) {
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
    MegaSync *sync = nullptr;
    int n_cus = 0;
    bool enabled = true;
    unsigned long long *prof = nullptr;
};

void mega_invalidate(pie_decoder *d) {  // the layer table is rebuilt at the next step; options survive
    if (!d->mega) return;
    if (d->mega->layers_dev) (void)hipFree(d->mega->layers_dev);
    if (d->mega->sync) (void)hipFree(d->mega->sync);
    d->mega->layers_dev = nullptr, d->mega->sync = nullptr;
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
    m->enabled = e && e[0] == '1';  // opt-in while the launch sequence is faster (measured 1.92 vs 1.26 ms per step)
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) m->enabled = false;
    else m->n_cus = p.multiProcessorCount;
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
    if (!d->uniform_int4() || d->block_table || d->combine || d->splits > GEMV_ATTN_SPLITS || d->tp()) return false;
    for (const pie_layer_weights &w : d->layers)
        if (w.bqkv || w.bo || w.bgateup || w.bdown) return false;
    const int rep = c.n_heads / c.n_kv_heads, QD = c.n_heads * c.head_dim;
    if (!((c.head_dim == 128 && (rep == 4 || rep == 8)) || (c.head_dim == 64 && rep == 4))) return false;
    const int G = d->mega->n_cus;
    if (G < 8 || G > MEGA_MAX_WGS || c.n_kv_heads * d->splits > G) return false;
    if (QD > 8192 || c.hidden > 8192 || c.inter > 32768) return false;
    const int W = G * MEGA_CONSUMERS;
    auto run_ok = [W](int N) { return (N / 2 + W - 1) / W <= GEMV_MAX_RUN; };
    if (!run_ok(QD + 2 * c.n_kv_heads * c.head_dim) || !run_ok(c.hidden) || !run_ok(2 * c.inter) || (with_logits && !run_ok(c.vocab))) return false;
    if (with_logits && W > TAIL_MAX_STATS) return false;
    if (with_logits && d->n_stats < W) return false;  // stats buffer holds one entry per streaming wave
    return true;
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
    PIE_HIP_TRY(hipMalloc((void **)&m->sync, sizeof(MegaSync)));
    PIE_HIP_TRY(hipMemset(m->sync, 0, sizeof(MegaSync)));
#ifdef PIE_MEGA_PROF
    if (!m->prof) {
        PIE_HIP_TRY(hipMalloc((void **)&m->prof, 512 * 16 * 8));
        PIE_HIP_TRY(hipMemset(m->prof, 0, 512 * 16 * 8));
    }
#endif
    return PIE_OK;
}

int mega_step_enqueue(pie_decoder *d, const int *token_ptr, bool with_logits, u16 *logits_dst, hipStream_t st) {
    const pie_decoder_config &c = d->cfg;
    MegaState *m = d->mega;
    PIE_REQUIRE(m && m->layers_dev && m->sync, PIE_E_STATE, "mega step: mega_prepare() was not called");
    MegaArgs a = {};
    a.layers = m->layers_dev, a.n_layers = c.n_layers, a.H = c.hidden, a.I = c.inter, a.n_heads = c.n_heads, a.n_kv = c.n_kv_heads, a.hd = c.head_dim, a.V = c.vocab;
    a.eps = c.rms_eps;
    a.embed_codes = d->glob.embed_codes, a.embed_scales = (const u16 *)d->glob.embed_scales, a.embed_biases = (const u16 *)d->glob.embed_biases;
    a.final_norm = (const u16 *)d->glob.final_norm, a.lm_head = (const char *)d->glob.lm_head, a.freqs = d->glob.rope_freqs;
    a.state = d->state, a.token_ptr = token_ptr, a.kv_table = d->kv_table;
    a.h = d->h, a.qbuf = d->qbuf, a.act = d->act, a.logits = logits_dst, a.part_acc = d->part_acc, a.part_ml = d->part_ml, a.logprobs = d->logprobs;
    a.stats = d->stats, a.token_out = d->token_out, a.history = d->history, a.hist_cap = d->hist_cap;
    a.with_logits = with_logits ? 1 : 0, a.splits = d->splits, a.rope_traditional = c.rope_traditional;
    a.sync = m->sync;
    a.prof = m->prof;
    {
        const char *e = getenv("PIE_MEGA_PROF_BLOCK");
        a.prof_block = e ? atoi(e) : 0;
    }
    const int QD = c.n_heads * c.head_dim;
    int kmax = c.hidden > c.inter ? c.hidden : c.inter;
    kmax = kmax > QD ? kmax : QD;
    unsigned lds = (unsigned)gemv_lds(kmax).total;
    const int a_rep = c.n_heads / c.n_kv_heads, a_waves = attn_short_waves(a_rep), a_tpw = 64 / (c.head_dim / 8);
    const bool a_wide = PIE_ATTN_WIDE && (size_t)a_rep * a_waves * a_tpw * c.head_dim * 4 <= 65536;  // attention.hpp: WIDE
    const unsigned attn_lds = (unsigned)(a_rep * (a_wide ? a_waves * a_tpw : a_waves) * (c.head_dim + 2) * 4);  // the online-softmax streams
    lds = lds > attn_lds ? lds : attn_lds;
    lds = (lds + 15u) & ~15u;
    a.lds_rope = lds, lds += (unsigned)c.head_dim * 4;
    a.lds_ctl = lds, lds += 16;
    lds = lds > MEGA_LDS_MIN ? lds : MEGA_LDS_MIN;  // more than half a CU's LDS: exactly one workgroup per CU, all co-resident
    PIE_REQUIRE(lds <= 160u * 1024u, PIE_E_SHAPE, "mega step: activation image does not fit LDS");
    PIE_HIP_TRY(hipMemsetAsync(m->sync, 0, sizeof(MegaSync), st));
    const int rep = c.n_heads / c.n_kv_heads;
    const int G = m->n_cus;
    const bool big = kmax > 4 * 4096;  // more than 4 activation pieces per staging thread
    if (c.dtype == PIE_BF16) return big ? mega_launch_t<BF16, 4, 8>(a, c.head_dim, rep, G, lds, st) : mega_launch_t<BF16, 4, 4>(a, c.head_dim, rep, G, lds, st);
    return big ? mega_launch_t<F16, 4, 8>(a, c.head_dim, rep, G, lds, st) : mega_launch_t<F16, 4, 4>(a, c.head_dim, rep, G, lds, st);
}

void *mega_prof_ptr(pie_decoder *d) { return d->mega ? (void *)d->mega->prof : nullptr; }

int mega_status(pie_decoder *d, unsigned *err) {
    *err = 0;
    if (!d->mega || !d->mega->sync) return PIE_OK;
    PIE_HIP_TRY(hipMemcpy(err, &d->mega->sync->error, sizeof(unsigned), hipMemcpyDeviceToHost));
    return PIE_OK;
}
