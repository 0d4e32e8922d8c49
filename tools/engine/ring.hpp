// ring.hpp -- the per-CU weight ring of the persistent decode step: one LOADER wave streams W4S units from HBM into LDS by
// LDS-DMA (global_load_lds_dwordx4, non-temporal) and never waits for the activation chain; three CONSUMER waves (one per
// remaining SIMD) run the W4S dot products out of LDS.  Shared by step_engine.hip and tools/ring_probe.hip.
//
// Why a ring and not the launched GEMV's register stream (w4_gemv.hpp): weights do not depend on activations, only consumers
// must wait at a dependency edge.  With the stream decoupled from the consumers, HBM keeps flowing through every edge of the
// layer (the ring holds ~4 us of stream per CU) -- the structure MI355X_MICROARCH.md's price list measures as winning for
// batch-1 decode chains (rows prefetch-credit, ldsdma-fill, engine-vs-launches).
//
// Geometry: a consumer wave owns a private ring of RING_SLOTS slots; a slot is 4 consecutive W4S units of that wave's unit
// stream (4 x 2304 B = 9216 B = exactly nine 1-KiB DMA pieces); the last slot of a matrix may hold fewer units (fewer pieces).
// Handshake: two monotonic counters per consumer wave in LDS -- FULL (fills landed, written by the loader behind a counted
// s_waitcnt vmcnt) and FREE (slots released, written by the consumer once the slot's last unit sits in registers).
#pragma once
#include <type_traits>

#include "common.hpp"

#ifndef PIE_RING_CONSUMERS
#define PIE_RING_CONSUMERS 6
#endif
// Measured with tools/ring_probe on MI355X (66 MB gate|up matrix x 12, bit-exact row sums; TB/s chip-wide, loaders-consumers-slots-in flight):
//   1-3-4-4: 3.65 (consumers alone on their SIMDs run at ~8 cycles per instruction) | 1-6-2-4: 4.65 | 1-7-2-4: 5.4 | 2-6-2-3: 6.0 |
//   2-6-2-2: 6.5 | 2-7-2-3: 6.3; loaders alone (consumers only release): one 5.1-5.8, two 6.4-6.7 -- an LDS-DMA piece costs its issuing
//   wave ~60 cycles, so one loader wave cannot issue 25 GB/s per CU next to its bookkeeping.  The launched GEMV streams the same matrix at 5.3.
constexpr int RING_CONSUMERS = PIE_RING_CONSUMERS;  // consumer waves per CU
#ifndef PIE_RING_LOADERS
#define PIE_RING_LOADERS 2
#endif
constexpr int RING_LOADERS = PIE_RING_LOADERS;      // loader waves: waves 0 .. RING_LOADERS-1 (loader l feeds the consumers c with c % RING_LOADERS == l)
constexpr int RING_WAVES = RING_CONSUMERS + RING_LOADERS;
constexpr int RING_SLOT_UNITS = 4;
constexpr int RING_SLOT_BYTES = RING_SLOT_UNITS * W4S_UNIT_BYTES;  // 9216
#ifndef PIE_RING_SLOTS
#define PIE_RING_SLOTS 2
#endif
constexpr int RING_SLOTS = PIE_RING_SLOTS;                          // per consumer wave
constexpr int RING_WAVE_BYTES = RING_SLOTS * RING_SLOT_BYTES;      // 36 KiB
constexpr int RING_BYTES = RING_CONSUMERS * RING_WAVE_BYTES;       // 108 KiB per CU
#ifndef PIE_RING_INFLIGHT
#define PIE_RING_INFLIGHT 2
#endif
constexpr int RING_INFLIGHT = PIE_RING_INFLIGHT;  // fills the loader keeps in flight (9 DMA pieces each; vmcnt holds 63)
static_assert(RING_INFLIGHT * 9 <= 63 && RING_INFLIGHT >= 1 && RING_INFLIGHT <= 6 && (RING_INFLIGHT - 1) * 9 <= 45 && RING_CONSUMERS <= 8, "in-flight fills must fit vmcnt and the 30-bit FIFO");

// ---- LDS words shared between waves of the workgroup: accesses the compiler can neither cache nor hoist nor make flat
__device__ __forceinline__ unsigned lds_ld(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned lds_ld_s(unsigned addr) { return __builtin_amdgcn_readfirstlane(lds_ld(addr)); }  // wave-uniform result (SGPR)
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// fire and forget; `lane_one` = 1 in lane 0 and 0 elsewhere: every active lane of a DS atomic adds its own operand
__device__ __forceinline__ void lds_inc(unsigned addr, unsigned lane_one) { asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(lane_one) : "memory"); }
__device__ __forceinline__ unsigned lds_add_rtn(unsigned addr, unsigned v) {
    unsigned r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const void *p) { return (unsigned)(unsigned long long)p; }  // flat LDS address: low 32 bits = LDS offset

__device__ __forceinline__ unsigned ring_uniform(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool ring_uniform(bool v) { return __builtin_amdgcn_readfirstlane((unsigned)v) != 0u; }
__device__ __forceinline__ const char *ring_uniform(const char *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));  // unsigned first: the builtin returns int
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}

// One 1-KiB LDS-DMA piece: lane l's 16 bytes at gsrc land at lds_dst + 16 l.  Inline asm on purpose: hipcc does not count
// it, so the loader's own LDS accesses do not drain it (guide 5.7 item 1); completion is the loader's counted vmcnt.
__device__ __forceinline__ void glds16_nt(const char *gsrc_lane, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_dst_uniform)
                 : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n <= N (the immediate must be a constant): a short compare chain, N + 1 waits of code
template <int N>
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    if constexpr (N == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (n >= N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
        else wait_vmcnt_dyn<N - 1>(n);
    }
}

// ---- how a matrix's row pairs are dealt to the chip's consumer waves
// n_pairs pairs in groups of `gran`, contiguous runs per consumer wave; stream index i = wave * n_cus + cu, so the (at most one
// group of) imbalance lands on the low waves of every CU alike and each CU streams the same number of bytes (+- one group).
struct RingRun {
    int first, count;  // first row pair, number of row pairs
};
static inline __host__ __device__ RingRun ring_run(int n_pairs, int gran, int n_cus, int cu, int wave) {
    const int n_streams = n_cus * RING_CONSUMERS, i = wave * n_cus + cu;
    const int n_groups = n_pairs / gran;  // host-checked: n_pairs % gran == 0
    const int base = n_groups / n_streams, rem = n_groups % n_streams;
    RingRun r;
    r.first = gran * (i * base + (i < rem ? i : rem));
    r.count = gran * (base + (i < rem ? 1 : 0));
    return r;
}

// ---- ring bookkeeping words in LDS (byte offsets from the control block): FULL[3], FREE[3], then engine-specific words
constexpr unsigned RING_CTL_FULL = 0, RING_CTL_FREE = 32, RING_CTL_USER = 64;

// One whole slot (nine 1-KiB pieces, 9216 contiguous bytes at `src`) in ONE statement: saddr form (SGPR base + lane offset +
// immediate), M0 re-pointed every four pieces (the immediate reaches 4095 and applies to the LDS address as well).  The
// loader wave is alone on its SIMD and issues one instruction per ~5 cycles, so the fill must be a few dozen instructions:
// the first version (a loop of guarded single pieces, runtime wave selects) took ~1 us per fill = 2.3 TB/s chip-wide.
__device__ __forceinline__ void glds_fill9_nt(const char *src_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
    unsigned keep;
    src_uniform = ring_uniform(src_uniform), lds_dst_uniform = ring_uniform(lds_dst_uniform);
    const char *p1 = src_uniform + 4096, *p2 = src_uniform + 8192;
    const unsigned d1 = lds_dst_uniform + 4096, d2 = lds_dst_uniform + 8192;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3 nt\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:3072 nt\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(src_uniform), "s"(p1), "s"(p2), "s"(lds_dst_uniform), "s"(d1), "s"(d2)
        : "memory");
}

// Loader side.  A `segment` is one consumer wave's byte range of one matrix; the engine / probe describes them through the STATELESS
// callback segment(w, m, &ptr, &bytes): consumer wave w's share of matrix number m (0 bytes = none), false once m is past the last
// matrix.  Everything here is wave-uniform (SGPRs) and the per-wave state is indexed by compile-time constants only: a callback
// that kept its own per-wave cursor in an array had it turned into a dynamically indexed scratch array -- VGPR values, and with
// them the whole loader, off the scalar unit.
template <int W, class F>
__device__ __forceinline__ void ring_static_for(F &&f) {  // f(integral_constant<w>) for w = W, W + RING_LOADERS, ... < RING_CONSUMERS
    if constexpr (W < RING_CONSUMERS) {
        f(std::integral_constant<int, W>{});
        ring_static_for<W + RING_LOADERS>(f);
    }
}

template <int L0, class NextSeg>
__device__ __forceinline__ bool ring_loader(unsigned ring_base, unsigned ctl, int lane, NextSeg next_segment, unsigned long long deadline) {
    const char *ptr[RING_CONSUMERS];
    unsigned rem[RING_CONSUMERS], issued[RING_CONSUMERS], freec[RING_CONSUMERS];
    bool more[RING_CONSUMERS];
    int seg[RING_CONSUMERS];
    auto advance = [&](auto wc) {  // wave wc's next non-empty segment
        constexpr int w = decltype(wc)::value;
        unsigned b = 0;
        bool m = true;
        const char *p = nullptr;
        while (m && b == 0) {
            m = ring_uniform(next_segment(w, seg[w], &p, &b));
            b = m ? ring_uniform(b) : 0u;
            ++seg[w];
        }
        more[w] = m, rem[w] = b, ptr[w] = ring_uniform(p);
    };
    ring_static_for<L0>([&](auto wc) {
        constexpr int w = decltype(wc)::value;
        issued[w] = freec[w] = 0, seg[w] = 0;
        advance(wc);
    });
    // in-flight fills, oldest in the low bits, 8 bits each: wave << 4 | pieces.  Branch-free push / pop (a register FIFO with
    // an if-chain on the fill count compiled to a page of branches, and the loader wave pays ~5 cycles per instruction).
    unsigned long long fifo = 0;
    int n_inflight = 0, out_pieces = 0;
    const unsigned lane_off = (unsigned)lane * 16u, lane_one = lane == 0 ? 1u : 0u;
    auto retire = [&]() {  // wait until only the younger fills' pieces are outstanding, then publish the oldest
        const unsigned e = (unsigned)fifo & 255u;
        fifo >>= 8;
        out_pieces -= (int)(e & 15u);
        wait_vmcnt_dyn<9 * (RING_INFLIGHT - 1)>(out_pieces);  // first test = the steady state (whole slots in flight)
        lds_inc(ctl + RING_CTL_FULL + 4u * (e >> 4), lane_one);
        --n_inflight;
    };
    auto any_more = [&]() {
        bool m = false;
#pragma unroll
        for (int w = L0; w < RING_CONSUMERS; w += RING_LOADERS) m |= more[w];
        return m;
    };
    for (;;) {
        bool any = false;
        constexpr int max_inflight = RING_INFLIGHT;
        ring_static_for<L0>([&](auto wc) {
            constexpr int w = decltype(wc)::value;
            if (!more[w]) return;
            if (issued[w] - freec[w] >= (unsigned)RING_SLOTS) freec[w] = lds_ld_s(ctl + RING_CTL_FREE + 4 * w);  // looks full: refresh
            if (issued[w] - freec[w] >= (unsigned)RING_SLOTS) return;
            while (n_inflight >= max_inflight) retire();
            const unsigned dst = ring_base + (unsigned)w * RING_WAVE_BYTES + (issued[w] % RING_SLOTS) * RING_SLOT_BYTES;
            unsigned take = RING_SLOT_BYTES;
            if (rem[w] >= (unsigned)RING_SLOT_BYTES) {
                glds_fill9_nt(ptr[w], lane_off, dst);
            } else {  // the tail of this wave's share of a matrix: fewer pieces, the last one partial
                take = rem[w];
                const int pieces = (int)((take + 1023u) >> 10);
                for (int p = 0; p < pieces; ++p)
                    if ((unsigned)(p * 1024) + lane_off < take) glds16_nt(ptr[w] + p * 1024 + lane_off, __builtin_amdgcn_readfirstlane(dst + p * 1024));
            }
            const unsigned pieces = (take + 1023u) >> 10;
            ++issued[w];
            fifo |= (unsigned long long)(((unsigned)w << 4) | pieces) << (8 * n_inflight);
            ++n_inflight, out_pieces += (int)pieces;
            ptr[w] += take, rem[w] -= take;
            if (rem[w] == 0) advance(wc);
            any = true;
        });
        if (!any) {
            if (n_inflight > 0) retire();  // every ring full or finished: publish what is in flight, the consumers may be waiting for exactly that
            else if (!any_more()) break;
            else {
                if (__builtin_amdgcn_s_memrealtime() > deadline) return false;  // the consumers gave up (or never will release): drain
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    return true;
}

// Consumer side: position in this wave's ring.
struct RingCursor {
    unsigned fills;     // fills of this wave consumed so far (== slots released)
    unsigned ring;      // LDS byte address of this wave's ring
    unsigned full_word, free_word;
};
__device__ __forceinline__ RingCursor ring_cursor(unsigned ring_base, unsigned ctl, int cwave) {
    RingCursor c;
    c.fills = 0, c.ring = ring_base + (unsigned)cwave * RING_WAVE_BYTES;
    c.full_word = ctl + RING_CTL_FULL + 4 * cwave, c.free_word = ctl + RING_CTL_FREE + 4 * cwave;
    return c;
}
// Blocks until fill number c.fills has landed; returns its LDS address.  `give_up` bounds the spin (s_memrealtime ticks).
__device__ __forceinline__ unsigned ring_wait_slot(const RingCursor &c, unsigned long long deadline, bool &ok) {
    unsigned spins = 0;
    while ((int)(lds_ld_s(c.full_word) - c.fills) <= 0) {
        if ((++spins & 1023u) == 0u && __builtin_amdgcn_s_memrealtime() > deadline) {  // the clock is read through scalar memory: not on every poll
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return c.ring + (c.fills % RING_SLOTS) * RING_SLOT_BYTES;
}
__device__ __forceinline__ void ring_release_slot(RingCursor &c) {  // the caller's LDS reads of the slot have completed (lgkmcnt(0))
    ++c.fills;
    lds_st(c.free_word, c.fills);
}

// One W4S unit out of the ring: the lane's two 16-byte code pieces and its {scale | bias << 16} word.
struct RingUnit {
    uint4 c0, c1;
    u32 sb;
};
__device__ __forceinline__ RingUnit ring_read_unit(const char *smem, unsigned smem_lds_base, unsigned slot_addr, int u, int lane) {
    const char *p = smem + (slot_addr - smem_lds_base) + u * W4S_UNIT_BYTES;
    RingUnit r;
    r.c0 = *reinterpret_cast<const uint4 *>(p + lane * 16);
    r.c1 = *reinterpret_cast<const uint4 *>(p + 1024 + lane * 16);
    r.sb = *reinterpret_cast<const u32 *>(p + 2048 + lane * 4);
    return r;
}
