// ring_gemv.hpp -- the consumer side of the weight ring: W4S row-pair dot products out of LDS slots.
//
// Same arithmetic, in the same order, as the launched GEMV (k_w4s_gemv in w4_gemv.hpp): per unit the four v_dot2c chains of
// w4s_unit_dot, per group one fma with {scale, bias}, a row pair's K slices accumulated in slice order in one register, the
// 32-lane DPP sum -- so a row sum is bit-identical whichever path produced it.  What differs is where operands come from:
// the unit's codes from a ring slot (filled by the loader wave's LDS-DMA), and, for matrices of two K slices (K <= 4096: q|k|v,
// o_proj, gate|up, lm_head of the 8B geometry = 73 % of its bytes), the activations from REGISTERS: both slices' 64 values per
// lane are read from the LDS image once per phase, so the loop's only LDS traffic is the slot itself (3 reads per unit instead
// of 12; measured in tools/ring_probe).
#pragma once
#include "ring.hpp"
#include "w4_gemv.hpp"

// x of one K slice for this lane: its quantisation group's 64 activations (pre-scaled, as staged in the image) and their sum
struct RingX {
    u32 xr[32];
    float sx;
    bool valid;
};
template <class T>
__device__ __forceinline__ void ring_load_x(const char *img, const GemvLds &L, int n_groups, int slice, int lane, RingX &x) {
    const int g = slice * 32 + (lane & 31);
    x.valid = g < n_groups;
    const int gc = x.valid ? g : n_groups - 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint4 v = *reinterpret_cast<const uint4 *>(img + ((size_t)q * L.stride + gc) * 16);
        x.xr[4 * q + 0] = v.x, x.xr[4 * q + 1] = v.y, x.xr[4 * q + 2] = v.z, x.xr[4 * q + 3] = v.w;
    }
    x.sx = reinterpret_cast<const float *>(img + L.off_sx)[gc];
}
template <class T>
__device__ __forceinline__ float ring_unit_term(const RingUnit &un, const RingX &x) {
    const float dd = w4s_unit_dot<T>(un.c0, un.c1, x.xr);
    const float scale = lo_f32<T>(un.sb), bias = hi_f32<T>(un.sb);
    const float pr = fmaf(scale, dd * T::DSCALE - T::OFFSET * x.sx, bias * x.sx);
    return x.valid ? pr : 0.0f;
}

// Consumes this wave's `n_pairs` row pairs of one [N, K] matrix from its ring; the two row sums of local pair p are left in
// outp[2 p], outp[2 p + 1] (LDS, this wave's region).  NSX = 1 / 2: K has exactly that many slices and x lives in registers
// (a slot then holds 4 / NSX whole row pairs: straight-line code); NSX = 0: any K, x re-read from the LDS image per unit.
// Returns false when a bounded wait gave up.
template <class T, int NSX>
__device__ __forceinline__ bool ring_consume(RingCursor &cur, const char *smem, unsigned lds0, const char *img, const GemvLds &L, int K, int n_pairs,
                                             float *outp, int lane, unsigned long long deadline) {
    const int n_groups = K >> 6;
    bool ok = true;
    if constexpr (NSX > 0) {
        constexpr int PPS = RING_SLOT_UNITS / NSX;  // row pairs per slot
        RingX x[NSX];
#pragma unroll
        for (int s = 0; s < NSX; ++s) ring_load_x<T>(img, L, n_groups, s, lane, x[s]);
        for (int p = 0; p < n_pairs; p += PPS) {
            const unsigned slot = ring_wait_slot(cur, deadline, ok);
            if (!ok) return false;
            const int np = n_pairs - p < PPS ? n_pairs - p : PPS;  // wave-uniform
            RingUnit un[RING_SLOT_UNITS];
#pragma unroll
            for (int k = 0; k < RING_SLOT_UNITS; ++k) un[k] = ring_read_unit(smem, lds0, slot, k < np * NSX ? k : 0, lane);
            ring_release_slot(cur);  // LDS executes a wave's accesses in order: the reads above are ahead of this store
#pragma unroll
            for (int q = 0; q < PPS; ++q) {
                if (q < np) {
                    float acc = 0.0f;  // 0 + term first, as the launched kernel's accumulator (matters for an all -0 row only)
#pragma unroll
                    for (int s = 0; s < NSX; ++s) acc += ring_unit_term<T>(un[q * NSX + s], x[s]);
                    const float tot = half_wave_sum(acc);
                    if ((lane & 31) == 31) outp[2 * (p + q) + (lane >> 5)] = tot;
                }
            }
        }
    } else {
        // x of the NEXT unit's slice is requested before the current unit is multiplied (two register sets, ping-pong): with one LDS
        // round trip per unit exposed (~0.1-0.2 us while the DMA and the other waves use the LDS) the down_proj stream drained at
        // 25 GB/s per CU, slower than the loaders deliver.
        const int ns = w4s_slices(K), n_units = n_pairs * ns;
        float acc = 0.0f;
        int sl = 0, pl = 0;
        RingX xa, xb;
        if (n_units > 0) ring_load_x<T>(img, L, n_groups, 0, lane, xa);
        for (int base = 0; base < n_units; base += RING_SLOT_UNITS) {
            const unsigned slot = ring_wait_slot(cur, deadline, ok);
            if (!ok) return false;
            const int nu = n_units - base < RING_SLOT_UNITS ? n_units - base : RING_SLOT_UNITS;
            RingUnit un[RING_SLOT_UNITS];
#pragma unroll
            for (int k = 0; k < RING_SLOT_UNITS; ++k) un[k] = ring_read_unit(smem, lds0, slot, k < nu ? k : 0, lane);
            ring_release_slot(cur);
#pragma unroll
            for (int k = 0; k < RING_SLOT_UNITS; ++k) {
                if (k < nu) {  // wave-uniform
                    const int nsl = sl + 1 == ns ? 0 : sl + 1;
                    if (k & 1) {
                        ring_load_x<T>(img, L, n_groups, nsl, lane, xa);
                        acc += ring_unit_term<T>(un[k], xb);
                    } else {
                        ring_load_x<T>(img, L, n_groups, nsl, lane, xb);
                        acc += ring_unit_term<T>(un[k], xa);
                    }
                    if (++sl == ns) {
                        const float tot = half_wave_sum(acc);
                        if ((lane & 31) == 31) outp[2 * pl + (lane >> 5)] = tot;
                        acc = 0.0f, sl = 0, ++pl;
                    }
                }
            }
        }
    }
    return ok;
}
