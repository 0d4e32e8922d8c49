// mfma_store.hpp -- the epilogue of the one-wave-per-SIMD MFMA GEMMs (k_w16l_gemm, k_w4l2_gemm): a wave's accumulator tile, rounded, through LDS
// to whole 128-byte lines.
//
// v_mfma_f32_32x32x16 leaves register i of lane l = 32 kh + n at output column 32 nt + (i & 3) + 8 (i >> 2) + 4 kh, row 32 mi + n: a lane holds four
// consecutive columns of 32 DIFFERENT rows.  Stored from there (8 bytes per lane and row) the output cost 12-27 % of the 16-bit kernel (tower qkv
// at 4096 rows 45.5 us, 33.0 without stores; gate|up 890 vs 781): every workgroup of a round finishes at the same time and the burst is made of
// 16-byte fragments.  So the rounded tile goes through LDS -- free after the K loop; each wave its own region, rows XOR-swizzled in 8-byte slots:
// conflict-free both ways -- and leaves as 16 bytes per lane, 8 (or 4) lanes per row.
#pragma once

// acc[s][mi]: strip nt0 + s (32 columns), row block mi of the wave's tile; ot: the wave's LDS region, 32 MB x OB bytes (OB = 64 SW, or 32 SW with
// SWIGLU); m0: the tile's first row, rows: its live rows; N: the Linear's out_features (stores clipped to it), ldy: output row stride (elements);
// bias (nullable): T(T(acc) + bias), like the text tower's Linear; SWIGLU: columns (2 i, 2 i + 1) = (gate_i, up_i) -> T(T(silu(g)) * u), N / 2 wide.
// The caller has made sure nobody still reads the LDS region (a barrier after the K loop).
template <class T, int MB, int SW, bool SWIGLU>
__device__ __forceinline__ void mfma_tile_store(const f32x16_t (&acc)[SW][MB], char *ot, int lane, int nt0, int m0, int rows, int N, int ldy, u16 *y,
                                                const u16 *bias, bool skip_stores = false) {
    constexpr int MT = 32 * MB;
    constexpr int OB = (SWIGLU ? 32 : 64) * SW;  // output bytes per row of the wave's tile
    constexpr int SL = OB / 8, RPB = 256 / OB;    // 8-byte slots per row; rows per 256-byte bank row
    const int n = lane & 31, kh = lane >> 5;
    const int sv = (n / RPB) & (SL - 1);  // the row's slot swizzle (rows 32 apart share it)
#pragma unroll
    for (int s = 0; s < SW; ++s) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * (nt0 + s) + 4 * kh + 8 * q;
                u32 p01 = w4m_pack<T>(acc[s][mi][4 * q], acc[s][mi][4 * q + 1]), p23 = w4m_pack<T>(acc[s][mi][4 * q + 2], acc[s][mi][4 * q + 3]);  // the Linear's rounding to T
                if (bias && col < N) {
                    uint2 bw;
                    if (col + 4 <= N && !(N & 3)) bw = *reinterpret_cast<const uint2 *>(bias + col);
                    else {  // any out_features: the last quad of a row may be partial, rows of the bias need not be 8-byte aligned
                        u16 e[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) e[i] = col + i < N ? bias[col + i] : (u16)0;
                        bw = make_uint2(e[0] | ((u32)e[1] << 16), e[2] | ((u32)e[3] << 16));
                    }
                    p01 = w4m_pack<T>(lo_f32<T>(p01) + lo_f32<T>(bw.x), hi_f32<T>(p01) + hi_f32<T>(bw.x));
                    p23 = w4m_pack<T>(lo_f32<T>(p23) + lo_f32<T>(bw.y), hi_f32<T>(p23) + hi_f32<T>(bw.y));
                }
                char *row = ot + (32 * mi + n) * OB;
                if (SWIGLU) {  // two activations, 4 bytes at byte 32 s + 8 q + 4 kh of the row
                    const float g0 = lo_f32<T>(p01), u0 = hi_f32<T>(p01), g1 = lo_f32<T>(p23), u1 = hi_f32<T>(p23);
                    const u16 o0 = T::from_f32(round_T<T>(g0 / (1.0f + expf(-g0))) * u0), o1 = T::from_f32(round_T<T>(g1 / (1.0f + expf(-g1))) * u1);
                    *reinterpret_cast<u32 *>(row + (((4 * s + q) ^ sv) << 3) + 4 * kh) = (u32)o0 | ((u32)o1 << 16);
                } else {       // 8 bytes at byte 64 s + 16 q + 8 kh
                    *reinterpret_cast<uint2 *>(row + (((8 * s + 2 * q + kh) ^ sv) << 3)) = make_uint2(p01, p23);
                }
            }
        }
    }
    // (the same wave reads what it wrote: no barrier, the compiler's lgkmcnt wait orders the LDS accesses)
    constexpr int PR = OB / 16;           // 16-byte pieces per row
    const int ldo = SWIGLU ? N >> 1 : N;  // output row length (the row stride ldy may be longer: a zero-padded operand of the next GEMM)
    const int c0 = (SWIGLU ? 16 : 32) * nt0;  // the wave's first output column
#pragma unroll
    for (int it = 0; it < MT * PR / 64; ++it) {
        const int pc = it * 64 + lane, r = pc / PR, j = pc % PR;
        const int rsv = ((r & 31) / RPB) & (SL - 1);
        const uint2 lo = *reinterpret_cast<const uint2 *>(ot + r * OB + (((2 * j) ^ rsv) << 3));
        const uint2 hi = *reinterpret_cast<const uint2 *>(ot + r * OB + (((2 * j + 1) ^ rsv) << 3));
        const int col = c0 + 8 * j;
        if (r >= rows || col >= ldo || skip_stores) continue;
        u16 *dst = y + (size_t)(m0 + r) * ldy + col;
        if (((ldo | ldy) & 7) == 0) *reinterpret_cast<uint4 *>(dst) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        else if (((ldo | ldy) & 3) == 0) {  // rows are 8-byte aligned; the last piece of a row may be half
            *reinterpret_cast<uint2 *>(dst) = lo;
            if (col + 4 < ldo) *reinterpret_cast<uint2 *>(dst + 4) = hi;
        } else {  // any out_features: element by element
            const u32 w[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (col + e < ldo) dst[e] = (u16)(w[e >> 1] >> (16 * (e & 1)));
        }
    }
}
