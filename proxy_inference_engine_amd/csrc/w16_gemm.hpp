// w16_gemm.hpp -- y[M, N] = x[M, K] . W[N, K]^T for 16-bit weights on the MFMA units (round 5): nn.Linear with dense weights on many
// rows -- the prompt pass of a dense checkpoint (models/llama/language.py:83,108,127 with no "quantization" entry, models/utils.py:96-97),
// the T copies the int8 / group-32 prompt path multiplies, and every Linear of the vision tower (models/intern/vision.py:87-442).
// Included by w4m_gemm.hip (shares its packing helpers and the K-split reduce).
//
// The kernel is k_w4l2_gemm without the conversion: one wave per SIMD with the whole register file (256 accumulators in AGPRs), the x
// tile [32 MB rows x 64 k] staged once per workgroup by LDS-DMA (source-side XOR swizzle, four buffers, two steps ahead), the weights NOT through LDS at all:
// W16M tiles hold them in the MFMA A-fragment order, so a wave's fragment is ONE coalesced 1 KiB load (global -> registers) and the
// registers a load fills ARE the operand.  Ring of four tiles per wave (2 strips x 4 k-steps x 16 B per lane = 128 registers), issued three
// steps ahead; per step one barrier and one counted wait.
//
// W16M layout (built once per matrix on the device: from row-major weights, from W16S units, or by the dequantisers): tiles of 32 output
// rows x 64 columns, 4096 B, tile (nt, g) at ((nt * ceil(K / 64)) + g) * 4096; inside, k-step s (0..3) is 1 KiB: lane l = 32 kh + n holds
// W[32 nt + n][64 g + 16 s + 8 kh .. + 8].  Rows past N and columns past K are zeros, so N and K of the Linear are free; x rows have to
// hold 64 ceil(K / 64) elements, zeros past K (any row stride: the host pads the tower's two odd widths, 1176 and 3420).
//
// K / 64 need not be a multiple of the 4-fold unrolled ring: the loop runs ceil4 steps, the first ones on a 4 KiB block of zeros.
#pragma once

constexpr int W16M_TILE_BYTES = 4096;
#ifndef W16L_ABL
#define W16L_ABL 0  // developer ablation mask (tools/w16_bench; 0 in the product): 1 no weight loads, 2 no barrier, 4 no x DMA, 8 no LDS fragment reads, 16 / 32 weight loads / x DMA always of group 0 (cache-hot), 64 no stores
#endif

static inline size_t w16m_bytes(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 63) / 64) * W16M_TILE_BYTES; }

// row-major [N][K] -> W16M.  One thread per 16-byte piece of the output.
__global__ void __launch_bounds__(256) k_rows_to_w16m(const u16 *w, int N, int K, int groups, size_t pieces, uint4 *out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pieces) return;
    const int l = (int)(i & 63), s = (int)((i >> 6) & 3);
    const size_t tile = i >> 8;
    const int g = (int)(tile % groups), nt = (int)(tile / groups);
    const int row = 32 * nt + (l & 31), k0 = 64 * g + 16 * s + 8 * (l >> 5);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N && k0 < K) {
        const u16 *src = w + (size_t)row * K + k0;
        if (k0 + 8 <= K && (K & 7) == 0) v = *reinterpret_cast<const uint4 *>(src);
        else {
            u16 e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = k0 + j < K ? src[j] : (u16)0;
            v = make_uint4(e[0] | ((u32)e[1] << 16), e[2] | ((u32)e[3] << 16), e[4] | ((u32)e[5] << 16), e[6] | ((u32)e[7] << 16));
        }
    }
    out[i] = v;
}

// W16S units (w4_gemv.hip: row pair x 512-wide slice, lane (row, 16-element chunk), two 16-byte pieces j) -> W16M, packed row order
__global__ void __launch_bounds__(256) k_w16s_to_w16m(const uint4 *w16s, int N, int K, int ns, int groups, size_t pieces, uint4 *out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pieces) return;
    const int l = (int)(i & 63), s = (int)((i >> 6) & 3);
    const size_t tile = i >> 8;
    const int g = (int)(tile % groups), nt = (int)(tile / groups);
    const int r = 32 * nt + (l & 31), k0 = 64 * g + 16 * s + 8 * (l >> 5);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < N && k0 < K) v = w16s[(((size_t)(r >> 1) * ns + (k0 >> 9)) << 7) + (((k0 >> 3) & 1) << 6) + (r & 1) * 32 + ((k0 & 511) >> 4)];
    out[i] = v;
}

struct W16Args {
    const char *w16m;
    const u16 *x;   // [M] rows of ldx elements, the first K = 64 * groups of them used
    int M, N, K;    // N: the Linear's output features (stores are clipped to it; any N -- 16-byte stores when N % 8 == 0)
    int ldx;        // x row stride (elements, >= K, a multiple of 8)
    int ldy;        // y row stride (elements, >= N or N / 2, a multiple of 4)
    u16 *y;         // [M][N], or the MLP activation [M][N / 2] (SWIGLU)
    float *part;    // gridDim.y > 1: fp32 slabs [z][M][N]
    const u16 *bias;
    const char *zero;  // 4 KiB of zeros (the dummy steps' weight tile)
    int tm, tn;     // row tiles; column tiles (of 128 SW columns)
    int bm, bc;     // the XCD-local traversal's block: bm row tiles x bc column tiles (bm * bc ~ the 32 workgroups an XCD runs at once)
#ifdef W16L_PROF
    unsigned long long *prof;  // per workgroup {s_memtime, s_memrealtime} at entry and exit (tools/w16_bench prof)
#endif
};

// MB: 32-row blocks of x per workgroup; SW: strips (32 columns) per wave -- the workgroup's tile is 32 MB rows x 128 SW columns;
// SWIGLU: the packed gate|up matrix (columns (2 i, 2 i + 1) = (gate_i, up_i)): y = T(silu(T(gate)) * T(up)), as k_w4l2_gemm.
// grid.x = 8 * ceil(tn / 8) * tm, XCD-aware: the workgroups an XCD runs together (x % 8 equal, consecutive x / 8) are the row tiles of
// one column tile, then of the column tile 8 further -- they stream the SAME weight tiles at the same time, so each XCD's L2 fetches a
// weight tile once per pass (16-bit weights are 4x the int4 kernel's bytes: re-read per row tile they would be 3.7 GB of HBM traffic for
// gate|up at 4096 rows); grid.y = K splits.
template <class T, int MB, int SW, bool SWIGLU>
__global__ void __launch_bounds__(256) k_w16l_gemm(const W16Args a) {
    constexpr int MT = 32 * MB, XJ = MB;  // rows per tile; DMA instructions per wave and tile (MT / 8 row groups over 4 waves)
    constexpr int WL = 4 * SW;            // weight loads per wave and step
    constexpr int XB = 4, CHUNK = MT * 128;  // x buffers (tile v lives in buffer v % 4); bytes each
    __shared__ __attribute__((aligned(1024))) char s_x[XB * CHUNK];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kh = lane >> 5, all_groups = a.K >> 6;
    const int per_z = (all_groups + (int)gridDim.y - 1) / (int)gridDim.y;
    const int g_lo = blockIdx.y * per_z, g_hi = min(all_groups, g_lo + per_z);
    const int groups = g_hi - g_lo;                   // >= 1 (launcher)
    const int nv = (groups + 3) & ~3, pad = nv - groups;  // virtual steps; leading dummies
    // workgroup -> tile: XCD x = blockIdx.x % 8 owns the column tiles c * 8 + x; it walks its [tm][cl] tiles block by block (bc columns,
    // then bm rows inside; row tile fastest), so the ~32 workgroups it runs at once share bm x tiles and bc weight tiles through its L2
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int cl = (a.tn + 7) >> 3;                       // local columns (the grid holds tm * cl workgroups per XCD)
    const int cb = sq / (a.tm * a.bc), r1 = sq - cb * a.tm * a.bc;
    const int bce = min(a.bc, cl - cb * a.bc);            // columns of this (maybe last, narrower) column block
    const int mbi = r1 / (a.bm * bce), r2 = r1 - mbi * a.bm * bce;
    const int bme = min(a.bm, a.tm - mbi * a.bm);
    const int mt = mbi * a.bm + r2 % bme, ncol = (cb * a.bc + r2 / bme) * 8 + xcd;
    if (ncol >= a.tn) return;  // (whole workgroup, before any barrier)
    const int m0 = mt * MT;
    const int rows = a.M - m0 < MT ? a.M - m0 : MT;
    const int n_strips = (a.N + 31) >> 5;
    const int nt0 = (ncol * 4 + wave) * SW;  // this wave's strips: nt0 .. nt0 + SW
#ifdef W16L_PROF
    if (threadIdx.x == 0 && a.prof) a.prof[4 * blockIdx.x] = __builtin_amdgcn_s_memtime(), a.prof[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    const char *strip[SW];
    bool has[SW];
#pragma unroll
    for (int s = 0; s < SW; ++s) {
        has[s] = nt0 + s < n_strips;  // wave-uniform; idle waves still stage x and join the barriers
        strip[s] = a.w16m + ((size_t)(has[s] ? nt0 + s : 0) * all_groups + g_lo) * W16M_TILE_BYTES;
    }

    // x staging: wave w moves rows [8 MB w, 8 MB (w + 1)) of the tile with MB DMA instructions of 8 rows each; per-lane byte offsets from
    // the tile's first row (the instruction adds a scalar base that advances by one group per step: no vector arithmetic per DMA)
    u32 xoffs[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int r = 8 * MB * wave + 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int rr = r < rows ? r : rows - 1;  // ragged tile: rows past the end repeat the last one (never stored)
        xoffs[j] = (u32)(((size_t)rr * a.ldx + (size_t)c * 8) * 2);
    }
    const unsigned long long xbase = (unsigned long long)(uintptr_t)(a.x + (size_t)m0 * a.ldx + (size_t)g_lo * 64);
    const unsigned lds0 = (unsigned)(size_t)(w4l_lds_void *)s_x;
    // (issued through asm: see k_w4l2_gemm; M0 = the wave's LDS destination)
    auto x_issue1 = [&](int v, int buf, int j) {  // tile v -> buffer buf (= v % 4, spelled statically by the callers)
        if (W16L_ABL & 4) return;
        int g = v - pad < 0 ? 0 : (v - pad < groups ? v - pad : groups - 1);  // a dummy step multiplies group 0 by zeros
        if (W16L_ABL & 32) g = 0;
        const unsigned long long srcv = xbase + (unsigned long long)g * 128ull;
        const unsigned src_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)srcv);  // (wave-uniform by construction; spelled out for the asm's scalar operand)
        const unsigned src_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(srcv >> 32));
        const unsigned long long src = (unsigned long long)src_lo | ((unsigned long long)src_hi << 32);
        const unsigned dst = lds0 + (unsigned)((buf & (XB - 1)) * CHUNK + (8 * MB * wave + 8 * j) * 128);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(xoffs[j]), "s"(src), "s"(dst)
                     : "memory");
    };
    // weight ring: tile v lives in slot v % 4 from its issue (step v - 3) to the end of step v; its registers are the A fragments
    uint4 wr[4][SW][4];
    auto w_issue1 = [&](int slot, int v, int i) {  // load i of tile v: strip i % SW, k-step i / SW
        if (W16L_ABL & 1) return;
        const int g = v - pad;
        const int s = i % SW, k = i / SW;
        const char *p = g < 0 ? a.zero : strip[s] + (size_t)((W16L_ABL & 16) ? 0 : (g < groups ? g : groups - 1)) * W16M_TILE_BYTES;  // (wave-uniform select)
        wr[slot][s][k] = *(reinterpret_cast<const uint4 *>(p + 1024 * k) + lane);
    };
    // prologue, in the queue order of the steady state: [x(0)] [W(0)] [x(1)] [W(1)] [W(2)]
#pragma unroll
    for (int j = 0; j < XJ; ++j) x_issue1(0, 0, j);
#pragma unroll
    for (int i = 0; i < WL; ++i) w_issue1(0, 0, i);
#pragma unroll
    for (int j = 0; j < XJ; ++j) x_issue1(1, 1, j);
#pragma unroll
    for (int d = 1; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < WL; ++i) w_issue1(d, d, i);
    if (W16L_ABL & 1) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int s = 0; s < SW; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) wr[d][s][k] = make_uint4(lane + d, s, k, 0x3c003c00u);
    }
    // an (empty) statement with an AGPR operand: without any, hipcc marks the kernel "no AGPRs needed" and may select the VGPR form of
    // the MFMAs, using the AGPR half of the file as a spill area
    asm volatile("" : : "a"(0.0f));

    f32x16_t acc[SW][MB];
#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s][mi][i] = 0.0f;
    // B fragment of (row block mi, k-step k): row 32 mi + n, 16-byte chunk 2 k + kh, swizzled by the row
    int xoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xoff[k] = n * 128 + (((2 * k + kh) ^ ((n >> 1) & 7)) << 4);
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(XJ + 2 * WL) : "memory");  // x(0) (and W(0)) have landed
    __syncthreads();

    // Four copies of the step with static ring slots and tile buffers and NO control flow around the accumulators; the order below is
    // pinned pair by pair (sched_barrier): left to the scheduler the fragment reads sink to one pair (64 cycles) before their use and the
    // weight loads bunch up at the end of the step -- 0.96 instead of 1.2+ PFLOP/s.
    //   pairs 0 .. NS-5: MFMAs of (k-step, row block) pair t; this step's memory issue, spread; the B fragment of pair t + 4
    //   pair NS-4:       the step's ONE wait + barrier: x(v + 1) and W(v + 1) (issued a whole step ago) have landed, for everyone,
    //                    and everyone has issued its last read of buffer v % 4
    //   pairs NS-4 .. :  MFMAs; the B fragments of the NEXT step's pairs 0 .. 3 (its buffer is complete now)
    // Queue order per step: [x DMA of tile v + 2][weight tile v + 3], all before the wait: younger than x(v + 1) are W(v + 2), x(v + 2), W(v + 3).
    constexpr int NS = 4 * MB;       // (k-step, row block) pairs of a step
    constexpr int OPS = XJ + WL;     // memory instructions a step issues
    constexpr int NM = NS - 4;       // ... spread over the pairs before the barrier
    // B fragments: ring of four, read three pairs ahead (into the slot the previous pair has released)
    uint4 bq[4];
    auto b_read = [&](const char *buf, int t) { return *reinterpret_cast<const uint4 *>(buf + (t % MB) * 4096 + xoff[t / MB]); };
#pragma unroll
    for (int t = 0; t < 3; ++t) bq[t] = b_read(s_x, t);
    for (int base = 0; base < nv; base += 4) {
        pa_static_for<0, 4>([&](auto dc) {  // (compile-time loops: a `#pragma unroll` the compiler declines would index the ring dynamically -- in scratch)
            constexpr int d = decltype(dc)::value;
            const int v = base + d;
            const char *xb = s_x + d * CHUNK, *xn = s_x + ((d + 1) & 3) * CHUNK;
            pa_static_for<0, NS>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int k = t / MB, mi = t % MB;
                if constexpr (t == NM) {
                    if (W16L_ABL & 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(XJ) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(XJ + 2 * WL) : "memory");
                    if (!(W16L_ABL & 2)) __syncthreads();
                }
                // An MFMA that finds the matrix pipe busy holds the wave's issue until it is free, so what sits between two MFMAs has to
                // fit the ~28 cycles the pipe still works after the second has entered: one thing per gap -- the fragment read behind the
                // first MFMA of the pair, the memory instruction behind the second
                acc[0][mi] = MfmaT<T>::run(wr[d][0][k], bq[t & 3], acc[0][mi]);
                __builtin_amdgcn_sched_barrier(0);
                if (!(W16L_ABL & 8)) {
                    if constexpr (t + 3 < NS) bq[(t + 3) & 3] = b_read(xb, t + 3);
                    else bq[(t + 3) & 3] = b_read(xn, t + 3 - NS);
                }
                if constexpr (SW == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    acc[1][mi] = MfmaT<T>::run(wr[d][1][k], bq[t & 3], acc[1][mi]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (t < NM) {
                    pa_static_for<0, OPS>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr ((i * NM) / OPS == t) {
                            if constexpr (i < XJ) x_issue1(v + 2, d + 2, i);     // buffer (v + 2) % 4 was last read in step v - 2
                            else w_issue1((d + 3) & 3, v + 3, i - XJ);          // that slot held tile v - 1
                        }
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the clamped look-ahead loads of the last steps
#ifdef W16L_PROF
    if (threadIdx.x == 0 && a.prof) a.prof[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime(), a.prof[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
#endif
    // epilogue: fp32 slabs of a K split straight from the accumulator layout; everything else rounded, through LDS, as whole lines (mfma_store.hpp)
    if (!SWIGLU && a.part) {  // K split: fp32 slabs, 16 bytes per lane and row (few rows by construction)
#pragma unroll
        for (int s = 0; s < SW; ++s) {
            if (!has[s]) continue;
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
                const int m = 32 * mi + n;
                if (m >= rows) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = 32 * (nt0 + s) + 4 * kh + 8 * q;
                    if (col >= a.N) continue;  // (N % 4 == 0: a quad is inside or outside)
                    *reinterpret_cast<float4 *>(a.part + (size_t)blockIdx.y * a.M * a.N + (size_t)(m0 + m) * a.N + col) =
                        make_float4(acc[s][mi][4 * q], acc[s][mi][4 * q + 1], acc[s][mi][4 * q + 2], acc[s][mi][4 * q + 3]);
                }
            }
        }
        return;
    }
    static_assert(4 * MT * ((SWIGLU ? 32 : 64) * SW) <= XB * CHUNK, "the output tile fits the x buffers");
    __syncthreads();  // everyone has read its last B fragments
    mfma_tile_store<T, MB, SW, SWIGLU>(acc, s_x + wave * (MT * (SWIGLU ? 32 : 64) * SW), lane, nt0, m0, rows, a.N, a.ldy, a.y, a.bias, (W16L_ABL & 64) && a.M > 1);
}
