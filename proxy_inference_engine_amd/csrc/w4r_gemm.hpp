// w4r_gemm.hpp -- int4 g=64 x T GEMM for 6 .. 256 rows: every CU streams its slab of W4M tiles ONCE, at the rate the decode GEMV
// streams them, converts each tile once and multiplies it into all the rows on the matrix cores.  (Included by w4m_gemm.hip.)
//
// Replaces, for prompts / suffixes / multi-sequence steps of 6 .. 256 rows, MLX's qmm regime of mx.quantized_matmul as reached from
// nn.QuantizedLinear (models/llama/language.py:83,108,127; the prompt pass of engine/inference_engine.py:277-279): weights dequantised to T,
// T x T products, fp32 accumulation, one rounding.
//
// Why a third form (round 5): the few-row kernels (k_w4m_gemm*, <= 32 rows) stream at <= 2.5 TB/s (one 32-row strip or four per workgroup,
// every strip re-reading x), the many-row kernel (k_w4l2_gemm) at 64 / 128 rows re-converts every weight tile per 64-row tile and leaves
// half the chip idle on gate|up (112 column workgroups): a 128-token prompt ran at 0.13-0.16 of either roofline (profiles/r02_prefill128).
// What the numbers say (tools/dequant_probe, profiles/r05_dequant_probe.txt): converting a tile costs ~290 VALU cycles per SIMD whatever the
// occupancy, a CU takes in ~26 GB/s of HBM = one tile per ~370 cycles and SIMD, the MFMAs of a tile take 128 cycles per 32 rows.  So up
// to ~64 rows the weight stream bounds the kernel, beyond ~96 rows the matrix cores do -- IF conversion, stream and MFMAs overlap and every
// CU works.  Design:
//   * workgroup = WAVES waves = (WAVES / KW strip groups) x (KW K-phases); a wave owns SPW 32-column strips and, per K step, ONE 64-wide
//     quantisation group of them (its phase): raw tiles ride a register ring XB steps deep (non-temporal loads, 1 KiB contiguous per
//     instruction), are converted in registers into MFMA A fragments (w4r_dequant: 17 VALU instructions per 8 weights) and multiplied
//     into all MB 32-row blocks of x: the conversion is done once per weight on the whole chip;
//   * two waves per SIMD: one wave's conversion runs beside the other's MFMAs without any software pipelining;
//   * the x chunk of a step ([32 MB rows] x [KW x 64 columns]) is staged ONCE per workgroup by LDS-DMA (no registers, no ds_write) into a
//     ring of XB buffers, XB - 1 steps ahead, rows XOR-swizzled on the source side so that the B-fragment ds_read_b128 are conflict-free;
//     one barrier per step; the closing wait is a COUNTED vmcnt that leaves the younger DMAs and weight tiles in flight;
//   * the K-phases' partial tiles are summed through LDS in phase order (deterministic) by all 512 threads, 8 consecutive output columns
//     per thread, which is also the shape the epilogues want: store (+ bias), SwiGLU on the interleaved gate|up columns, RoPE + cache
//     append on the packed q|k|v columns, or un-rounded fp32 slabs of a K split that the consumer kernel sums (W4lSlabs, prefill.hip).
#pragma once

enum { W4R_STORE = 0, W4R_SWIGLU = 1, W4R_ROPE = 2, W4R_SLAB = 3 };
#ifndef W4R_ABL
#define W4R_ABL 0  // developer ablation mask (tools/w4r_bench): 1 no conversion, 2 no MFMA, 4 no x DMA, 8 no LDS fragment reads, 16 no weight loads; 0 in the product
#endif

struct W4rArgs {
    const char *w4m;   // W4M tiles of the [N, K] matrix
    const u16 *x;      // [M, K]
    int M, N, K;
    int steps;         // K steps (of KW groups) per blockIdx.y
    u16 *y;            // STORE: [M, N]; SWIGLU: the activation [M, N / 2]
    float *part;       // SLAB: [gridDim.y][M][N] un-rounded fp32 sums
    const u16 *bias;   // the Linear's bias (nullable; STORE / SWIGLU; ROPE takes W4mRope::bias)
    int epi;           // W4R_STORE / W4R_SWIGLU / W4R_ROPE / W4R_SLAB (run-time: the epilogues are outside the loop, one kernel per geometry)
#ifdef W4R_PROF
    unsigned long long *prof;  // developer build: [workgroup][wave][step][4] s_memtime stamps (tools/w4r_bench)
#endif
};

// v_fma_mix_f32 with an f16 first operand taken from the low / high half of a dword: D = fp32(h) * s + b, one rounding
__device__ __forceinline__ float w4r_mix_lo(u32 h, float s, float b) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(s), "v"(b));
    return d;
}
__device__ __forceinline__ float w4r_mix_hi(u32 h, float s, float b) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(s), "v"(b));
    return d;
}
// 8 codes of one W4M word -> 8 weights in T = the MFMA A fragment of one k-step, bit for bit mx.dequantize's T(fp32(s * q) + b):
// a masked nibble in a 16-bit half IS the f16 denormal q * 2^-24 (bits 0-3) or q * 2^-20 (bits 4-7), v_fma_mix_f32 widens it exactly, the
// product with s24 = s * 2^24 (s20 = s * 2^20; exact scalings) is s * q exactly (<= 15 significant bits), so the fused add of b rounds
// once, exactly as fp32(s * q) + b does.  4 masks + 1 shift + 8 v_fma_mix_f32 + 4 v_cvt_pk: 17 instructions instead of 19-23 (w4m_dequant).
// Domain: |s| < 2^104 (s * 2^24 must not overflow; f16 scales always qualify) -- the tile repack checks it (w4m_repack_launch).
template <class T>
__device__ __forceinline__ uint4 w4r_dequant(u32 w, float s24, float s20, float b) {
    const u32 w8 = w >> 8;
    const u32 t0 = w & 0x000F000Fu, t1 = w & 0x00F000F0u, t2 = w8 & 0x000F000Fu, t3 = w8 & 0x00F000F0u;
    return make_uint4(w4m_pack<T>(w4r_mix_lo(t0, s24, b), w4r_mix_hi(t0, s24, b)), w4m_pack<T>(w4r_mix_lo(t1, s20, b), w4r_mix_hi(t1, s20, b)),
                      w4m_pack<T>(w4r_mix_lo(t2, s24, b), w4r_mix_hi(t2, s24, b)), w4m_pack<T>(w4r_mix_lo(t3, s20, b), w4r_mix_hi(t3, s20, b)));
}
// The same weights by plain (unpacked) instructions: 2 masks + 1 shift, 8 v_cvt_f32_ubyteN, 8 v_fma_f32, 4 v_cvt_pk = 23.  More instructions, but
// beside MFMAs a VOP3P instruction (v_fma_mix_f32, v_pk_fma_f32, v_dot2*) costs the SIMD two to three times a plain one
// (MI355X_MICROARCH.md, cycle constants: "an anti-lever beside MFMAs"): from 128 rows on, where the matrix pipe is the busy unit, this form wins.
template <class T>
__device__ __forceinline__ uint4 w4r_dequant_plain(u32 word, float s, float b) {
    u32 e = word & 0x0F0F0F0Fu, o = (word >> 4) & 0x0F0F0F0Fu;  // bytes: codes (0, 4, 1, 5) and (2, 6, 3, 7)
    asm volatile("" : "+v"(e), "+v"(o));                        // keeps the byte extraction as v_cvt_f32_ubyteN (see w4m_dequant)
    float c[8] = {(float)(e & 0xFFu), (float)((e >> 16) & 0xFFu), (float)(o & 0xFFu), (float)((o >> 16) & 0xFFu),
                  (float)((e >> 8) & 0xFFu), (float)(e >> 24), (float)((o >> 8) & 0xFFu), (float)(o >> 24)};
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r[i] = __builtin_fmaf(s, c[i], b);  // s * q is exact in fp32, so the fused form rounds exactly like fp32(s * q) + b
        asm volatile("" : "+v"(r[i]));      // keeps the SLP vectoriser from pairing them into v_pk_fma_f32
    }
    return make_uint4(w4m_pack<T>(r[0], r[1]), w4m_pack<T>(r[2], r[3]), w4m_pack<T>(r[4], r[5]), w4m_pack<T>(r[6], r[7]));
}

// s_waitcnt vmcnt(N), N <= 63, as a statement no memory operation moves across
#define W4R_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory")

// ---------------- epilogue (both kernel forms): per 32-row block, all phases' partial tiles -> LDS, summed in phase order by all threads
// block (phase, strip ts) at ((phase NS + ts) RED_BLK): lane l's 16 accumulators at l 80 (+ 16 q): conflict-free b128 writes and reads
constexpr int W4R_RED_BLK = 64 * 80;  // one (phase, strip) block: 64 lanes x (16 floats + pad)
template <class T, int WAVES, int MB, int SPW, int KW>
__device__ __forceinline__ void w4r_epilogue(const f32x16_t (&acc)[SPW][MB], char *smem, const W4rArgs &a, const W4mRope &rp, int p, int sg, int lane) {
    constexpr int NS = (WAVES / KW) * SPW, RED_BLK = W4R_RED_BLK;
    const int EPI = a.epi;  // uniform
    static_assert(WAVES * 64 >= 128 * NS, "at most one output octet per reducer thread");
    const int n_strips = a.N >> 5;
    const int t = threadIdx.x, em = t / (4 * NS), ej = t % (4 * NS), ets = ej >> 2, ec = ej & 3;  // reducer thread: row em, strip ets, columns 8 ec .. + 8
    const int ent = blockIdx.x * NS + ets;
    // a compile-time loop: with a run-time `break` in a body of this size the compiler stops unrolling at 8 row blocks, indexes the
    // accumulators dynamically and moves ALL of them to scratch (measured: the 256-row kernel 9 x slower)
    pa_static_for<0, MB>([&](auto mi_c) {
        constexpr int mi = decltype(mi_c)::value;
        if (mi * 32 >= a.M) return;  // uniform: this row block is past the end (so are the later ones)
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            char *blk = smem + (p * NS + sg * SPW + s) * RED_BLK + lane * 80;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(blk + 16 * q) = make_float4(acc[s][mi][4 * q], acc[s][mi][4 * q + 1], acc[s][mi][4 * q + 2], acc[s][mi][4 * q + 3]);
        }
        __syncthreads();
        {  // WAVES x 64 threads = 32 rows x (4 NS) column octets
            const int m = em, row = 32 * mi + m;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0f;
#pragma unroll
            for (int ph = 0; ph < KW; ++ph) {
                const char *blk = smem + (ph * NS + ets) * RED_BLK + 16 * ec;
                const float4 lo = *reinterpret_cast<const float4 *>(blk + m * 80), hi = *reinterpret_cast<const float4 *>(blk + (m + 32) * 80);
                v[0] += lo.x, v[1] += lo.y, v[2] += lo.z, v[3] += lo.w, v[4] += hi.x, v[5] += hi.y, v[6] += hi.z, v[7] += hi.w;
            }
            if (t < 128 * NS && row < a.M && ent < n_strips) {
                const int col = 32 * ent + 8 * ec;  // first of this thread's 8 output columns
                if (EPI == W4R_SLAB) {
                    float *pr = a.part + ((size_t)blockIdx.y * a.M + row) * a.N + col;
                    *reinterpret_cast<float4 *>(pr) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4 *>(pr + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else if (EPI == W4R_ROPE) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) w4m_rope_pair<T>(v[e], v[e + 1], row, col + e, rp);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_T<T>(v[e]);  // the Linear's rounding to T
                    if (a.bias) {
                        const uint4 bb = *reinterpret_cast<const uint4 *>(a.bias + col);
                        const u32 bw[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[2 * e] = round_T<T>(v[2 * e] + lo_f32<T>(bw[e])), v[2 * e + 1] = round_T<T>(v[2 * e + 1] + hi_f32<T>(bw[e]));
                    }
                    if (EPI == W4R_SWIGLU) {  // columns (2 i, 2 i + 1) = (gate_i, up_i): act = T(T(silu(g)) * u)  (language.py:127)
                        u16 o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g = v[2 * e], u = v[2 * e + 1];
                            o[e] = T::from_f32(round_T<T>(g / (1.0f + expf(-g))) * u);
                        }
                        *reinterpret_cast<uint2 *>(a.y + (size_t)row * (a.N >> 1) + (col >> 1)) = make_uint2((u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16));
                    } else {
                        *reinterpret_cast<uint4 *>(a.y + (size_t)row * a.N + col) =
                            make_uint4(w4m_pack<T>(v[0], v[1]), w4m_pack<T>(v[2], v[3]), w4m_pack<T>(v[4], v[5]), w4m_pack<T>(v[6], v[7]));
                    }
                }
            }
        }
        __syncthreads();  // the next row block reuses the blocks
    });
}

// WAVES: waves per workgroup (8: two per SIMD, one workgroup per CU)
// MB: 32-row blocks of x (all M <= 32 MB rows in ONE workgroup: a weight tile is converted once)
// SPW: strips per wave; KW: K-phases (groups per step); XB: x buffers = weight ring slots
// PLAIN: convert with w4r_dequant_plain (matrices whose scales exceed w4r_dequant's domain; rows from which it is no slower)
template <class T, int WAVES, int MB, int SPW, int KW, int XB, bool PLAIN>
__global__ void __launch_bounds__(WAVES * 64) k_w4r_gemm(const W4rArgs a, const W4mRope rp) {
    constexpr int NSW = WAVES / KW, NS = NSW * SPW;       // strip groups; strips per workgroup
    constexpr int ROWB = KW * 128, MT = 32 * MB;              // bytes per staged x row; rows
    constexpr int CHUNK = MT * ROWB;                          // bytes per x buffer
    constexpr int PPR = ROWB / 16, RPI = 64 / PPR;            // 16-byte pieces per row; rows per DMA instruction
    constexpr int XJ = MT / RPI / WAVES;                  // DMA instructions per wave and chunk
    constexpr int WL = 2 * SPW;                               // weight load instructions per wave and step
    constexpr int VMN = WL + (XB - 2) * (XJ + WL);            // loads younger than the DMA of chunk j + 1 at the end of step j
    constexpr int RED_BYTES = KW * NS * W4R_RED_BLK;           // the epilogue's reduction blocks reuse the x buffers
#ifdef W4R_PROF
    constexpr int LDS_BYTES = (XB * CHUNK > RED_BYTES ? XB * CHUNK : RED_BYTES) + WAVES * 32 * 4 * 8;
#else
    constexpr int LDS_BYTES = XB * CHUNK > RED_BYTES ? XB * CHUNK : RED_BYTES;
#endif
    static_assert(MT % (RPI * WAVES) == 0 && XJ >= 1, "x chunk must split evenly over the waves");
    static_assert(VMN <= 63 && XB >= 2, "vmcnt is a 6-bit counter");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kh = lane >> 5;
    const int p = wave % KW, sg = wave / KW;
    const int all_groups = a.K >> 6, n_strips = a.N >> 5;
    const int step0 = blockIdx.y * a.steps;
    const int left = (all_groups + KW - 1) / KW - step0;
    const int nsteps = left < a.steps ? left : a.steps;  // >= 1 (launcher)
    const int g0 = step0 * KW;
    // K / 64 need not be a multiple of KW: the matrix's LAST step then covers its last KW groups (launcher: K / 64 >= KW), of which the first
    // few were multiplied by the step before it -- those phases run it with zero weights (like the dummy steps below)
    const int g_last = all_groups - KW;
    auto g_base = [&](int j) {  // first group of real step j (clamped: never branch around a load)
        const int g = g0 + (j < 0 ? 0 : (j < nsteps ? j : nsteps - 1)) * KW;
        return g < g_last ? g : g_last;
    };

    // this wave's strips: tile (strip, group g0 + j KW + p) of step j
    const char *strip[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int nt = blockIdx.x * NS + sg * SPW + s;  // wave-uniform; a wave without a strip still stages x, joins the barriers and multiplies strip 0 (never stored)
        strip[s] = a.w4m + ((size_t)(nt < n_strips ? nt : 0) * all_groups + p) * W4M_TILE_BYTES;
    }
    typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
    uint4 cw[XB][SPW];
    u32 sb[XB][SPW];
    auto w_issue = [&](int slot, int j) {
        const size_t o = (size_t)g_base(j) * W4M_TILE_BYTES;
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            if (W4R_ABL & 16) {
                cw[slot][s] = make_uint4(0x12345678u + j, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u), sb[slot][s] = 0x3c003c00u;
                continue;
            }
            const nt_u32x4 c = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(strip[s] + o) + lane);
            cw[slot][s] = make_uint4(c.x, c.y, c.z, c.w);
            sb[slot][s] = __builtin_nontemporal_load(reinterpret_cast<const u32 *>(strip[s] + o + 1024) + n);
        }
    };

    // x staging by LDS-DMA: instruction q = wave XJ + jj moves rows [q RPI, (q + 1) RPI) of the chunk, 1 KiB, lane-linear in LDS;
    // lane l -> row r = q RPI + l / PPR, LDS slot l % PPR, which holds SOURCE piece slot ^ (r & 15) (the reads apply the same XOR)
    u32 xoffs[XJ];
#pragma unroll
    for (int jj = 0; jj < XJ; ++jj) {
        const int r = (wave * XJ + jj) * RPI + lane / PPR, sl = lane % PPR;
        const int rr = r < a.M ? r : a.M - 1;  // rows past the end repeat the last one (never stored)
        xoffs[jj] = (u32)(((size_t)rr * a.K + (size_t)((sl ^ (r & 15)) * 8)) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem;
    auto x_issue1 = [&](int j, int buf, int jj) {
        if (W4R_ABL & 4) return;
        const char *src = reinterpret_cast<const char *>(a.x) + (size_t)g_base(j) * 128;
        const unsigned dst = lds0 + (unsigned)(buf * CHUNK + (wave * XJ + jj) * 1024);
        unsigned keep;
        // issued through asm on purpose (see k_w4l2_gemm): through the builtin the compiler drains vmcnt before every ds_read
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(xoffs[jj]), "s"(src), "s"(dst)
                     : "memory");
    };
    auto x_issue = [&](int j, int buf) {
#pragma unroll
        for (int jj = 0; jj < XJ; ++jj) x_issue1(j, buf, jj);
    };
    // B fragment of (row block mi, k-step kk): row 32 mi + n, piece p 8 + 2 kk + kh, swizzled by the row
    int xoff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xoff[kk] = n * ROWB + (((p * 8 + 2 * kk + kh) ^ (n & 15)) << 4);

    // The step loop is unrolled XB-fold with NO control flow inside (with a per-step `if` the compiler's waitcnt pass gives up at the loop
    // header and drains vmcnt to 0 there: the whole look-ahead lost every XB steps), so a workgroup runs nv = pad + nsteps VIRTUAL steps, a
    // multiple of XB; the `pad` dummy steps come FIRST and multiply zeros by zeros (scale = bias = 0 -> A = 0 exactly; their x buffers are
    // zero-filled: no 0 x inf): the sums are bit for bit those of the real steps alone.  Virtual step v = real step v - pad.
    const int pad = (XB - nsteps % XB) % XB, nv = nsteps + pad;
    // prologue in the steady state's queue order: W(0), then [X(i), W(i + 1)] for i < XB - 1
    w_issue(0, 0 - pad);
#pragma unroll
    for (int i = 0; i < XB - 1; ++i) {
        x_issue(i - pad, i);
        w_issue(i + 1, i + 1 - pad);
    }
    f32x16_t acc[SPW][MB];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s][mi][i] = 0.0f;
    if (pad) {  // uniform; the dummy steps' buffers: let their DMAs land, then overwrite them with zeros
        W4R_WAIT_VM(0);
        __syncthreads();
        for (int o = threadIdx.x * 16; o < pad * CHUNK; o += WAVES * 64 * 16) *reinterpret_cast<uint4 *>(smem + o) = make_uint4(0, 0, 0, 0);
    } else {
        W4R_WAIT_VM(VMN);  // X(0) (and W(0)) have landed
    }
    __syncthreads();

    for (int vb = 0; vb < nv; vb += XB) {
#pragma unroll
        for (int d = 0; d < XB; ++d) {
            const int v = vb + d;
            const bool live = v >= pad && g_base(v - pad) + p >= g0 + (v - pad) * KW;  // wave-uniform: not a dummy step, not a group the previous step had
#ifdef W4R_PROF
            if (lane == 0 && v < 32) *reinterpret_cast<unsigned long long *>(smem + XB * CHUNK + ((wave * 32 + v) * 4 + 0) * 8) = __builtin_amdgcn_s_memtime();
#endif
            const char *xb = smem + d * CHUNK;
            float s24[SPW], s20[SPW], bi[SPW], sp[SPW];
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                // each product behind an opaque statement: left to the SLP vectoriser the two scalings become one v_pk_mul_f32 whose operand
                // PAIR includes whatever register sits next to sv -- here the next ring slot's {scale | bias} word, still in flight
                const u32 sbw = live ? sb[d][s] : 0u;  // a dummy step multiplies by zero weights
                float sv = lo_f32<T>(sbw);
                asm volatile("" : "+v"(sv));
                sp[s] = sv;
                s24[s] = sv * 0x1p24f;
                asm volatile("" : "+v"(s24[s]));
                s20[s] = sv * 0x1p20f, bi[s] = hi_f32<T>(sbw);
                asm volatile("" : "+v"(s20[s]), "+v"(bi[s]));
            }
            // NT (k-step, row block) pairs t = kk MB + mi; B fragments ride a register ring PB pairs ahead of their MFMAs.  A vector-memory
            // instruction costs its wave ~60-180 cycles of issue (the CU's one address path takes 1 KiB per instruction at 64 B per clock), so
            // the step's memory issue is SPREAD between the MFMAs -- DMA piece jj after pair XT0 + jj XTS, the weight tiles after the last
            // conversion -- instead of standing in front of them, where both waves of a SIMD would sit in it at the same time.  Queue order per
            // step stays [x DMA x XJ][weights x WL]: the closing counted wait relies on it.
            constexpr int NT = 4 * MB, PB = NT < 4 ? NT : 4;
            constexpr int XTS = (NT - 2) / XJ > 0 ? (NT - 2) / XJ : 1, XT0 = 1;
            auto b_read = [&](int t) {
                const int kk = t / MB, mi = t % MB;
                return (W4R_ABL & 8) ? make_uint4(xoff[kk], mi, kk, d) : *reinterpret_cast<const uint4 *>(xb + mi * 32 * ROWB + xoff[kk]);
            };
            uint4 bq[PB], af[SPW];
#pragma unroll
            for (int t = 0; t < PB; ++t) bq[t] = b_read(t);
            int xj = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int kk = t / MB, mi = t % MB;
                if (mi == 0) {
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        const u32 wk = kk == 0 ? cw[d][s].x : (kk == 1 ? cw[d][s].y : (kk == 2 ? cw[d][s].z : cw[d][s].w));
                        af[s] = (W4R_ABL & 1) ? make_uint4(wk, wk ^ sb[d][s], wk + 1, sb[d][s]) : (PLAIN ? w4r_dequant_plain<T>(wk, sp[s], bi[s]) : w4r_dequant<T>(wk, s24[s], s20[s], bi[s]));
                    }
                }
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    if (W4R_ABL & 2) acc[s][mi][kk] += __builtin_bit_cast(float, af[s].x ^ af[s].y ^ af[s].z ^ af[s].w ^ bq[t % PB].x);
                    else acc[s][mi] = MfmaT<T>::run(af[s], bq[t % PB], acc[s][mi]);
                }
                if (t + PB < NT) bq[t % PB] = b_read(t + PB);
                // never a dummy step's chunk; its buffer was last read in step v - 1, closed by that step's barrier
                if (t >= XT0 && (t - XT0) % XTS == 0 && xj < XJ) x_issue1(v + XB - 1 - pad, (d + XB - 1) % XB, xj++);
                if (t == NT - 1) {
                    for (; xj < XJ; ++xj) x_issue1(v + XB - 1 - pad, (d + XB - 1) % XB, xj);
                    w_issue(d, v + XB - pad);  // slot d held this step's tile (all four words converted above)
                }
            }
#ifdef W4R_PROF
            if (lane == 0 && v < 32) *reinterpret_cast<unsigned long long *>(smem + XB * CHUNK + ((wave * 32 + v) * 4 + 1) * 8) = __builtin_amdgcn_s_memtime();
#endif
            W4R_WAIT_VM(VMN);          // this wave's DMA of chunk v + 1 (and everything older, tile v + 1 included) has landed
#ifdef W4R_PROF
            if (lane == 0 && v < 32) *reinterpret_cast<unsigned long long *>(smem + XB * CHUNK + ((wave * 32 + v) * 4 + 2) * 8) = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();           // ... everyone's has, and everyone is done reading buffer d
#ifdef W4R_PROF
            if (lane == 0 && v < 32) *reinterpret_cast<unsigned long long *>(smem + XB * CHUNK + ((wave * 32 + v) * 4 + 3) * 8) = __builtin_amdgcn_s_memtime();
#endif
            // steps are not interleaved by the scheduler: left alone it hoists later steps' conversions across the barrier, runs to 250
            // registers and re-uses ring registers as ds_read destinations, which drains vmcnt in the middle of a step
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    W4R_WAIT_VM(0);  // the clamped look-ahead DMAs of the last steps target the buffers the epilogue reuses
    __syncthreads();
#ifdef W4R_PROF
    if (a.prof && (blockIdx.x % 50) == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < WAVES * 32 * 4; i += WAVES * 64)
            a.prof[(size_t)(blockIdx.x / 50) * WAVES * 32 * 4 + i] = *reinterpret_cast<unsigned long long *>(smem + XB * CHUNK + i * 8);
    __syncthreads();
#endif
    w4r_epilogue<T, WAVES, MB, SPW, KW>(acc, smem, a, rp, p, sg, lane);
}

