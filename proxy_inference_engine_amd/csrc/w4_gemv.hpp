// w4_gemv.hpp -- the W4S streaming GEMV: device kernel template + launch arguments.
//
// Replaces mx.quantized_matmul(x, w, scales, biases, transpose=True, group_size=64, bits=4) as reached
// through nn.QuantizedLinear at models/llama/language.py:83,108,127,207-209 of the reference, with the
// neighbouring elementwise ops of the decode graph fused into its prologue / epilogue.
//
// HBM-bound (3.6 FLOP/B at M=1): the design goal is bytes in flight, not math.  Measured on MI355X
// (tools/w4s_bench, 66 MB gate/up problem): the bare W4S stream reaches 5.7 TB/s; a workgroup-per-tile kernel
// lost half of that to per-tile activation staging, barriers and epilogues, hence this structure:
//   * PERSISTENT WAVES: the grid is sized to the chip (one 8-wave workgroup per CU); each wave owns a contiguous run of
//     row pairs and streams their units (row pair x 2048-wide K slice, 2304 B) through a register ring that
//     keeps 2 units (3 buffer loads each: 2 x dwordx4 codes + 1 x dword {scale,bias}) in flight; no barrier, no
//     cross-wave reduction inside the loop;
//   * activations: the workgroup stages x ONCE: coalesced 16-byte loads issued BEFORE the weight stream (vmcnt
//     retires in order), fused RMSNorm on 8 elements per thread, then a conflict-free LDS image
//     [8 pieces][groups|1][16 B]; per unit a lane pulls its quantisation group (64 activations) back with
//     8 x ds_read_b128 (LDS traffic = 3.5x the HBM traffic = 14 % of the LDS rate);
//   * dequant: none -- a masked nibble pair (w >> 4i) & 0x000F000F is fed to v_dot2c_f32_bf16 as two bf16
//     DENORMALS (q * 2^-133, honoured exactly by the instruction) against activations pre-scaled by 2^64; per
//     group one fma applies scale (times 2^69) and bias*sum(x).  v_and_or_b32 and v_dot2c are both half-rate on
//     gfx950 (tools/inst_rate), so dropping the exponent OR saves 21 % of the loop's VALU cycles;
//   * a row pair's K slices accumulate in a register; 32-lane DPP reduction; the two row sums are parked in LDS
//     and the epilogue runs once after the stream, one lane per row pair (coalesced stores, vectorised RoPE/SiLU):
//     global stores inside the loop would share vmcnt with the loads and make the compiler drain the ring.
#pragma once
#ifndef PIE_GEMV_NT
#define PIE_GEMV_NT 1  // non-temporal loads on every weight stream (0: only lm_head); A/B in DESIGN.md 2
#endif
#include "attention.hpp"
#include "common.hpp"

// PRO_ATTN: x is the merge of the split-KV attention partials (o_proj input, language.py:107-108)
// PRO_EMBED: x is the embedding row of the step's token, dequantised by every workgroup itself, then RMSNorm as PRO_RMSNORM: the first
//            layer's q|k|v launch, which saves the step its embedding launch (workgroup 0 also leaves the row in the residual stream
//            and the step's RoPE table for the later layers)
enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_ATTN = 2, PRO_EMBED = 3 };
constexpr int GEMV_ATTN_SPLITS = 4;  // split-KV factor the PRO_ATTN prologue merges (register budget: 10 floats per split and piece)
// EPI_PARTIAL_F32: the un-rounded fp32 row sums (row-parallel Linear of a tensor-parallel shard: summed over ranks, THEN rounded)
// EPI_TP_PUSH: the same sums PUSHED straight from the lane's registers into this rank's slot of every peer's receive area, as 8-byte
//              {value, epoch} granules (tp_comm.hip: the one-shot all-reduce's first half; its pull + rounding + residual add is k_tp_pull)
enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_ROPE_KV = 2, EPI_SWIGLU = 3, EPI_LOGITS = 4, EPI_PARTIAL_F32 = 5, EPI_TP_PUSH = 6 };
constexpr int GEMV_TP_MAX_WORLD = 8;

constexpr int GEMV_WAVES = 8;        // waves per workgroup
#ifndef PIE_GEMV_DEPTH
#define PIE_GEMV_DEPTH 2
#endif
constexpr int GEMV_DEPTH = PIE_GEMV_DEPTH;  // units in flight per wave (round 1 swept 2..12: 2-3 best; re-checked with round 2's kernels: 1 / 2 / 3 / 4 -> 1.454 / 1.235 / 1.286 / 1.334 ms per 8B step;
                                            // round 3, the q|k|v launch alone at 3 / 4 with its pairs balanced over the CUs: 1.247 / 1.261 against 1.217)
constexpr int GEMV_MAX_WAVES = 2048; // 256 CUs x ONE 8-wave workgroup: measured best (sweep 1024..6144 in DESIGN.md); the
                                     // activation staging is paid once per CU and no CU runs a second, later wave of groups
constexpr int GEMV_MAX_RUN = 64;     // row pairs per wave (one epilogue lane each)

struct LogitStat {  // per-wave partial of the log-softmax / argmax tail
    float max, sumexp;
    int argmax, pad;
};

struct GemvArgs {
    int fmt;        // FMT_W4S (int4 g=64 units) or FMT_W16S (dense 16-bit units)
    const char *w;  // W4S / W16S
    int n_pairs, n_slices, n_waves, K, N;
    int full_rounds, rem_pairs, n_blocks;  // launcher: n_pairs = full_rounds * n_waves + rem_pairs; workgroups the leftover pairs are dealt over (0: the first waves); see k_w4s_gemv
    const u16 *x;         // [M,K]
    const u16 *norm_w;    // PRO_RMSNORM
    float eps;
    u16 *y;               // EPI_STORE / EPI_LOGITS [M,N]; EPI_SWIGLU act [N/2]
    float *y32;           // EPI_PARTIAL_F32 [M,N] fp32
    unsigned long long *prof;  // developer build (-DPIE_GEMV_PROF): 4 s_memrealtime stamps of workgroup 0 (nullptr otherwise)
    const u16 *lin_bias;  // optional nn.Linear / nn.QuantizedLinear bias [N] in PACKED row order (attention_bias / mlp_bias,
                          // language.py:42-53,117-126): added to the T-rounded product, rounded again; not with EPI_LOGITS
    u16 *resid;           // EPI_RESIDUAL: residual stream, updated in place
    // EPI_ROPE_KV
    const float *freqs;
    const float *rope_cs;  // optional [head_dim/2][2] (cos, sin) of the step's position, filled by the embedding kernel
    const DecState *state;
    u16 *q_out;
    const unsigned long long *kv_table;  // [2*n_layers] device pointers: K buffers then V buffers
    const int *block_table;              // paged KV (nullable): kv_table holds the layers' slab K / V bases, position p lives in
    int n_pages;                         //   page block_table[p / 64] (K then V block, each [n_kv_heads, 64, head_dim]) at row p % 64
    int layer, n_layers, n_heads, n_kv_heads, head_dim;
    int rope_traditional;  // EPI_ROPE_KV: packed q/k rows (2i, 2i+1) are the interleaved pair itself (mx.fast.rope traditional=True)
    LogitStat *stats;     // EPI_LOGITS: one entry per wave of the grid
    const float *part_acc, *part_ml;  // PRO_ATTN: split-KV partials [Hq, splits, D] / [Hq, splits, 2]
    int splits;
    // PRO_EMBED: the int4 g=64 embedding triplet (nn.QuantizedEmbedding, language.py:176), the token, where the row and the RoPE table go
    const u32 *emb_codes;
    const u16 *emb_scales, *emb_biases;
    const int *token;
    int emb_vocab;
    u16 *h_out;
    float *rope_cs_out;
    // EPI_TP_PUSH (tensor-parallel row-parallel Linear): every rank's receive area (device table of world pointers), the communicator's epoch
    // word (this collective is number *tp_epoch + 1), this rank, the world size and the granules per slot
    unsigned long long *const *tp_peers;
    const unsigned *tp_epoch;
    int tp_rank, tp_world;
    unsigned tp_stride;
    // FUSE (EPI_ROPE_KV, the 32 / 8 / 128 head geometry): the step's attention runs behind an XCD-local seam of this launch
    int fuse;             // launcher: take the FUSE instantiation
    AttnArgs attn;        // what k_attn_decode would be launched with
    unsigned *seam;       // per XCD class c: a monotonic arrival counter at [32 c] (+32 per launch); [256]: a spin gave up (never, unless CUs are masked)
};

// out[0..8) = T(scale * q + bias) of one code word: separate multiply and add roundings, like the oracle (mx.dequantize)
template <class T>
__device__ __forceinline__ uint4 dequant_word_u4(u32 word, float s, float b) {
    u32 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float lo = __fadd_rn(__fmul_rn(s, (float)((word >> (8 * j)) & 0xFu)), b);
        const float hi = __fadd_rn(__fmul_rn(s, (float)((word >> (8 * j + 4)) & 0xFu)), b);
        o[j] = pack2<T>(lo, hi);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// LDS carve-up (dynamic, 16-byte aligned): x image | group sums | block reduction scratch | per-wave row sums
struct GemvLds {
    int stride;  // 16-byte slots per piece row: groups | 1 (odd -> conflict-free ds_write_b128 across the 8 pieces)
    int off_sx, off_red, off_out, total;
};
static inline __host__ __device__ GemvLds gemv_lds(int K, bool g32 = false) {  // g32: one activation sum per 32-wide group (FMT_W4S32)
    GemvLds l;
    const int G = K >> 6;
    l.stride = G | 1;
    l.off_sx = 8 * l.stride * 16;
    l.off_red = l.off_sx + (((g32 ? 2 * G : G) * 4 + 15) & ~15);
    l.off_out = l.off_red + 32 * 4;
    l.total = l.off_out + GEMV_WAVES * 2 * GEMV_MAX_RUN * 4;
    return l;
}

template <class T>
__device__ __forceinline__ float w4s_unit_dot(const uint4 &c0, const uint4 &c1, const u32 (&xr)[32]) {
    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // four independent v_dot2c chains (one per code position) for ILP
    const u32 w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) T::dot_word(w[t], xr[4 * t], xr[4 * t + 1], xr[4 * t + 2], xr[4 * t + 3], d);
    return (d[0] + d[2]) + (d[1] + d[3]) * T::ODD_SCALE;
}

// W2S: one 16-byte piece = the lane's 64 two-bit codes; word t covers activation pairs 8 t .. 8 t + 7 (common.hpp: dot_word2).
template <class T>
__device__ __forceinline__ float w2s_unit_dot(const uint4 &c0, const u32 (&xr)[32]) {
    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const u32 w[4] = {c0.x, c0.y, c0.z, c0.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) T::dot_word2(w[t], &xr[8 * t], d);
    return (d[0] + d[2] * T::W2_S2) + (d[1] * T::W2_S1 + d[3] * T::W2_S3);
}

// W4S32: the two code pieces of a lane are two 32-wide groups -- the same chains, summed per piece.
template <class T>
__device__ __forceinline__ void w4s_unit_dot2(const uint4 &c0, const uint4 &c1, const u32 (&xr)[32], float &da, float &db) {
    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const u32 w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) T::dot_word(w[t], xr[4 * t], xr[4 * t + 1], xr[4 * t + 2], xr[4 * t + 3], d);
#pragma unroll
    for (int t = 4; t < 8; ++t) T::dot_word(w[t], xr[4 * t], xr[4 * t + 1], xr[4 * t + 2], xr[4 * t + 3], e);
    da = (d[0] + d[2]) + (d[1] + d[3]) * T::ODD_SCALE;
    db = (e[0] + e[2]) + (e[1] + e[3]) * T::ODD_SCALE;
}

// The same dot product in two steps (several activation rows per code word): operands once, four chains per row -- bit-identical to
// w4s_unit_dot (the same operands enter the same v_dot2c chains in the same order).
template <class T>
__device__ __forceinline__ void w4s_unit_ops(const uint4 &c0, const uint4 &c1, u32 (&e)[32]) {
    const u32 w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        u32 q[4];
        T::word_ops(w[t], q);
        e[4 * t] = q[0], e[4 * t + 1] = q[1], e[4 * t + 2] = q[2], e[4 * t + 3] = q[3];
    }
}
template <class T>
__device__ __forceinline__ float w4s_ops_dot(const u32 (&e)[32], const u32 (&xr)[32]) {
    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = T::dot2(e[4 * t + i], xr[4 * t + i], d[i]);
    }
    return (d[0] + d[2]) + (d[1] + d[3]) * T::ODD_SCALE;
}

// W8S: 16 code words (4 codes each) against the group's 32 packed activation pairs; word i covers pairs 2i, 2i+1.
template <class T>
__device__ __forceinline__ float w8s_unit_dot(const uint4 &c0, const uint4 &c1, const uint4 &c2, const uint4 &c3, const u32 (&xr)[32]) {
    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const u32 w[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        d[(2 * i) & 3] = T::dot2(T::bytes2(w[i] & 0x00FF00FFu), xr[2 * i], d[(2 * i) & 3]);
        d[(2 * i + 1) & 3] = T::dot2(T::bytes2((w[i] >> 8) & 0x00FF00FFu), xr[2 * i + 1], d[(2 * i + 1) & 3]);
    }
    return (d[0] + d[2]) + (d[1] + d[3]);
}

// W8S32: code pieces (0, 1) and (2, 3) are two 32-wide groups.
template <class T>
__device__ __forceinline__ void w8s_unit_dot2(const uint4 &c0, const uint4 &c1, const uint4 &c2, const uint4 &c3, const u32 (&xr)[32], float &da, float &db) {
    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const u32 w[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d[(2 * i) & 3] = T::dot2(T::bytes2(w[i] & 0x00FF00FFu), xr[2 * i], d[(2 * i) & 3]);
        d[(2 * i + 1) & 3] = T::dot2(T::bytes2((w[i] >> 8) & 0x00FF00FFu), xr[2 * i + 1], d[(2 * i + 1) & 3]);
    }
#pragma unroll
    for (int i = 8; i < 16; ++i) {
        e[(2 * i) & 3] = T::dot2(T::bytes2(w[i] & 0x00FF00FFu), xr[2 * i], e[(2 * i) & 3]);
        e[(2 * i + 1) & 3] = T::dot2(T::bytes2((w[i] >> 8) & 0x00FF00FFu), xr[2 * i + 1], e[(2 * i + 1) & 3]);
    }
    da = (d[0] + d[2]) + (d[1] + d[3]);
    db = (e[0] + e[2]) + (e[1] + e[3]);
}

// 8 packed activations times the trait's exact power-of-two pre-scale (identity for f16)
template <class T>
__device__ __forceinline__ uint4 scale8(const uint4 &v) {
    if (T::XSCALE == 1.0f) return v;
    return make_uint4(pack2<T>(lo_f32<T>(v.x) * T::XSCALE, hi_f32<T>(v.x) * T::XSCALE), pack2<T>(lo_f32<T>(v.y) * T::XSCALE, hi_f32<T>(v.y) * T::XSCALE),
                      pack2<T>(lo_f32<T>(v.z) * T::XSCALE, hi_f32<T>(v.z) * T::XSCALE), pack2<T>(lo_f32<T>(v.w) * T::XSCALE, hi_f32<T>(v.w) * T::XSCALE));
}

template <class T>
__device__ __forceinline__ float sum8(const uint4 &v) {
    return ((lo_f32<T>(v.x) + hi_f32<T>(v.x)) + (lo_f32<T>(v.y) + hi_f32<T>(v.y))) +
           ((lo_f32<T>(v.z) + hi_f32<T>(v.z)) + (lo_f32<T>(v.w) + hi_f32<T>(v.w)));
}

// A wave-uniform pointer made opaque to the compiler at this point: it has to be in SGPRs here and can only be kept (or spilled to a
// VGPR lane) afterwards, never re-fetched from the kernel-argument segment later.
template <class U>
__device__ __forceinline__ U *gemv_pin(U *p) {
    unsigned long long v = reinterpret_cast<unsigned long long>(p);
    asm volatile("" : "+s"(v));
    return reinterpret_cast<U *>(v);
}

__device__ __forceinline__ unsigned gemv_pin_u32(unsigned v) {
    v = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(v));
    return v;
}

__device__ __forceinline__ float lane_value(float v, int lane) {  // wave-uniform broadcast of one lane (v_readlane)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// ABL: developer ablation switches for tools/w4s_bench (0 in every product instantiation):
//   1 = no weight loads, 2 = no dot products, 4 = no activation staging / LDS reads
// NPT: activation pieces (8 elements) per thread of the staging pass, ceil(K/8/512) rounded up to 1, 2, 4 or 8.
// FMT: weight format of the stream.  FMT_W16S serves dense checkpoints (nn.Linear, language.py:83,108,127,209 when the
//      config has no "quantization" entry): same persistent-wave stream, prologues and epilogues; a unit is 2 x 1 KB of
//      16-bit weights, a lane multiplies its 16 weights with 16 activations (8 v_dot2), no scale/bias.
#ifdef PIE_GEMV_PROF
// stamps 0..3: workgroup 0; with PIE_GEMV_PROF=2 also every workgroup's start / end at prof[64 + 2 b], prof[65 + 2 b] (gate/up only: its prof slot is the last)
#define GEMV_STAMP(i)                                                                                                   \
    do {                                                                                                                \
        if (a.prof && blockIdx.y == 0 && threadIdx.x == 0) {                                                            \
            if (blockIdx.x == 0) a.prof[i] = __builtin_amdgcn_s_memrealtime();                                          \
            if (PIE_GEMV_PROF == 2 && EPI == EPI_SWIGLU && ((i) == 0 || (i) == 3)) a.prof[64 + 2 * blockIdx.x + ((i) == 3)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                               \
    } while (0)
#else
#define GEMV_STAMP(i)
#endif

// FUSE (round 5; q|k|v of a model with 32 query heads, 8 kv heads of 128: Llama-3-8B, Mistral-7B): the rows of kv-group g -- its 4 q heads, its
// k and its v head -- are computed by the 32 workgroups with blockIdx.x % 8 == g, which the dispatcher places on ONE XCD (tools/pilot_probe,
// checked again at decoder creation), and the step's attention for that group runs in the same launch: stores are write-through to the XCD's
// L2, every workgroup adds 1 to its XCD's counter (an atomic performed AT that L2: no release fence; the counter only grows, 32 per
// launch), and the group's attention workgroups poll it with sc1 loads and then run k_attn_decode's body (no acquire fence: the CU's L1
// was invalidated at launch and has not seen these lines).  tools/seam_probe: 0.12 us from the last arrival to the release, 0.96 us until 32 KB
// of the group's fresh data are read -- against 3.9 us for a dependent launch.  The other 28 workgroups of the XCD warm the Infinity Cache
// with o_proj's weights (what the attention launch's idle workgroups did).  Same rows, same units, same order: bit-identical to two launches.
template <class T, int PRO, int EPI, int NPT, int ABL = 0, int FMT = FMT_W4S, int FUSE = 0>
__global__ void __launch_bounds__(GEMV_WAVES * 64, 4) k_w4s_gemv(const GemvArgs a) {
#if defined(PIE_GEMV_PROF) && PIE_GEMV_PROF == 3
    // fine stamps of the prologue (workgroup 0, thread 0; stored at the very end at prof[8..12]): kernel entry | first kernel argument usable |
    // activations arrived | norm reduced (first barrier) | image staged
    const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
    {
        int k_probe = a.K;
        asm volatile("" ::"s"(k_probe));
    }
    const unsigned long long ts_args = __builtin_amdgcn_s_memrealtime();
    unsigned long long ts_x = 0, ts_b1 = 0, ts_staged = 0;
#endif
    GEMV_STAMP(0);
    constexpr int D = GEMV_DEPTH;  // also for the 1280-byte W2S units: twice the depth (the W4S bytes in flight) measured 1.007 vs 0.993 ms per 2-bit 8B step
    constexpr int UB = fmt_unit_bytes(FMT);
    constexpr bool G32 = FMT == FMT_W4S32 || FMT == FMT_W8S32;  // two 32-wide groups per lane
    constexpr bool W8 = FMT == FMT_W8S || FMT == FMT_W8S32;      // byte codes: four pieces per lane
    constexpr int NT = GEMV_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ns = a.n_slices;
    const int m = blockIdx.y;
    const GemvLds L = gemv_lds(a.K, FMT == FMT_W4S32 || FMT == FMT_W8S32);
    float *sxs = reinterpret_cast<float *>(smem + L.off_sx);
    float *red = reinterpret_cast<float *>(smem + L.off_red);
    float *outp = reinterpret_cast<float *>(smem + L.off_out) + wave * (2 * GEMV_MAX_RUN);  // this wave's row sums

    // this wave's row pairs: gw, gw + W, gw + 2W, ...  Interleaving keeps the set of units in flight across the chip
    // a compact window that sweeps through the matrix (like a non-persistent grid would), instead of 4096 distant
    // streams -- measured faster on HBM (tools/w4s_bench).
    // FUSE: physical workgroup 8 j + g works as logical workgroup 32 g + j, so group g's q pairs (logical workgroups 32 g .. 32 g + 31) stay on XCD class g
    const int bx = FUSE ? (int)((blockIdx.x & 7) * 32 + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int gw = bx * GEMV_WAVES + wave;
    const int W = a.n_waves;
    // the XCD's arrival counter only ever grows, by 32 per launch: whatever this workgroup reads here (before it arrives) lies in [32 k, 32 k + 31], and the
    // launch is complete at 32 (k + 1)
    unsigned seam_target = 0;
    if (FUSE && threadIdx.x == 0) seam_target = (__hip_atomic_load(a.seam + 32 * (blockIdx.x & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~31u) + 32u;
    // The pairs that do not fill a whole round of W waves are dealt over the WORKGROUPS, not over the first waves: q|k|v of the 8B
    // model is 3072 pairs on 2048 waves, and with the leftover 1024 on waves 0..1023 half of the CUs streamed twice what the other half
    // did (a launch lasts as long as its busiest CU's ingest).  Leftover pair r goes to wave r / n_blocks of workgroup r % n_blocks.
    const int kf = a.full_rounds;
    int rem_r = a.n_blocks > 0 ? wave * a.n_blocks + (int)blockIdx.x : gw;  // this wave's leftover pair, if r < rem_pairs (n_blocks 0: dealt over the first waves)
    if (FUSE) {  // the 1024 leftover pairs are the k rows (0 .. 511) and the v rows (512 ..): group g's 64 + 64 go to its own 32 workgroups, waves 0 .. 3
        const int u = wave * 32 + (int)(blockIdx.x >> 3), g8 = (int)(blockIdx.x & 7);
        rem_r = wave < 4 ? (u < 64 ? 64 * g8 + u : 512 + 64 * g8 + (u - 64)) : a.rem_pairs;
    }
    const bool has_rem = rem_r < a.rem_pairs;
    const int run = (gw < W ? kf : 0) + (has_rem ? 1 : 0);       // <= GEMV_MAX_RUN (host-checked)
    const int last_pair = kf * W + rem_r;                        // local pair kf, when has_rem
    const int n_units = run * ns;
    const size_t pstride = (size_t)W * ns * UB;  // bytes between consecutive pairs of this wave

    // 1. activations first (coalesced, 8 elements per piece), then the head of the weight stream.
    const int n_pieces = a.K >> 3;
    const uint4 *xg = reinterpret_cast<const uint4 *>(a.x + (size_t)m * a.K);
    uint4 xv[NPT], nv[NPT];
    AttnMergeRegs<GEMV_ATTN_SPLITS> mr[PRO == PRO_ATTN ? NPT : 1];
    int attn_active = 1;
    if (PRO == PRO_ATTN) {
        attn_active = a.splits;  // idle splits hold the neutral partial (k_attn_decode): no wait for the position here
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            int j = threadIdx.x + i * NT;
            j = j < n_pieces ? j : n_pieces - 1;
            const int ppd = a.head_dim >> 3;  // pieces per head
            attn_merge_load<GEMV_ATTN_SPLITS>(a.part_acc, a.part_ml, a.splits, attn_active, j / ppd, a.head_dim, (j % ppd) * 8, mr[i]);
        }
    } else if (!(ABL & 4)) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            int j = threadIdx.x + i * NT;
            j = j < n_pieces ? j : n_pieces - 1;  // clamp, never branch around a load
            if (PRO != PRO_EMBED) xv[i] = xg[j];
            if (PRO == PRO_RMSNORM || PRO == PRO_EMBED) nv[i] = reinterpret_cast<const uint4 *>(a.norm_w)[j];
        }
    }
    // epilogue-side operands, fetched now so their latency hides under the stream (lane l owns pair gw + l*W)
    // ... including the kernel ARGUMENTS only the epilogue uses: hipcc sinks their s_load to the first use, i.e. behind the stream, where
    // a scalar-cache miss costs several hundred cycles while every CU streams (gate|up: `s_load_dwordx2 .., 0x48` twenty instructions ahead
    // of its store; lm_head: two of them).  Pinned in SGPRs here.
    u16 *y_out = a.y + (size_t)m * a.N;
    float *y32_out = a.y32 + (size_t)m * a.N;
    LogitStat *stats_out = a.stats + (size_t)m * a.n_waves;
    u16 *q_dst = a.q_out;
    if (EPI == EPI_STORE || EPI == EPI_LOGITS || EPI == EPI_SWIGLU) y_out = gemv_pin(y_out);
    if (EPI == EPI_PARTIAL_F32) y32_out = gemv_pin(y32_out);
    if (EPI == EPI_LOGITS) stats_out = gemv_pin(stats_out);
    if (EPI == EPI_ROPE_KV) q_dst = gemv_pin(q_dst);
    // EPI_TP_PUSH: the peers' areas and the collective's number, in SGPRs before the stream (the pushes then leave straight from the epilogue)
    unsigned long long *tp_dst[EPI == EPI_TP_PUSH ? GEMV_TP_MAX_WORLD : 1];
    unsigned tp_e = 0;
    if constexpr (EPI == EPI_TP_PUSH) {
        tp_e = gemv_pin_u32(*a.tp_epoch + 1u);
        const size_t slot = ((size_t)(tp_e & 1u) * a.tp_world + a.tp_rank) * a.tp_stride;
#pragma unroll
        for (int r = 0; r < GEMV_TP_MAX_WORLD; ++r) {
            const unsigned long long p = reinterpret_cast<unsigned long long>(a.tp_peers[r < a.tp_world ? r : 0] + slot);  // wave-uniform, but loaded through a pointer: VGPRs to hipcc
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
            tp_dst[r] = gemv_pin(reinterpret_cast<unsigned long long *>(((unsigned long long)hi << 32) | lo));
        }
    }
    const bool live = lane < run;
    const int pair = lane < kf ? gw + lane * W : last_pair;
    const int R = 2 * pair;  // packed row index; the pair is rows R, R+1
    u32 pre_u = 0;           // EPI_RESIDUAL: the residual pair
    u32 pre_b = 0;           // the linear bias pair
    float pre_cs = 1.0f, pre_sn = 0.0f;
    int pos = 0, cap = 0, kvrow = 0;
    u16 *kdst = nullptr, *vdst = nullptr;
    if (EPI == EPI_RESIDUAL && live) pre_u = *reinterpret_cast<const u32 *>(a.resid + R);
    const bool has_bias = EPI != EPI_LOGITS && EPI != EPI_PARTIAL_F32 && EPI != EPI_TP_PUSH && a.lin_bias != nullptr;  // wave-uniform
    if (has_bias && live) pre_b = *reinterpret_cast<const u32 *>(a.lin_bias + R);
    if (EPI == EPI_ROPE_KV) {
        pos = a.state->pos, cap = a.state->cap;
        kdst = reinterpret_cast<u16 *>(a.kv_table[a.layer]);
        vdst = reinterpret_cast<u16 *>(a.kv_table[a.n_layers + a.layer]);
        kvrow = pos;
        if (a.block_table) {  // paged: the page of this position is one more "buffer" of 64 rows per head
            const size_t pg_off = (size_t)min((unsigned)a.block_table[pos >> 6], (unsigned)a.n_pages - 1u) * 2 * 64 * a.n_kv_heads * a.head_dim;
            kdst += pg_off, vdst += pg_off, cap = 64, kvrow = pos & 63;
        }
        if (live && R < (a.n_heads + a.n_kv_heads) * a.head_dim) {
            const int ii = (R % a.head_dim) >> 1;
            if (a.rope_cs) pre_cs = a.rope_cs[2 * ii], pre_sn = a.rope_cs[2 * ii + 1];
            else sincosf((float)pos * (1.0f / a.freqs[ii]), &pre_sn, &pre_cs);
        }
    }
    uint4 c0[D], c1[D];  // W2S: one code piece per lane (c1 stays unused)
    uint4 c2[(W8 || FMT == FMT_W6S) ? D : 1], c3[W8 ? D : 1];  // W6S: c2 = the high-bit plane  // W8S: a lane's 64 codes are four pieces
    u32 sb[D];
    u32 sb2[G32 ? D : 1];  // W4S32 / W8S32: the second 32-wide group's {scale | bias << 16}
    // Weight loads go through a buffer descriptor over the whole matrix: a ring slot that has no unit left to fetch is
    // given an out-of-range offset, which the hardware bounds check drops (no memory traffic, no branch around a load).
    // (Clamping to the last unit instead cost up to D redundant loads per wave: 70 % extra at 7-unit runs.)
    typedef __attribute__((ext_vector_type(4))) u32 u32x4_t;
    const unsigned w_bytes = (unsigned)((size_t)a.n_pairs * ns * UB);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.w), 0, (int)w_bytes, 0x00020000);
    const unsigned woff0 = (unsigned)((size_t)gw * ns * UB) + lane * 16;  // this wave's first unit, this lane's piece
    const unsigned woff_last = (unsigned)((size_t)last_pair * ns * UB) + lane * 16;
    const unsigned pstride32 = (unsigned)pstride;
    int iss_sl = 0, iss_pl = 0;  // slice / local pair of the next unit to issue
    // K that is not a multiple of the slice width leaves zero-padded lanes in every row's last unit (Llama-3.2-3B: H = 3072 ->
    // 1.5 slices, a third more bytes than the checkpoint holds).  Those lanes get the out-of-range offset too: the padding is
    // 128-byte aligned runs of 8+ lanes, so whole cache lines are never fetched; the lane then computes on zeros, as before.
    const int my_chunks = FMT == FMT_W16S ? (a.K + 15) >> 4 : (a.K + 63) >> 6;  // valid per-lane column chunks (16 weights / one 64-group)
    const bool ragged = (my_chunks & 31) != 0;                                    // wave-uniform
    auto issue = [&](int d, int) {  // ring slot d <- next unit of this wave
        unsigned off = iss_pl < run ? (iss_pl < kf ? woff0 + (unsigned)iss_pl * pstride32 : woff_last) + (unsigned)iss_sl * UB : 0xFFFFF000u;
        if (ragged && iss_sl * 32 + (lane & 31) >= my_chunks) off = 0xFFFFF000u;
        if (++iss_sl == ns) iss_sl = 0, ++iss_pl;
        if (ABL & 1) {
            c0[d] = c1[d] = make_uint4(lane, off, d, 7);
            sb[d] = 0x3c003c00u;
        } else {
            // Weights are read once per step and a step streams 4.2 GB through the 256 MiB Infinity Cache: non-temporal
            // loads (aux = 2).  lm_head alone: 53.6 -> 48.3 us.  The per-layer matrices looked 8-15 % SLOWER with nt when one
            // kernel was replayed back to back (its <= 66 MB then hit the cache); inside the real step, where nothing is ever
            // re-read, nt on every stream measured 761 -> 780 tok/s (two A/B rounds in one session), gate/up 13.1 -> 12.7 us
            constexpr int AUX = (EPI == EPI_LOGITS || PIE_GEMV_NT) ? 2 : 0;
            const u32x4_t v0 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, AUX);
            c0[d] = make_uint4(v0.x, v0.y, v0.z, v0.w);
            if constexpr (FMT == FMT_W2S) {  // one code piece per lane, then the lane's {scale | bias << 16}
                sb[d] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, off + 1024 - lane * 12, 0, AUX);
            } else {
                const u32x4_t v1 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + 1024, 0, AUX);
                c1[d] = make_uint4(v1.x, v1.y, v1.z, v1.w);
            }
            if (FMT == FMT_W4S) sb[d] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, off + 2048 - lane * 12, 0, AUX);
            if constexpr (FMT == FMT_W6S) {  // the high two bits of the lane's 64 codes (W2S word order), then {scale | bias << 16}
                const u32x4_t v2 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + 2048, 0, AUX);
                c2[d] = make_uint4(v2.x, v2.y, v2.z, v2.w);
                sb[d] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, off + 3072 - lane * 12, 0, AUX);
            }
            if (FMT == FMT_W4S32) {
                typedef __attribute__((ext_vector_type(2))) u32 u32x2_t;
                const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(wrsrc, off + 2048 - lane * 8, 0, AUX);
                sb[d] = v.x, sb2[d] = v.y;
            }
            if (W8) {
                const u32x4_t v2 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + 2048, 0, AUX);
                const u32x4_t v3 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + 3072, 0, AUX);
                c2[d] = make_uint4(v2.x, v2.y, v2.z, v2.w);
                c3[d] = make_uint4(v3.x, v3.y, v3.z, v3.w);
                if (FMT == FMT_W8S) sb[d] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, off + 4096 - lane * 12, 0, AUX);
                if (FMT == FMT_W8S32) {
                    typedef __attribute__((ext_vector_type(2))) u32 u32x2_t;
                    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(wrsrc, off + 4096 - lane * 8, 0, AUX);
                    sb[d] = v.x, sb2[d] = v.y;
                }
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    if constexpr (PRO == PRO_EMBED) {  // behind the head of the stream: the row's address hangs on a scalar load of the token
        int id = *a.token;
        id = id < 0 ? 0 : (id >= a.emb_vocab ? a.emb_vocab - 1 : id);
        const u32 *row = a.emb_codes + (size_t)id * n_pieces;
        const u16 *srow = a.emb_scales + (size_t)id * (a.K >> 6), *brow = a.emb_biases + (size_t)id * (a.K >> 6);
        u32 wd[NPT];
        u16 sc[NPT], bi[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            int j = threadIdx.x + i * NT;
            j = j < n_pieces ? j : n_pieces - 1;
            wd[i] = row[j], sc[i] = srow[j >> 3], bi[i] = brow[j >> 3];
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            xv[i] = dequant_word_u4<T>(wd[i], T::to_f32(sc[i]), T::to_f32(bi[i]));
            const int j = threadIdx.x + i * NT;
            if (blockIdx.x == 0 && j < n_pieces) reinterpret_cast<uint4 *>(a.h_out)[j] = xv[i];  // h = embed_tokens(inputs): the residual stream of o_proj's epilogue
        }
        if (blockIdx.x == 0 && (int)threadIdx.x < (a.head_dim >> 1)) {  // cos / sin of pos / freqs[i] for the later layers' q|k|v epilogues
            float sn, cs;
            sincosf((float)a.state->pos * (1.0f / a.freqs[threadIdx.x]), &sn, &cs);
            a.rope_cs_out[2 * threadIdx.x] = cs, a.rope_cs_out[2 * threadIdx.x + 1] = sn;
        }
    }

    // 2. stage x through LDS once per workgroup, with the fused mx.fast.rms_norm
    //    (nn.RMSNorm, language.py:137-141,168): w * T(x * rsqrt(mean(x^2) + eps)), or the split-KV merge.
#if defined(PIE_GEMV_PROF) && PIE_GEMV_PROF == 3
    if (PRO != PRO_ATTN && PRO != PRO_EMBED) {
        asm volatile("" ::"v"(xv[0].x));
        ts_x = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (!(ABL & 4)) {
        if (PRO == PRO_ATTN) {
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                float o[8];
                attn_merge_finish<GEMV_ATTN_SPLITS>(mr[i], attn_active, o);
                xv[i] = make_uint4(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7]));
            }
        }
        if (PRO == PRO_RMSNORM || PRO == PRO_EMBED) {
            float ssq = 0.0f;
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const bool ok = threadIdx.x + i * NT < n_pieces;
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
            __syncthreads();
#if defined(PIE_GEMV_PROF) && PIE_GEMV_PROF == 3
            ts_b1 = __builtin_amdgcn_s_memrealtime();
#endif
            const float4 ra = *reinterpret_cast<const float4 *>(red), rb = *reinterpret_cast<const float4 *>(red + 4);
            const float tot = ((ra.x + ra.y) + (ra.z + ra.w)) + ((rb.x + rb.y) + (rb.z + rb.w));
            const float inv = 1.0f / sqrtf(tot / (float)a.K + a.eps);
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const u32 v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, g[4] = {nv[i].x, nv[i].y, nv[i].z, nv[i].w};
                u32 o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = pack2<T>(round_T<T>(lo_f32<T>(v[k]) * inv) * lo_f32<T>(g[k]), round_T<T>(hi_f32<T>(v[k]) * inv) * hi_f32<T>(g[k]));
                xv[i] = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        // publish: piece j = 8*group + r lands at slot (r*stride + group); group sums via DPP over 8 lanes
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int j = threadIdx.x + i * NT;
            const bool ok = j < n_pieces;
            float ps = ok ? sum8<T>(xv[i]) : 0.0f;
            ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
            if (G32) {  // one sum per 32-wide group: lanes 4k .. 4k+3 (all hold it after the two quad steps)
                if (ok && (j & 3) == 0) sxs[j >> 2] = ps;
            }
            ps += __builtin_amdgcn_update_dpp(0.0f, ps, 0x141, 0xF, 0xF, true);  // row_half_mirror: 8-lane sums
            if (ok) {
                *reinterpret_cast<uint4 *>(smem + ((size_t)(j & 7) * L.stride + (j >> 3)) * 16) = FMT != FMT_W16S ? scale8<T>(xv[i]) : xv[i];
                if (!G32 && (j & 7) == 0) sxs[j >> 3] = ps;
            }
        }
        __syncthreads();
    }

    GEMV_STAMP(1);  // x staged
#if defined(PIE_GEMV_PROF) && PIE_GEMV_PROF == 3
    ts_staged = __builtin_amdgcn_s_memrealtime();
#endif
    // 3. the stream: unit i = (pair p_begin + i / ns, slice i % ns); lanes 0-31 / 32-63 = the pair's two rows.
    //    No global store and no data-dependent branch in here: stores share vmcnt with the loads and would make the
    //    compiler drain the ring.  Row sums are parked in LDS; the epilogue runs after the loop, one lane per pair.
    const int n_groups = a.K >> 6;
    float acc = 0.0f;           // this lane's share of the current row, across slices
    int sl = 0, pl = 0;         // slice / local pair of the unit being consumed
    for (int base = 0; base < n_units; base += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = base + d;
            if (i < n_units) {  // wave-uniform
              // The two waves of a SIMD take turns at the higher issue priority, two units each.  With equal priorities the older wave of
              // every SIMD wins the arbitration: stamps showed waves 0-3 of a gate/up workgroup finishing their (equal) share of the
              // stream 0.8-1.0 us before waves 4-7, and the workgroup -- and the launch -- waiting for the late half.  Turns of one
              // unit, a wider priority gap or the younger wave first measured within noise of this; the younger wave ALWAYS ahead was
              // slower than no priorities (1.262 vs 1.243 ms per step); this form: 1.227-1.232.
#ifndef PIE_GEMV_NO_PRIO
              if (((i >> 1) + (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
              else __builtin_amdgcn_s_setprio(0);
#endif
              if (FMT == FMT_W16S) {  // lane (row, chunk): 16 weights x 16 activations
                const int chunk = sl * 32 + (lane & 31);
                const bool cvalid = chunk * 16 < a.K;
                const int cc = cvalid ? chunk : 0;
                const uint4 x0 = *reinterpret_cast<const uint4 *>(smem + ((size_t)((cc & 3) * 2) * L.stride + (cc >> 2)) * 16);
                const uint4 x1 = *reinterpret_cast<const uint4 *>(smem + ((size_t)((cc & 3) * 2 + 1) * L.stride + (cc >> 2)) * 16);
                float d0 = T::dot2(c0[d].x, x0.x, 0.0f), d1 = T::dot2(c0[d].y, x0.y, 0.0f);
                d0 = T::dot2(c0[d].z, x0.z, d0), d1 = T::dot2(c0[d].w, x0.w, d1);
                d0 = T::dot2(c1[d].x, x1.x, d0), d1 = T::dot2(c1[d].y, x1.y, d1);
                d0 = T::dot2(c1[d].z, x1.z, d0), d1 = T::dot2(c1[d].w, x1.w, d1);
                acc += cvalid ? d0 + d1 : 0.0f;
              } else {
                const int g = sl * 32 + (lane & 31);
                const bool gvalid = g < n_groups;
                const int gc = gvalid ? g : n_groups - 1;
                u32 xr[32];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint4 v = (ABL & 4) ? make_uint4(0x3f803f80u + r, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u)
                                              : *reinterpret_cast<const uint4 *>(smem + ((size_t)r * L.stride + gc) * 16);
                    xr[4 * r + 0] = v.x, xr[4 * r + 1] = v.y, xr[4 * r + 2] = v.z, xr[4 * r + 3] = v.w;
                }
                if constexpr (G32) {
                    float da, db;
                    if constexpr (FMT == FMT_W8S32) w8s_unit_dot2<T>(c0[d], c1[d], c2[d], c3[d], xr, da, db);
                    else w4s_unit_dot2<T>(c0[d], c1[d], xr, da, db);
                    const float2 sx2 = *reinterpret_cast<const float2 *>(sxs + 2 * gc);
                    const float pa = fmaf(lo_f32<T>(sb[d]), da * T::DSCALE - T::OFFSET * sx2.x, hi_f32<T>(sb[d]) * sx2.x);
                    const float pb = fmaf(lo_f32<T>(sb2[d]), db * T::DSCALE - T::OFFSET * sx2.y, hi_f32<T>(sb2[d]) * sx2.y);
                    acc += gvalid ? pa + pb : 0.0f;
                } else {
                const float sx = (ABL & 4) ? 64.0f : sxs[gc];
                float dd;
                if (FMT == FMT_W8S) dd = w8s_unit_dot<T>(c0[d], c1[d], c2[d], c3[d], xr);
                else if (FMT == FMT_W2S) dd = (ABL & 2) ? __builtin_bit_cast(float, c0[d].x ^ c0[d].y ^ c0[d].z ^ c0[d].w ^ xr[d] ^ xr[8 + d] ^ xr[16 + d] ^ xr[24 + d]) : w2s_unit_dot<T>(c0[d], xr);
                else if (FMT == FMT_W6S) dd = w4s_unit_dot<T>(c0[d], c1[d], xr) + 16.0f * w2s_unit_dot<T>(c2[d], xr);  // q = lo + 16 hi
                else dd = (ABL & 2) ? __builtin_bit_cast(float, c0[d].x ^ c0[d].y ^ c0[d].z ^ c0[d].w ^ c1[d].x ^ c1[d].y ^ c1[d].z ^ c1[d].w ^ xr[d])
                                    : w4s_unit_dot<T>(c0[d], c1[d], xr);
                const float scale = lo_f32<T>(sb[d]), bias = hi_f32<T>(sb[d]);
                const float pr = fmaf(scale, dd * T::DSCALE - T::OFFSET * sx, bias * sx);
                acc += gvalid ? pr : 0.0f;  // padded groups carry zero codes and zero {scale,bias}; the select keeps a NaN x out
                }
              }
                if (++sl == ns) {
                    const float tot = half_wave_sum(acc);
                    if ((lane & 31) == 31) outp[2 * pl + (lane >> 5)] = tot;
                    acc = 0.0f, sl = 0, ++pl;
                }
            }
            issue(d, i + D);
        }
    }

#if defined(PIE_GEMV_PROF) && PIE_GEMV_PROF == 2
    if (a.prof && EPI == EPI_SWIGLU && blockIdx.y == 0 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 201))
        a.prof[32 + (blockIdx.x == 0 ? 0 : (blockIdx.x == 100 ? 8 : 16)) + wave] = __builtin_amdgcn_s_memrealtime();  // every wave's stream end
#endif
    GEMV_STAMP(2);  // stream done
    // 4. epilogue: lane l owns local pair l (rows R, R+1 of the packed order); consecutive lanes -> consecutive addresses.
    float va = 0.0f, vb = 0.0f;
    if (live) {
        const float2 o = *reinterpret_cast<const float2 *>(outp + 2 * lane);
        va = o.x, vb = o.y;
    }
    if (has_bias && EPI != EPI_STORE) {  // y = T(T(x W^T) + b): fold the first rounding and the add here, the epilogues round again
        va = round_T<T>(va) + lo_f32<T>(pre_b);
        vb = round_T<T>(vb) + hi_f32<T>(pre_b);
    }
    if (EPI == EPI_STORE || EPI == EPI_LOGITS) {
        float oa = round_T<T>(va), ob = round_T<T>(vb);
        if (EPI == EPI_STORE && has_bias && live) {
            oa = round_T<T>(oa + lo_f32<T>(pre_b));
            ob = round_T<T>(ob + hi_f32<T>(pre_b));
        }
        if (live) *reinterpret_cast<u32 *>(y_out + R) = pack2<T>(oa, ob);
        if (EPI == EPI_LOGITS) {  // per-wave log-softmax partial: max, first argmax, sum exp(x - max)
            const float mx = live ? fmaxf(oa, ob) : -INFINITY;
            const int ix = live ? (ob > oa ? R + 1 : R) : 0x7fffffff;
            const float wmax = wave_max(mx);
            int cand = (live && mx == wmax) ? ix : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
            float se = live ? expf(oa - wmax) + expf(ob - wmax) : 0.0f;
            se = wave_sum(se);
            if (lane == 0 && gw < a.n_waves) {
                LogitStat st;
                st.max = wmax, st.sumexp = se, st.argmax = cand, st.pad = 0;
                stats_out[gw] = st;
            }
        }
    } else if (EPI == EPI_PARTIAL_F32) {
        if (live) *reinterpret_cast<float2 *>(y32_out + R) = make_float2(va, vb);
    } else if constexpr (EPI == EPI_TP_PUSH) {
        // one system-scope 8-byte store per element and peer: the data is its own flag (tp_comm.hip); rows R, R + 1 are adjacent granules
        if (live) {
            typedef __attribute__((address_space(1))) unsigned long long gu64_t;
            const unsigned long long ga = ((unsigned long long)tp_e << 32) | __builtin_bit_cast(unsigned, va), gb = ((unsigned long long)tp_e << 32) | __builtin_bit_cast(unsigned, vb);
#pragma unroll
            for (int r = 0; r < GEMV_TP_MAX_WORLD; ++r)
                if (r < a.tp_world) {
                    __hip_atomic_store((gu64_t *)(unsigned long long)(tp_dst[r] + R), ga, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store((gu64_t *)(unsigned long long)(tp_dst[r] + R + 1), gb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
        }
    } else if (EPI == EPI_RESIDUAL) {
        // h = x + r (language.py:151,153): Linear output rounded to T, then the add rounded to T
        if (live) *reinterpret_cast<u32 *>(a.resid + R) = pack2<T>(lo_f32<T>(pre_u) + round_T<T>(va), hi_f32<T>(pre_u) + round_T<T>(vb));
    } else if (EPI == EPI_SWIGLU) {
        // down_proj input: nn.silu(gate) * up (language.py:127); packed rows (2i, 2i+1) = (gate_i, up_i)
        if (live) {
            const float gte = round_T<T>(va), up = round_T<T>(vb);
            const float slu = round_T<T>(gte / (1.0f + expf(-gte)));
            y_out[pair] = T::from_f32(slu * up);
        }
    } else if (EPI == EPI_ROPE_KV) {
        // packed rows of [q;k;v]: for q/k heads (2i, 2i+1) = dims (i, i + D/2) of one head; v rows natural.
        // RoPE (llama/utils.py:42-50, offset = cache.offset) + cache append (reusable.py:136-137).
        if (live) {
            const int HD = a.head_dim, half = HD >> 1;
            const int q_rows = a.n_heads * HD, k_rows = a.n_kv_heads * HD;
            const float ra = round_T<T>(va), rb = round_T<T>(vb);
            if (R < q_rows + k_rows) {
                const int rr = R < q_rows ? R : R - q_rows;
                const int head = rr / HD, ii = (rr % HD) >> 1;
                u16 *dst = R < q_rows ? q_dst + (size_t)head * HD : kdst + ((size_t)head * cap + kvrow) * HD;
                const int i0 = a.rope_traditional ? 2 * ii : ii, i1 = a.rope_traditional ? 2 * ii + 1 : ii + half;
                dst[i0] = T::from_f32(__fsub_rn(__fmul_rn(ra, pre_cs), __fmul_rn(rb, pre_sn)));
                dst[i1] = T::from_f32(__fadd_rn(__fmul_rn(ra, pre_sn), __fmul_rn(rb, pre_cs)));
            } else {
                const int rr = R - q_rows - k_rows;
                const int head = rr / HD, dd2 = rr % HD;
                *reinterpret_cast<u32 *>(vdst + ((size_t)head * cap + kvrow) * HD + dd2) = pack2<T>(ra, rb);
            }
        }
    }
    GEMV_STAMP(3);
#if defined(PIE_GEMV_PROF) && PIE_GEMV_PROF == 3
    if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        a.prof[8] = ts_entry, a.prof[9] = ts_args, a.prof[10] = ts_x, a.prof[11] = ts_b1, a.prof[12] = ts_staged, a.prof[13] = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (FUSE) {
        static_assert(EPI == EPI_ROPE_KV, "the seam follows the q|k|v epilogue");
        const int c = blockIdx.x & 7, j = blockIdx.x >> 3;
        unsigned *cnt = a.seam + 32 * c;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's q / k / v stores are in the XCD's L2
        __syncthreads();
        // the arrival: an atomic performed AT the L2, nothing returned, nothing re-armed -- the release is the 32nd add itself
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // attention workgroups of the group: one per (query head, split) -- the LAST ones of the class: workgroups are dispatched in index order, so when
        // one of them spins every lower-indexed workgroup of the launch is already running (and will arrive and leave); the slots a launch can hold while
        // it waits are its <= 128 attention workgroups, and five such launches still leave the dispatcher room (3 workgroups of this kernel per CU)
        const int n_attn = 4 * a.attn.splits, ja = j - (32 - n_attn);
        if (ja >= 0) {  // query head 4 c + (ja & 3), split ja >> 2 of kv-group c
            auto seam_wait = [&]() {  // called by the attention body once its old K / V rows are in flight
                if (threadIdx.x == 0) {
                    unsigned spins = 0;
                    while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seam_target) < 0) {  // sc1: from the L2
                        __builtin_amdgcn_s_sleep(0);
                        if (++spins > (1u << 24)) {  // ~ seconds: the group is not co-resident (masked CUs?) -- say so and go on rather than hang the device
                            a.seam[256] = 1u;
                            break;
                        }
                    }
                }
                __syncthreads();
            };
            // one head per workgroup (the launch form scores the group's four heads in one): 16 instead of 4 of the group's 32 workgroups share the
            // post-seam work -- scoring 1.1 -> ~0.5 us on the critical path; a head's arithmetic does not depend on the grouping.
            // (The K / V prefetch goes out before the wait; issued ahead of the weight stream it delayed the GEMV: 1.174 -> 1.195 ms.)
            if (a.attn.block_table) attn_decode_body<T, 128, 1, true, false, 4>(a.attn, c, ja >> 2, 0, 4 * c + (ja & 3), seam_wait);  // T pages (uniform)
            else attn_decode_body<T, 128, 1, false, false, 4>(a.attn, c, ja >> 2, 0, 4 * c + (ja & 3), seam_wait);
        } else if (a.attn.pf_rows > 0) {  // Infinity-Cache warm-up of o_proj (attention.hpp: AttnArgs::pf_ptr)
            const unsigned nblk = 8u * (32u - (unsigned)n_attn), bid = (unsigned)c * (32u - (unsigned)n_attn) + (unsigned)j;
            unsigned acc = 0;
            const unsigned long long n16 = a.attn.pf_bytes >> 4;
            const uint4 *src = reinterpret_cast<const uint4 *>(a.attn.pf_ptr);
            for (unsigned long long i = (unsigned long long)bid * NT + threadIdx.x; i < n16; i += (unsigned long long)nblk * NT) {
                const uint4 v = src[i];
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
            if (acc == 0x9e3779b9u) a.attn.pf_sink[0] = acc;
        }
    }
}

// ---------------------------------------------------------------- a few activation rows against ONE pass over the weights
// y[m, :] = T(x[m, :] @ W^T) for 2 .. 5 rows (MLX's qmv regime: below 6 rows nn.QuantizedLinear multiplies row by row in exact
// fp32, mx.quantized_matmul): the same persistent-wave stream as k_w4s_gemv with the MR rows' activation images side by side in
// LDS -- every weight unit is fetched once and multiplied MR times, each row with the batch-1 kernel's arithmetic in the batch-1
// kernel's order, so a row's result is bit-identical to that row multiplied alone.  (k_w4s_gemv with blockIdx.y = row streams
// the matrix once per row: two sequences cost two batch-1 steps.)  W4S only.  The same prologues / epilogues as the batch-1
// kernel, per row: RMSNorm; plain store (+ bias) | residual add | SwiGLU | RoPE + append to the row's own page | logits + per-wave
// log-softmax partials -- the multi-sequence decode step of up to 5 sequences is then the batch-1 launch sequence, once.
struct GemvRowsArgs {
    const char *w;
    int n_pairs, n_slices, n_waves, K, N, M;
    int full_rounds, rem_pairs, n_blocks;  // launcher: the row-pair map of k_w4s_gemv (GemvArgs)
    const u16 *x;        // [M, K]
    const u16 *norm_w;   // PRO_RMSNORM
    float eps;
    u16 *y;              // EPI_STORE / EPI_LOGITS [M, N]; EPI_SWIGLU act [M, N / 2]
    const u16 *lin_bias; // EPI_STORE only
    u16 *resid;          // EPI_RESIDUAL [M, N], updated in place
    LogitStat *stats;    // EPI_LOGITS [M, n_waves]
    // EPI_ROPE_KV: every row is its own sequence (pie_decoder_step_batch): position ctx_len[m] - 1 (< 0: idle slot), K / V rows to page
    // block_table[m * bt_stride + pos / 64] of this layer's slab (K block then V block, each [n_kv_heads, 64, head_dim])
    const float *rope_cs;  // [M, head_dim / 2, 2] (cos, sin) of every row's position (k_rope_cs_rows)
    const int *ctx_len, *block_table;
    u16 *slab, *q_out;     // q_out [M, n_heads, head_dim]
    int bt_stride, n_pages, n_heads, n_kv_heads, head_dim, rope_traditional;
};
static inline __host__ __device__ unsigned gemv_rows_image_bytes(int K) { return (unsigned)((gemv_lds(K).off_red + 15) & ~15); }
static inline __host__ __device__ unsigned gemv_rows_lds_bytes(int K, int MR) {
    return (unsigned)MR * (gemv_rows_image_bytes(K) + GEMV_WAVES * 2 * GEMV_MAX_RUN * 4) + 64u * (unsigned)MR;  // images | row sums | RMSNorm partials
}

template <class T, int MR, int PRO, int EPI>
__global__ void __launch_bounds__(GEMV_WAVES * 64, 2) k_w4s_gemv_rows(const GemvRowsArgs a) {
    constexpr int D = GEMV_DEPTH, UB = W4S_UNIT_BYTES, NT = GEMV_WAVES * 64, NPT = 2;  // NPT: activation pieces per thread of a NORMALISED input (K <= 8192)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ns = a.n_slices;
    const GemvLds L = gemv_lds(a.K);
    const unsigned img = gemv_rows_image_bytes(a.K);
    const int row0 = blockIdx.y * MR;
    const int nr = a.M - row0 < MR ? a.M - row0 : MR;  // rows of this chunk (wave-uniform)
    constexpr int OUT_ROW = GEMV_WAVES * 2 * GEMV_MAX_RUN;
    float *outp = reinterpret_cast<float *>(smem + (size_t)MR * img) + wave * (2 * GEMV_MAX_RUN);  // + r * OUT_ROW per activation row
    float *red = reinterpret_cast<float *>(smem + (size_t)MR * img) + MR * OUT_ROW;                // [MR][16]

    const int gw = blockIdx.x * GEMV_WAVES + wave;
    const int W = a.n_waves;
    // leftover pairs dealt over the workgroups, as in k_w4s_gemv (here the busiest CU's VALU work, not only its ingest, sets the launch's time)
    const int kf = a.full_rounds;
    const int rem_r = a.n_blocks > 0 ? wave * a.n_blocks + (int)blockIdx.x : gw;
    const bool has_rem = rem_r < a.rem_pairs;
    const int run = (gw < W ? kf : 0) + (has_rem ? 1 : 0);
    const int last_pair = kf * W + rem_r;
    const int n_units = run * ns;
    typedef __attribute__((ext_vector_type(4))) u32 u32x4_t;
    const unsigned w_bytes = (unsigned)((size_t)a.n_pairs * ns * UB);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.w), 0, (int)w_bytes, 0x00020000);
    const unsigned woff0 = (unsigned)((size_t)gw * ns * UB) + lane * 16;
    const unsigned woff_last = (unsigned)((size_t)last_pair * ns * UB) + lane * 16;
    const unsigned pstride32 = (unsigned)((size_t)W * ns * UB);
    const int my_chunks = (a.K + 63) >> 6;
    const bool ragged = (my_chunks & 31) != 0;
    int iss_sl = 0, iss_pl = 0;
    uint4 c0[D], c1[D];
    u32 sb[D];
    auto issue = [&](int d) {
        unsigned off = iss_pl < run ? (iss_pl < kf ? woff0 + (unsigned)iss_pl * pstride32 : woff_last) + (unsigned)iss_sl * UB : 0xFFFFF000u;
        if (ragged && iss_sl * 32 + (lane & 31) >= my_chunks) off = 0xFFFFF000u;
        if (++iss_sl == ns) iss_sl = 0, ++iss_pl;
        const u32x4_t v0 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 2);
        const u32x4_t v1 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off + 1024, 0, 2);
        c0[d] = make_uint4(v0.x, v0.y, v0.z, v0.w);
        c1[d] = make_uint4(v1.x, v1.y, v1.z, v1.w);
        sb[d] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, off + 2048 - lane * 12, 0, 2);
    };
    // 1. the rows' activations (their loads retire before the weight stream's), then the head of the stream
    const int n_pieces = a.K >> 3;
    auto publish = [&](int r, int j, const uint4 &v) {  // piece j of row r -> its LDS image, group sums by DPP over 8 lanes
        char *im = smem + (size_t)r * img;
        const bool ok = j < n_pieces;
        float ps = ok ? sum8<T>(v) : 0.0f;
        ps = lanes8_sum(ps);
        if (ok) {
            *reinterpret_cast<uint4 *>(im + ((size_t)(j & 7) * L.stride + (j >> 3)) * 16) = scale8<T>(v);
            if ((j & 7) == 0) reinterpret_cast<float *>(im + L.off_sx)[j >> 3] = ps;
        }
    };
    if constexpr (PRO == PRO_RMSNORM) {  // mx.fast.rms_norm per row, with the batch-1 prologue's summation tree
        uint4 xv[MR][NPT], nv[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            int j = threadIdx.x + i * NT;
            j = j < n_pieces ? j : n_pieces - 1;
            nv[i] = reinterpret_cast<const uint4 *>(a.norm_w)[j];
#pragma unroll
            for (int r = 0; r < MR; ++r) xv[r][i] = reinterpret_cast<const uint4 *>(a.x + (size_t)(row0 + (r < nr ? r : 0)) * a.K)[j];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            float ssq = 0.0f;
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const bool ok = threadIdx.x + i * NT < n_pieces;
                const u32 v[4] = {xv[r][i].x, xv[r][i].y, xv[r][i].z, xv[r][i].w};
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
            if (lane == 0) red[r * 16 + wave] = ssq;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const float4 ra = *reinterpret_cast<const float4 *>(red + r * 16), rb = *reinterpret_cast<const float4 *>(red + r * 16 + 4);
            const float tot = ((ra.x + ra.y) + (ra.z + ra.w)) + ((rb.x + rb.y) + (rb.z + rb.w));
            const float inv = 1.0f / sqrtf(tot / (float)a.K + a.eps);
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const u32 v[4] = {xv[r][i].x, xv[r][i].y, xv[r][i].z, xv[r][i].w}, g[4] = {nv[i].x, nv[i].y, nv[i].z, nv[i].w};
                u32 o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = pack2<T>(round_T<T>(lo_f32<T>(v[k]) * inv) * lo_f32<T>(g[k]), round_T<T>(hi_f32<T>(v[k]) * inv) * hi_f32<T>(g[k]));
                if (r < nr && i * NT < n_pieces) publish(r, threadIdx.x + i * NT, make_uint4(o[0], o[1], o[2], o[3]));
            }
        }
    } else {
        // every row's pieces of a pass are requested before the first is published: written row by row (load, publish, next row) each
        // row paid its own L2 round trip -- ~1.5 us per row and launch, most of what a second row cost the o_proj launch
        const int n_iter = (n_pieces + NT - 1) / NT;
        constexpr int NI = 2;  // passes in flight (K <= 8192: all of them)
        for (int i0 = 0; i0 < n_iter; i0 += NI) {
            uint4 xs[MR][NI];
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                const uint4 *xg = reinterpret_cast<const uint4 *>(a.x + (size_t)(row0 + (r < nr ? r : 0)) * a.K);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int j = threadIdx.x + (i0 + i) * NT;
                    xs[r][i] = xg[j < n_pieces ? j : n_pieces - 1];
                }
            }
            if (i0 == 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) issue(d);
            }
#pragma unroll
            for (int r = 0; r < MR; ++r)
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (r < nr && i0 + i < n_iter) publish(r, threadIdx.x + (i0 + i) * NT, xs[r][i]);
        }
    }
    __syncthreads();

    // epilogue-side operands, requested now so that their latency hides under the stream (lane l owns local pair l of every row): in the
    // epilogue itself each row's loads waited behind the previous row's stores
    const bool live = lane < run;
    const int pair = lane < kf ? gw + lane * W : last_pair, R = 2 * pair;
    u32 pre_u[EPI == EPI_RESIDUAL ? MR : 1];
    int pre_pos[EPI == EPI_ROPE_KV ? MR : 1];
    unsigned pre_pg[EPI == EPI_ROPE_KV ? MR : 1];
    float2 pre_cs[EPI == EPI_ROPE_KV ? MR : 1];
    if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int r = 0; r < MR; ++r) pre_u[r] = (r < nr && live) ? *reinterpret_cast<const u32 *>(a.resid + (size_t)(row0 + r) * a.N + R) : 0u;
    }
    if constexpr (EPI == EPI_ROPE_KV) {
        const int HD = a.head_dim, q_rows = a.n_heads * HD, k_rows = a.n_kv_heads * HD;
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const int row = row0 + (r < nr ? r : 0);
            pre_pos[r] = a.ctx_len[row] - 1;
            const int pc = pre_pos[r] < 0 ? 0 : pre_pos[r];
            pre_pg[r] = min((unsigned)a.block_table[(size_t)row * a.bt_stride + (pc >> 6)], (unsigned)a.n_pages - 1u);
            pre_cs[r] = make_float2(1.0f, 0.0f);
            if (live && R < q_rows + k_rows) pre_cs[r] = *reinterpret_cast<const float2 *>(a.rope_cs + ((size_t)row * (HD >> 1) + ((R % HD) >> 1)) * 2);
        }
    }

    // 2. the stream: every unit once; its dot2 operands are formed once (5 of the 9 VALU instructions per code word) and multiplied
    //    with every row.  XREG: for K of at most two slices and at most two rows the rows' activations live in registers (2 x MR x 33),
    //    otherwise they are re-read from the LDS images per unit (8 x ds_read_b128 per row).
    const int n_groups = a.K >> 6;
    constexpr bool XREG = MR <= 2 && D == 2;
    const bool xreg = XREG && ns == 2;  // wave-uniform; with two slices per row and a ring of two, ring slot d always holds slice d: static indexing
    u32 xc[XREG ? 2 : 1][XREG ? MR : 1][32];
    float sxc[XREG ? 2 : 1][XREG ? MR : 1];
    auto load_x = [&](int r, int slice, u32 (&xr)[32], float &sx) {
        const int g2 = slice * 32 + (lane & 31);
        const int gc2 = g2 < n_groups ? g2 : n_groups - 1;
        const char *im = smem + (size_t)r * img;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 v = *reinterpret_cast<const uint4 *>(im + ((size_t)q * L.stride + gc2) * 16);
            xr[4 * q + 0] = v.x, xr[4 * q + 1] = v.y, xr[4 * q + 2] = v.z, xr[4 * q + 3] = v.w;
        }
        sx = reinterpret_cast<const float *>(im + L.off_sx)[gc2];
    };
    if constexpr (XREG) {
        if (xreg) {
#pragma unroll
            for (int sl2 = 0; sl2 < 2; ++sl2)
#pragma unroll
                for (int r = 0; r < MR; ++r) load_x(r < nr ? r : 0, sl2, xc[sl2][r], sxc[sl2][r]);
        }
    }
    float acc[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[r] = 0.0f;
    int sl = 0, pl = 0;
    for (int base = 0; base < n_units; base += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (base + d < n_units) {  // wave-uniform
                if (((((base + d) >> 1) + (wave >> 2)) & 1)) __builtin_amdgcn_s_setprio(1);  // the two waves of a SIMD take turns (k_w4s_gemv)
                else __builtin_amdgcn_s_setprio(0);
                const bool gvalid = sl * 32 + (lane & 31) < n_groups;
                const float scale = lo_f32<T>(sb[d]), bias = hi_f32<T>(sb[d]);
                u32 e[32];
                w4s_unit_ops<T>(c0[d], c1[d], e);
#pragma unroll
                for (int r = 0; r < MR; ++r) {
                    if (r < nr) {
                        float dd = 0.0f, sx = 0.0f;
                        bool done = false;
                        if constexpr (XREG) {
                            if (xreg) {
                                dd = w4s_ops_dot<T>(e, xc[d & 1][r]), sx = sxc[d & 1][r];
                                done = true;
                            }
                        }
                        if (!done) {
                            u32 xr[32];
                            load_x(r, sl, xr, sx);
                            dd = w4s_ops_dot<T>(e, xr);
                        }
                        const float pr = fmaf(scale, dd * T::DSCALE - T::OFFSET * sx, bias * sx);
                        acc[r] += gvalid ? pr : 0.0f;
                    }
                }
                if (++sl == ns) {
#pragma unroll
                    for (int r = 0; r < MR; ++r) {
                        const float tot = half_wave_sum(acc[r]);
                        if ((lane & 31) == 31) outp[r * OUT_ROW + 2 * pl + (lane >> 5)] = tot;
                        acc[r] = 0.0f;
                    }
                    sl = 0, ++pl;
                }
            }
            issue(d);
        }
    }
    // 3. epilogue: lane l owns local pair l (rows R, R + 1 of the packed order) of every activation row
    u32 pre_b = 0;
    if (EPI == EPI_STORE && a.lin_bias && live) pre_b = *reinterpret_cast<const u32 *>(a.lin_bias + R);
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        if (r >= nr) continue;  // wave-uniform
        const int row = row0 + r;
        float va = 0.0f, vb = 0.0f;
        if (live) {
            const float2 o = *reinterpret_cast<const float2 *>(outp + r * OUT_ROW + 2 * lane);
            va = o.x, vb = o.y;
        }
        if constexpr (EPI == EPI_STORE || EPI == EPI_LOGITS) {
            float oa = round_T<T>(va), ob = round_T<T>(vb);
            if (EPI == EPI_STORE && a.lin_bias) oa = round_T<T>(oa + lo_f32<T>(pre_b)), ob = round_T<T>(ob + hi_f32<T>(pre_b));  // y = T(T(x W^T) + b)
            if (live) *reinterpret_cast<u32 *>(a.y + (size_t)row * a.N + R) = pack2<T>(oa, ob);
            if constexpr (EPI == EPI_LOGITS) {  // per-wave log-softmax partial of this row: max, first argmax, sum exp(x - max)
                const float mx = live ? fmaxf(oa, ob) : -INFINITY;
                const int ix = live ? (ob > oa ? R + 1 : R) : 0x7fffffff;
                const float wmax = wave_max(mx);
                int cand = (live && mx == wmax) ? ix : 0x7fffffff;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
                float se = live ? expf(oa - wmax) + expf(ob - wmax) : 0.0f;
                se = wave_sum(se);
                if (lane == 0 && gw < a.n_waves) {
                    LogitStat st;
                    st.max = wmax, st.sumexp = se, st.argmax = cand, st.pad = 0;
                    a.stats[(size_t)row * a.n_waves + gw] = st;
                }
            }
        } else if constexpr (EPI == EPI_RESIDUAL) {  // h = x + r: Linear output rounded to T, then the add rounded to T
            if (live) {
                const u32 h2 = pre_u[r];
                *reinterpret_cast<u32 *>(a.resid + (size_t)row * a.N + R) = pack2<T>(lo_f32<T>(h2) + round_T<T>(va), hi_f32<T>(h2) + round_T<T>(vb));
            }
        } else if constexpr (EPI == EPI_SWIGLU) {  // nn.silu(gate) * up; packed rows (2 i, 2 i + 1) = (gate_i, up_i)
            if (live) {
                const float gte = round_T<T>(va), up = round_T<T>(vb);
                const float slu = round_T<T>(gte / (1.0f + expf(-gte)));
                a.y[(size_t)row * (a.N >> 1) + pair] = T::from_f32(slu * up);
            }
        } else if constexpr (EPI == EPI_ROPE_KV) {  // RoPE at the row's own position + append to the row's own page
            const int pos = pre_pos[r];
            if (live && pos >= 0) {
                const int HD = a.head_dim, half = HD >> 1, q_rows = a.n_heads * HD, k_rows = a.n_kv_heads * HD;
                const unsigned pg = pre_pg[r];
                u16 *kdst = a.slab + (size_t)pg * 2 * 64 * a.n_kv_heads * HD, *vdst = kdst + (size_t)a.n_kv_heads * 64 * HD;
                const int kvrow = pos & 63;
                const float ra = round_T<T>(va), rb = round_T<T>(vb);
                if (R < q_rows + k_rows) {
                    const int rr = R < q_rows ? R : R - q_rows;
                    const int head = rr / HD, ii = (rr % HD) >> 1;
                    const float2 csn = pre_cs[r];
                    u16 *dst = R < q_rows ? a.q_out + ((size_t)row * a.n_heads + head) * HD : kdst + ((size_t)head * 64 + kvrow) * HD;
                    const int i0 = a.rope_traditional ? 2 * ii : ii, i1 = a.rope_traditional ? 2 * ii + 1 : ii + half;
                    dst[i0] = T::from_f32(__fsub_rn(__fmul_rn(ra, csn.x), __fmul_rn(rb, csn.y)));
                    dst[i1] = T::from_f32(__fadd_rn(__fmul_rn(ra, csn.y), __fmul_rn(rb, csn.x)));
                } else {
                    const int rr = R - q_rows - k_rows;
                    *reinterpret_cast<u32 *>(vdst + ((size_t)(rr / HD) * 64 + kvrow) * HD + rr % HD) = pack2<T>(ra, rb);
                }
            }
        }
    }
}

// Host-side launch: sizes the persistent grid for (N, K) and dispatches the template.
int w4s_gemv_launch(int dtype, int pro, int epi, GemvArgs &a, int M, hipStream_t stream);
// Number of waves (= log-softmax partials with EPI_LOGITS) the launcher uses for an [N, K] weight.
int w4s_gemv_waves(int N, int K);
// y[M, N] = T(x[M, K] @ W^T) (+ bias) for 1 <= M <= GEMV_ROWS_MAX rows of one W4S matrix in one pass over the weights (see k_w4s_gemv_rows)
constexpr int GEMV_ROWS_MAX = 5;
int w4s_gemv_rows_launch(int dtype, const void *packed, int N, int K, const u16 *x, int M, u16 *y, const u16 *lin_bias, hipStream_t stream);
// the fused forms (pro: PRO_NONE / PRO_RMSNORM; epi: EPI_STORE / EPI_RESIDUAL / EPI_SWIGLU / EPI_ROPE_KV / EPI_LOGITS): w, K, N, M and the operands of the
// chosen prologue / epilogue set by the caller
int w4s_gemv_rows_fused_launch(int dtype, int pro, int epi, GemvRowsArgs &a, hipStream_t stream);
