// sampler.hip -- the stochastic samplers of the reference over the fp32 log-probabilities, sort-free.
//
// samplers/__init__.py:37-46 picks exactly one branch per request: top-p (samplers/top_p.py:18-33), min-p (min_p.py:42-60), top-k
// (top_k.py:24-29) or plain categorical (categorical.py:6-8); each scales by 1 / temperature, filters, and draws with
// mx.random.categorical = argmax(x + Gumbel noise).  The reference's default request has temp = 1.0, so this runs once per token right
// behind the decode step.  Restated with library ops it is a sort of 128 k floats plus ~10 small launches (150-200 us, 10-15 % of a
// step); a first single-workgroup version of this file took 120-480 us because one CU did all the arithmetic.  This form spreads a
// row over up to 256 workgroups and needs no sort:
//   * the filter is "every id whose value is at least T": T comes from a 3-digit radix select (11 + 11 + 10 bits) over the
//     order-preserving 32-bit keys of x = logprob / temp.  Each digit is one launch: workgroups histogram their slice in LDS and add
//     the bins to a global histogram with 64-bit INTEGER atomics (order-independent); the next launch's workgroups all walk that
//     histogram to the same digit.  top-k counts ids; top-p weighs them with exp(x - max) as 2^-40 fixed-point integers and looks
//     for the ascending cumulative mass (1 - top_p) x total -- exact arithmetic where the reference's fp32 cumsum over the sorted row
//     rounds as it goes, so the kept set can differ from the reference's only for ids whose cumulative mass lies within that rounding
//     of 1 - top_p; min-p is a plain threshold, max + log(min_p), joined with the top min_tokens_to_keep;
//   * equal values at the top-k boundary are common (log-probabilities of bf16 logits repeat): exactly k ids are kept, the ties with
//     the lowest vocabulary index first (mx.argpartition leaves the choice open);
//   * the draw is argmax(x + G) over the kept ids, G = -log(-log(u)), u from Philox-4x32-10 keyed by (seed, call counter, row,
//     vocabulary index); per-workgroup winners meet in one 64-bit atomic max.  The call counter lives in device memory and is advanced
//     by the last workgroup, so a captured graph keeps drawing fresh numbers.  The random stream is this library's own, not MLX's:
//     parity is the kept-token set and the distribution.
#include "common.hpp"

namespace {

constexpr int SMP_T = 256;          // threads per workgroup
constexpr int SMP_SLICE = 512;      // ids per workgroup
constexpr int SMP_MAX_WGS = 1024;    // V <= 524288
constexpr int SMP_BINS = 2048;
// per-row workspace, in 64-bit words: header | hist digit 0 | hist digit 1 | hist digit 2 | per-workgroup digit-2 counts (u32)
constexpr int SMP_HDR = 8;
constexpr size_t smp_row_words(int wgs) { return SMP_HDR + 3 * (size_t)SMP_BINS + (size_t)wgs * 1024 / 2; }
enum { H_MAXKEY = 0, H_BEST = 1, H_ARRIVE = 2 };

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
}
// four 32-bit words for counter (idx4, row, call) under key `seed` (Philox-4x32-10, Salmon et al. 2011)
__device__ __forceinline__ void philox(unsigned long long seed, unsigned long long call, unsigned row, unsigned idx4, unsigned (&out)[4]) {
    unsigned c[4] = {idx4, row, (unsigned)call, (unsigned)(call >> 32)};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}
__device__ __forceinline__ float gumbel(unsigned r) {
    const float u = ((float)(r >> 8) + 0.5f) * 0x1p-24f;  // (0, 1), never 0 or 1
    return -logf(-logf(u));
}
// order-preserving key of a float (larger float <-> larger key)
__device__ __forceinline__ unsigned okey(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

enum { SMP_CATEGORICAL = 0, SMP_TOP_K = 1, SMP_TOP_P = 2, SMP_MIN_P = 3 };

struct SmpArgs {
    const float *logprobs;
    int V, mode;
    float inv_temp, thr;  // thr: fp32(1 - top_p) or fp32(log(min_p))
    int k;                // top_k, or min_tokens_to_keep
    unsigned long long seed;
    unsigned long long *counter;  // [0] calls so far, [1] rows finished in this call
    unsigned long long *ws;       // rows x smp_row_words(gridDim.x)
    int *token_out, *kept_count;
    unsigned char *kept_mask;
};

// The selection runs in ASCENDING key order with an integer target: the answer is the smallest key T with
//   weight{key < T} <= target < weight{key <= T}.
// top-p: key = okey(x), weight = exp(x - max) in fixed point, target = (1 - top_p) x total: T = smallest kept value.
// top-k / min_tokens_to_keep: key = ~okey(x) (descending values), weight = 1, target = k - 1: T = complement of the k-th largest value.
__device__ __forceinline__ bool smp_select_on_probs(int mode) { return mode == SMP_TOP_P; }
__device__ __forceinline__ bool smp_has_select(const SmpArgs &a) { return a.mode == SMP_TOP_P || a.mode == SMP_TOP_K || (a.mode == SMP_MIN_P && a.k > 1); }

// Walks one global histogram (n bins) from bin 0 up: first bin b with cum + h[b] > target.  Every thread returns the same (b, cum below b).
// *total = sum of all bins.  256 threads, n <= 2048.
__device__ void smp_walk(const unsigned long long *h, int n, unsigned long long cum0, unsigned long long target, unsigned *bin, unsigned long long *cum,
                         unsigned long long *total, unsigned long long *s_scan /* [SMP_T + 2] */) {
    const int per = n / SMP_T;  // 8 or 4
    unsigned long long loc[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        loc[j] = j < per ? __hip_atomic_load(h + threadIdx.x * per + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        sum += loc[j];
    }
    // exclusive prefix over the threads: shuffle scan inside a wave, wave totals through LDS (a serial 256-step walk per thread made
    // each call 4 us: the three-digit select spent 60 of its 80 us walking)
    unsigned long long incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_scan[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long before = incl - sum, all = 0;
#pragma unroll
    for (int w = 0; w < SMP_T / 64; ++w) {
        const unsigned long long v = s_scan[w];
        if (w < (int)(threadIdx.x >> 6)) before += v;
        all += v;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_scan[SMP_T] = 0xFFFFFFFFFFFFFFFFull, s_scan[SMP_T + 1] = 0;
    __syncthreads();
    unsigned long long c = cum0 + before;
    if (c <= target && c + sum > target) {  // exactly one thread: the crossing lies in its bins
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < per && s_scan[SMP_T] == 0xFFFFFFFFFFFFFFFFull) {
                if (c + loc[j] > target) s_scan[SMP_T] = (unsigned long long)(threadIdx.x * per + j), s_scan[SMP_T + 1] = c;
                else c += loc[j];
            }
    }
    __syncthreads();
    const bool found = s_scan[SMP_T] != 0xFFFFFFFFFFFFFFFFull;
    *bin = found ? (unsigned)s_scan[SMP_T] : (unsigned)(n - 1);   // target beyond the total: the last bin (the caller keeps nothing above it)
    *cum = found ? s_scan[SMP_T + 1] : cum0 + all;
    *total = all;
    __syncthreads();
}

__device__ __forceinline__ unsigned long long smp_weight(const SmpArgs &a, float xs, float xmax) {
    return a.mode == SMP_TOP_P ? (unsigned long long)(expf(xs - xmax) * 0x1p40f) : 1ull;
}
__device__ __forceinline__ unsigned smp_key(const SmpArgs &a, float xs) { return a.mode == SMP_TOP_P ? okey(xs) : ~okey(xs); }

// launch 0: clears the row's workspace and finds max(x) (needed by top-p's weights and min-p's threshold)
__global__ void __launch_bounds__(SMP_T) k_smp_init(const SmpArgs a) {
    __shared__ unsigned s_m[SMP_T / 64];
    const unsigned row = blockIdx.y, G = gridDim.x;
    unsigned long long *ws = a.ws + (size_t)row * smp_row_words(G);
    const size_t words = SMP_HDR + 3 * (size_t)SMP_BINS;  // the per-workgroup count table is fully rewritten by digit 2's launch
    for (size_t i = (size_t)blockIdx.x * SMP_T + threadIdx.x; i < words; i += (size_t)G * SMP_T)
        if (i != H_MAXKEY) ws[i] = 0;
    const float *x = a.logprobs + (size_t)row * a.V;
    unsigned m = 0;
    for (int i = blockIdx.x * SMP_SLICE + threadIdx.x; i < min(a.V, (int)(blockIdx.x + 1) * SMP_SLICE); i += SMP_T) m = max(m, okey(x[i] * a.inv_temp));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SMP_T / 64; ++w) m = max(m, s_m[w]);
        atomicMax(reinterpret_cast<unsigned *>(ws + H_MAXKEY), m);
    }
}
// H_MAXKEY must be zero before k_smp_init's atomicMax: cleared by the previous call's draw launch (and by the host at allocation)

// launches 1..3: one radix digit each (DIGIT 0: bits 31..21, 1: bits 20..10, 2: bits 9..0)
template <int DIGIT>
__global__ void __launch_bounds__(SMP_T) k_smp_digit(const SmpArgs a) {
    __shared__ unsigned long long s_h[SMP_BINS];
    __shared__ unsigned long long s_scan[SMP_T + 2];
    const unsigned row = blockIdx.y, G = gridDim.x;
    unsigned long long *ws = a.ws + (size_t)row * smp_row_words(G);
    unsigned long long *h0 = ws + SMP_HDR, *h1 = h0 + SMP_BINS, *h2 = h1 + SMP_BINS;
    const float xmax = okey_inv((unsigned)__hip_atomic_load(ws + H_MAXKEY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    // the digits decided so far (every workgroup repeats the walk: 16 KB of L2-resident bins)
    unsigned prefix = 0;
    if (DIGIT >= 1) {
        unsigned long long total, cum, target;
        unsigned b0;
        smp_walk(h0, SMP_BINS, 0, 0xFFFFFFFFFFFFFFFFull - 1, &b0, &cum, &total, s_scan);  // total first
        target = a.mode == SMP_TOP_P ? (a.thr <= 0.0f ? 0ull : (unsigned long long)((double)a.thr * (double)total)) : (unsigned long long)(a.k - 1);
        smp_walk(h0, SMP_BINS, 0, target, &b0, &cum, &total, s_scan);
        prefix = b0;
        if (DIGIT == 2) {
            unsigned b1;
            unsigned long long t1;
            smp_walk(h1, SMP_BINS, cum, target, &b1, &cum, &t1, s_scan);
            prefix = (b0 << 11) | b1;
        }
    }
    const int nb = DIGIT == 2 ? 1024 : SMP_BINS;
    for (int i = threadIdx.x; i < nb; i += SMP_T) s_h[i] = 0;
    __syncthreads();
    const float *x = a.logprobs + (size_t)row * a.V;
    for (int i = blockIdx.x * SMP_SLICE + threadIdx.x; i < min(a.V, (int)(blockIdx.x + 1) * SMP_SLICE); i += SMP_T) {
        const float xs = x[i] * a.inv_temp;
        const unsigned key = smp_key(a, xs);
        bool mine;
        unsigned d;
        if (DIGIT == 0) mine = true, d = key >> 21;
        else if (DIGIT == 1) mine = (key >> 21) == prefix, d = (key >> 10) & 2047u;
        else mine = (key >> 10) == prefix, d = key & 1023u;
        if (mine) atomicAdd(&s_h[d], smp_weight(a, xs, xmax));
    }
    __syncthreads();
    unsigned long long *hg = DIGIT == 0 ? h0 : (DIGIT == 1 ? h1 : h2);
    for (int i = threadIdx.x; i < nb; i += SMP_T)
        if (s_h[i]) atomicAdd(hg + i, s_h[i]);
    if (DIGIT == 2) {  // this workgroup's counts per last digit: the draw launch rations ties in index order with them
        unsigned *wg = reinterpret_cast<unsigned *>(ws + SMP_HDR + 3 * SMP_BINS) + (size_t)blockIdx.x * 1024;
        for (int i = threadIdx.x; i < 1024; i += SMP_T) wg[i] = (unsigned)s_h[i];
    }
}

// last launch: the filter threshold from the three histograms, then the Gumbel-max draw over the kept ids
__global__ void __launch_bounds__(SMP_T) k_smp_draw(const SmpArgs a) {
    __shared__ unsigned long long s_scan[SMP_T + 2];
    __shared__ unsigned s_cnt[SMP_T];
    __shared__ unsigned long long s_best[SMP_T / 64];
    __shared__ unsigned s_kept[SMP_T / 64];
    const unsigned row = blockIdx.y, G = gridDim.x;
    unsigned long long *ws = a.ws + (size_t)row * smp_row_words(G);
    unsigned long long *h0 = ws + SMP_HDR, *h1 = h0 + SMP_BINS, *h2 = h1 + SMP_BINS;
    const float xmax = okey_inv((unsigned)__hip_atomic_load(ws + H_MAXKEY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));

    // keep(i) <=> okey(x_i) > T, or == T for the first `eq_take` ties in index order (eq_take = ~0: every tie, as the inclusive bound)
    unsigned T = 0, eq_take = 0xFFFFFFFFu;  // T = 0, all ties: everything (keys of floats are > 0)
    bool keep_none = false;
    if (smp_has_select(a)) {
        unsigned long long total, cum, t1, target;
        unsigned b0, b1, b2;
        smp_walk(h0, SMP_BINS, 0, 0xFFFFFFFFFFFFFFFFull - 1, &b0, &cum, &total, s_scan);
        target = a.mode == SMP_TOP_P ? (a.thr <= 0.0f ? 0ull : (unsigned long long)((double)a.thr * (double)total)) : (unsigned long long)(a.k - 1);
        keep_none = target >= total;  // top_p so small that not even the largest probability crosses 1 - top_p: the draw falls back to argmax
        smp_walk(h0, SMP_BINS, 0, target, &b0, &cum, &t1, s_scan);
        smp_walk(h1, SMP_BINS, cum, target, &b1, &cum, &t1, s_scan);
        smp_walk(h2, 1024, cum, target, &b2, &cum, &t1, s_scan);
        const unsigned sel = (b0 << 21) | (b1 << 10) | b2;
        if (a.mode == SMP_TOP_P) T = sel;  // smallest kept key; every tie kept
        else T = ~sel, eq_take = (unsigned)((unsigned long long)a.k - cum);  // cum ids lie strictly above; the rest of the k come from the ties
        if (a.mode != SMP_TOP_P) {
            // ties before this workgroup's slice (index order): the per-workgroup counts of the last digit
            const unsigned *wg = reinterpret_cast<const unsigned *>(ws + SMP_HDR + 3 * SMP_BINS);
            unsigned before = 0;
            for (int g = threadIdx.x; g < (int)blockIdx.x; g += SMP_T) before += wg[(size_t)g * 1024 + b2];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) before += __shfl_xor((int)before, o, 64);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = before;
            __syncthreads();
            before = 0;
            for (int w = 0; w < SMP_T / 64; ++w) before += s_cnt[w];
            __syncthreads();
            eq_take = eq_take > before ? eq_take - before : 0;  // what is left for this slice and the ones after it
        }
    }
    if (a.mode == SMP_MIN_P) {
        // x >= max + log(min_p) (min_p.py:50-52), and always the first min_tokens_to_keep of the descending order (:54): both sets are
        // "everything from some value up", so the union is the larger one
        const float thr = xmax + a.thr;
        const unsigned Tp = okey(thr);
        if (a.k <= 1 || Tp <= T) T = Tp, eq_take = 0xFFFFFFFFu;  // min-p's set contains the top min_tokens_to_keep
    }

    // this workgroup's slice: thread t owns the two ADJACENT ids 2 t, 2 t + 1 of the slice, so thread order is index order
    const float *x = a.logprobs + (size_t)row * a.V;
    const int base = blockIdx.x * SMP_SLICE + 2 * threadIdx.x;
    float xs[2];
    unsigned key[2];
    bool live[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        live[j] = base + j < a.V && base + j < (int)(blockIdx.x + 1) * SMP_SLICE;
        xs[j] = live[j] ? x[base + j] * a.inv_temp : -INFINITY;
        key[j] = okey(xs[j]);
    }
    unsigned tie_rank = 0;
    if (eq_take != 0xFFFFFFFFu) {  // exclusive scan of the tie counts over the threads
        const unsigned mine = (live[0] && key[0] == T ? 1u : 0u) + (live[1] && key[1] == T ? 1u : 0u);
        s_cnt[threadIdx.x] = mine;
        __syncthreads();
        for (int t = 0; t < (int)threadIdx.x; ++t) tie_rank += s_cnt[t];
        __syncthreads();
    }
    const unsigned long long call = a.counter[0];
    unsigned r4[4];
    philox(a.seed, call, row, (unsigned)base >> 2, r4);  // ids base, base + 1 share one 4-word block (base is even)
    unsigned long long best = 0;  // packed: okey(x + G) << 32 | ~index  (max = larger value, then lower index)
    unsigned n_kept = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bool keep = live[j] && !keep_none && (key[j] > T || (key[j] == T && (eq_take == 0xFFFFFFFFu || tie_rank < eq_take)));
        if (live[j] && key[j] == T && eq_take != 0xFFFFFFFFu) ++tie_rank;
        if (a.kept_mask && live[j]) a.kept_mask[(size_t)row * a.V + base + j] = keep ? 1 : 0;
        if (keep_none && live[j]) {  // fall back to the plain argmax
            const unsigned long long cand = ((unsigned long long)key[j] << 32) | (unsigned)~(unsigned)(base + j);
            best = cand > best ? cand : best;
        }
        if (!keep) continue;
        ++n_kept;
        const float v = xs[j] + gumbel(r4[(base + j) & 3]);
        const unsigned long long cand = ((unsigned long long)okey(v) << 32) | (unsigned)~(unsigned)(base + j);
        best = cand > best ? cand : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(best, o, 64);
        best = ob > best ? ob : best;
        n_kept += (unsigned)__shfl_xor((int)n_kept, o, 64);
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best, s_kept[threadIdx.x >> 6] = n_kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned kept = 0;
        for (int w = 0; w < SMP_T / 64; ++w) best = s_best[w] > best ? s_best[w] : best, kept += s_kept[w];
        atomicMax(ws + H_BEST, best);
        // arrival word: low 32 bits = workgroups done, high 32 bits = kept ids so far
        const unsigned long long arr = atomicAdd(ws + H_ARRIVE, 1ull | ((unsigned long long)kept << 32)) + (1ull | ((unsigned long long)kept << 32));
        if ((unsigned)arr == G) {  // last workgroup of the row
            const unsigned long long win = __hip_atomic_load(ws + H_BEST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.token_out[row] = (int)~(unsigned)win;
            if (a.kept_count) a.kept_count[row] = (int)(arr >> 32);
            ws[H_MAXKEY] = 0;  // ready for the next call's atomic max
            const unsigned long long rows_done = atomicAdd(a.counter + 1, 1ull) + 1;
            if (rows_done == gridDim.y) a.counter[1] = 0, a.counter[0] = call + 1;  // the last row of the call advances the stream
        }
    }
}

}  // namespace

extern "C" {

size_t pie_sample_workspace_bytes(int rows, int V) {
    if (rows < 1 || V < 1) return 0;
    const int G = (V + SMP_SLICE - 1) / SMP_SLICE;
    return G > SMP_MAX_WGS ? 0 : (size_t)rows * smp_row_words(G) * sizeof(unsigned long long);
}

int pie_sample(const float *logprobs, int rows, int V, int mode, double temp, double p, int k, unsigned long long seed, unsigned long long *counter,
               void *workspace, int32_t *tokens, int32_t *kept_count, unsigned char *kept_mask, void *stream) {
    PIE_REQUIRE(logprobs && counter && tokens && workspace, PIE_E_ARG, "pie_sample: null pointer");
    PIE_REQUIRE(rows >= 1 && V >= 1 && V <= SMP_MAX_WGS * SMP_SLICE, PIE_E_SHAPE, "pie_sample: rows >= 1 and 1 <= V <= 524288");
    PIE_REQUIRE(pie_aligned(workspace, 8), PIE_E_ALIGN, "pie_sample: workspace needs 8-byte alignment");
    PIE_REQUIRE(mode >= SMP_CATEGORICAL && mode <= SMP_MIN_P, PIE_E_ARG, "pie_sample: unknown mode");
    PIE_REQUIRE(temp > 0.0, PIE_E_ARG, "pie_sample: temperature must be positive (temp = 0 is the greedy tail, pie_logprobs_argmax)");
    // the reference's own argument checks (top_k.py:20-24, min_p.py:30-40)
    PIE_REQUIRE(mode != SMP_TOP_K || (k > 0 && k < V), PIE_E_ARG, "pie_sample: `top_k` has to be an integer in the (0, V) interval");
    PIE_REQUIRE(mode != SMP_MIN_P || (p > 0.0 && p <= 1.0), PIE_E_ARG, "pie_sample: `min_p` has to be a float in the (0, 1] interval");
    PIE_REQUIRE(mode != SMP_MIN_P || (k >= 1 && k <= V), PIE_E_ARG, "pie_sample: `min_tokens_to_keep` has to be a positive integer");
    PIE_REQUIRE(mode != SMP_TOP_P || (p > 0.0 && p < 1.0), PIE_E_ARG, "pie_sample: `top_p` has to be in (0, 1) (samplers/__init__.py:39)");
    SmpArgs a = {};
    a.logprobs = logprobs, a.V = V, a.mode = mode, a.seed = seed, a.counter = counter, a.ws = (unsigned long long *)workspace;
    a.token_out = tokens, a.kept_count = kept_count, a.kept_mask = kept_mask;
    // Python scalars enter the reference's fp32 arithmetic as the fp32 value of the double expression: 1 / temperature, 1 - top_p, log(min_p)
    a.inv_temp = (float)(1.0 / temp);
    a.thr = mode == SMP_TOP_P ? (float)(1.0 - p) : (mode == SMP_MIN_P ? (float)log(p) : 0.0f);
    a.k = mode == SMP_TOP_K || mode == SMP_MIN_P ? k : 0;
    const dim3 grid((unsigned)((V + SMP_SLICE - 1) / SMP_SLICE), (unsigned)rows), block(SMP_T);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_smp_init, grid, block, 0, st, a);
    if (mode == SMP_TOP_P || mode == SMP_TOP_K || (mode == SMP_MIN_P && k > 1)) {
        hipLaunchKernelGGL(k_smp_digit<0>, grid, block, 0, st, a);
        hipLaunchKernelGGL(k_smp_digit<1>, grid, block, 0, st, a);
        hipLaunchKernelGGL(k_smp_digit<2>, grid, block, 0, st, a);
    }
    hipLaunchKernelGGL(k_smp_draw, grid, block, 0, st, a);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

}  // extern "C"
