// tail.hpp -- the end of _inference (engine/inference_engine.py:254,268-271) with the greedy sampler
// (samplers/__init__.py:37-38): logprobs = f32(logits) - logsumexp(f32(logits)); token = first argmax.
//
// Two stages: per-tile (max, sum exp(x - max), first argmax) partials -- produced either by the lm_head
// GEMV epilogue (EPI_LOGITS) or by k_logits_stats -- then k_logits_finish, in which every workgroup merges
// the partials (a few thousand, L2-resident) and writes its slice of the fp32 logprobs.
#pragma once
#include "common.hpp"
#include "w4_gemv.hpp"  // LogitStat, DecState

constexpr int TAIL_STAT_TILES = 256;
constexpr int TAIL_FINISH_BLOCKS = 64;

template <class T>
__global__ void __launch_bounds__(256) k_logits_stats(const u16 *logits, int V, LogitStat *stats) {
    __shared__ float s_max[4], s_sum[4];
    __shared__ int s_arg[4];
    logits += (size_t)blockIdx.y * V, stats += (size_t)blockIdx.y * gridDim.x;  // blockIdx.y: row of a batch of logit vectors
    const int tile_len = (V + gridDim.x - 1) / gridDim.x;
    const int begin = blockIdx.x * tile_len, end = min(V, begin + tile_len);
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    for (int i = begin + threadIdx.x; i < end; i += 256) {
        const float v = T::to_f32(logits[i]);
        if (v > mx) mx = v, arg = i;  // ascending i per thread: first maximal index wins
    }
    const float wmax = wave_max(mx);
    int cand = (mx == wmax) ? arg : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_max[wave] = wmax, s_arg[wave] = cand;
    __syncthreads();
    float tmax = s_max[0];
    int targ = s_arg[0];
    for (int w = 1; w < 4; ++w)
        if (s_max[w] > tmax || (s_max[w] == tmax && s_arg[w] < targ)) tmax = s_max[w], targ = s_arg[w];
    float se = 0.0f;
    for (int i = begin + threadIdx.x; i < end; i += 256) se += expf(T::to_f32(logits[i]) - tmax);
    se = wave_sum(se);
    if ((threadIdx.x & 63) == 0) s_sum[wave] = se;
    __syncthreads();
    if (threadIdx.x == 0) {
        LogitStat st;
        st.max = tmax, st.sumexp = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3], st.argmax = targ, st.pad = 0;
        stats[blockIdx.x] = st;  // empty tile: max = -inf, sumexp = 0, argmax = INT_MAX
    }
}

constexpr int TAIL_MAX_STATS = 4096;  // 256 threads x 16 register-resident partials

template <class T>
__global__ void __launch_bounds__(256) k_logits_finish(const u16 *logits, int V, const LogitStat *stats, int n_stats, float *logprobs,
                                                       int *token, DecState *state, int *history, int hist_cap, const unsigned *err_word) {
    __shared__ float s_max[4], s_sum[4];
    __shared__ int s_arg[4];
    logits += (size_t)blockIdx.y * V, stats += (size_t)blockIdx.y * n_stats, logprobs += (size_t)blockIdx.y * V, token += blockIdx.y;  // batch row
    // every workgroup merges all partials: 16 independent 16-byte loads per thread, then two register passes
    LogitStat st[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = threadIdx.x + 256 * k;
        const uint4 raw = *reinterpret_cast<const uint4 *>(stats + (i < n_stats ? i : n_stats - 1));
        st[k].max = i < n_stats ? __builtin_bit_cast(float, raw.x) : -INFINITY;
        st[k].sumexp = i < n_stats ? __builtin_bit_cast(float, raw.y) : 0.0f;
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
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_max[wave] = wmax, s_arg[wave] = cand;
    __syncthreads();
    float M = s_max[0];
    int tok = s_arg[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (s_max[w] > M || (s_max[w] == M && s_arg[w] < tok)) M = s_max[w], tok = s_arg[w];
    float se = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) se += st[k].sumexp > 0.0f ? st[k].sumexp * expf(st[k].max - M) : 0.0f;
    se = wave_sum(se);
    if ((threadIdx.x & 63) == 0) s_sum[wave] = se;
    __syncthreads();
    const float lse = M + logf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
    const int slice = (((V + gridDim.x - 1) / gridDim.x) + 7) & ~7;  // 8 logits per thread and pass
    const int begin = blockIdx.x * slice, end = min(V, begin + slice);
    for (int i = begin + threadIdx.x * 8; i < end; i += 256 * 8) {
        if (i + 8 <= end && (V & 7) == 0) {
            const uint4 v = *reinterpret_cast<const uint4 *>(logits + i);
            float4 o0 = make_float4(lo_f32<T>(v.x) - lse, hi_f32<T>(v.x) - lse, lo_f32<T>(v.y) - lse, hi_f32<T>(v.y) - lse);
            float4 o1 = make_float4(lo_f32<T>(v.z) - lse, hi_f32<T>(v.z) - lse, lo_f32<T>(v.w) - lse, hi_f32<T>(v.w) - lse);
            *reinterpret_cast<float4 *>(logprobs + i) = o0;
            *reinterpret_cast<float4 *>(logprobs + i + 4) = o1;
        } else {
            for (int k = i; k < min(i + 8, end); ++k) logprobs[k] = T::to_f32(logits[k]) - lse;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (err_word && *err_word != 0u) tok = -1;  // the persistent launch gave up a bounded wait (sticky): no plausible-looking id from garbage
        *token = tok;
        if (state) {
            const int next_pos = state->pos + 1;  // the position the chosen token will occupy
            if (history && next_pos < hist_cap) history[next_pos] = tok;  // device-side token history (PromptCache.computed_ids)
            state->token = tok;   // greedy auto-feed of the next step
            state->pos = next_pos;  // cache.offset += 1 (reusable.py:139)
        }
    }
}

// `rows` logit vectors [rows, V] at once (the multi-sequence decode step): stats_buf = rows * TAIL_STAT_TILES partials of scratch.
static inline int logits_tail_rows_launch(int dtype, const u16 *logits, int V, int rows, LogitStat *stats_buf, float *logprobs, int *tokens,
                                          hipStream_t st) {
    if (dtype != PIE_BF16 && dtype != PIE_F16) return pie::fail(PIE_E_ARG, "logits tail: dtype must be PIE_BF16 or PIE_F16");
    const dim3 g1(TAIL_STAT_TILES, rows), g2(TAIL_FINISH_BLOCKS, rows);
    if (dtype == PIE_BF16) {
        hipLaunchKernelGGL(k_logits_stats<BF16>, g1, dim3(256), 0, st, logits, V, stats_buf);
        hipLaunchKernelGGL(k_logits_finish<BF16>, g2, dim3(256), 0, st, logits, V, stats_buf, TAIL_STAT_TILES, logprobs, tokens, nullptr, nullptr, 0, (const unsigned *)nullptr);
    } else {
        hipLaunchKernelGGL(k_logits_stats<F16>, g1, dim3(256), 0, st, logits, V, stats_buf);
        hipLaunchKernelGGL(k_logits_finish<F16>, g2, dim3(256), 0, st, logits, V, stats_buf, TAIL_STAT_TILES, logprobs, tokens, nullptr, nullptr, 0, (const unsigned *)nullptr);
    }
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

// stats == nullptr (op-level API): the per-tile partials are computed from the logits first, into
// stream-ordered scratch (hipMallocAsync; the decoder path passes its own stats and never allocates).
static inline int logits_tail_launch(int dtype, const u16 *logits, int V, const LogitStat *stats, int n_stats, float *logprobs,
                                     int *token, DecState *state, int *history, int hist_cap, hipStream_t st, const unsigned *err_word = nullptr) {
    if (dtype != PIE_BF16 && dtype != PIE_F16) return pie::fail(PIE_E_ARG, "logits tail: dtype must be PIE_BF16 or PIE_F16");
    if (stats && n_stats > TAIL_MAX_STATS) return pie::fail(PIE_E_SHAPE, "logits tail: too many partials");
    LogitStat *tmp = nullptr;
    if (!stats) {
        if (hipMallocAsync((void **)&tmp, sizeof(LogitStat) * TAIL_STAT_TILES, st) != hipSuccess)
            return pie::fail(PIE_E_HIP, "logits tail: hipMallocAsync failed");
        if (dtype == PIE_BF16) hipLaunchKernelGGL(k_logits_stats<BF16>, dim3(TAIL_STAT_TILES), dim3(256), 0, st, logits, V, tmp);
        else hipLaunchKernelGGL(k_logits_stats<F16>, dim3(TAIL_STAT_TILES), dim3(256), 0, st, logits, V, tmp);
        PIE_LAUNCH_CHECK();
        stats = tmp, n_stats = TAIL_STAT_TILES;
    }
    if (dtype == PIE_BF16)
        hipLaunchKernelGGL(k_logits_finish<BF16>, dim3(TAIL_FINISH_BLOCKS), dim3(256), 0, st, logits, V, stats, n_stats, logprobs, token, state, history, hist_cap, err_word);
    else
        hipLaunchKernelGGL(k_logits_finish<F16>, dim3(TAIL_FINISH_BLOCKS), dim3(256), 0, st, logits, V, stats, n_stats, logprobs, token, state, history, hist_cap, err_word);
    PIE_LAUNCH_CHECK();
    if (tmp) PIE_HIP_TRY(hipFreeAsync(tmp, st));
    return PIE_OK;
}
