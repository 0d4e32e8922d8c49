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

template <class T>
__global__ void __launch_bounds__(256) k_logits_finish(const u16 *logits, int V, const LogitStat *stats, int n_stats, float *logprobs,
                                                       int *token, DecState *state, int *history, int hist_cap) {
    __shared__ float s_max[4], s_sum[4];
    __shared__ int s_arg[4];
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    for (int i = threadIdx.x; i < n_stats; i += 256) {
        const LogitStat st = stats[i];
        if (st.max > mx || (st.max == mx && st.argmax < arg)) mx = st.max, arg = st.argmax;
    }
    const float wmax = wave_max(mx);
    int cand = (mx == wmax) ? arg : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_max[wave] = wmax, s_arg[wave] = cand;
    __syncthreads();
    float M = s_max[0];
    int tok = s_arg[0];
    for (int w = 1; w < 4; ++w)
        if (s_max[w] > M || (s_max[w] == M && s_arg[w] < tok)) M = s_max[w], tok = s_arg[w];
    float se = 0.0f;
    for (int i = threadIdx.x; i < n_stats; i += 256) {
        const LogitStat st = stats[i];
        se += st.sumexp > 0.0f ? st.sumexp * expf(st.max - M) : 0.0f;
    }
    se = wave_sum(se);
    if ((threadIdx.x & 63) == 0) s_sum[wave] = se;
    __syncthreads();
    const float lse = M + logf(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
    const int slice = (V + gridDim.x - 1) / gridDim.x;
    const int begin = blockIdx.x * slice, end = min(V, begin + slice);
    for (int i = begin + threadIdx.x; i < end; i += 256) logprobs[i] = T::to_f32(logits[i]) - lse;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *token = tok;
        if (state) {
            const int next_pos = state->pos + 1;  // the position the chosen token will occupy
            if (history && next_pos < hist_cap) history[next_pos] = tok;  // device-side token history (PromptCache.computed_ids)
            state->token = tok;   // greedy auto-feed of the next step
            state->pos = next_pos;  // cache.offset += 1 (reusable.py:139)
        }
    }
}

// stats == nullptr (op-level API): the per-tile partials are computed from the logits first, into
// stream-ordered scratch (hipMallocAsync; the decoder path passes its own stats and never allocates).
static inline int logits_tail_launch(int dtype, const u16 *logits, int V, const LogitStat *stats, int n_stats, float *logprobs,
                                     int *token, DecState *state, int *history, int hist_cap, hipStream_t st) {
    if (dtype != PIE_BF16 && dtype != PIE_F16) return pie::fail(PIE_E_ARG, "logits tail: dtype must be PIE_BF16 or PIE_F16");
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
        hipLaunchKernelGGL(k_logits_finish<BF16>, dim3(TAIL_FINISH_BLOCKS), dim3(256), 0, st, logits, V, stats, n_stats, logprobs, token, state, history, hist_cap);
    else
        hipLaunchKernelGGL(k_logits_finish<F16>, dim3(TAIL_FINISH_BLOCKS), dim3(256), 0, st, logits, V, stats, n_stats, logprobs, token, state, history, hist_cap);
    PIE_LAUNCH_CHECK();
    if (tmp) PIE_HIP_TRY(hipFreeAsync(tmp, st));
    return PIE_OK;
}
