// tp_comm.hip -- tensor-parallel communicator of the decode step: one-shot all-reduce over IPC-mapped peer buffers.
//
// The reference has no parallelism at all; BASELINE.json configs[4] asks for Llama-3-70B sharded over the 8 GPUs of a
// node.  A decode step then has 2 all-reduces of ONE hidden vector per layer (16-32 KiB of fp32, after o_proj and after down_proj)
// plus a 3-number exchange for the vocabulary-parallel tail: 161 tiny collectives per token, all latency.  A ring (RCCL's
// default shape) pays 2 (world - 1) hops per collective; xGMI is point to point with every peer one hop away, so this is the
// one-shot form: every rank PUSHES its vector into a slot of every peer's receive area and then sums the slots of its own area
// in rank order (identical fp32 result on every rank, no second hop).  NCCL-LL style transport: each fp32 travels as an 8-byte
// {value, epoch} granule written by ONE system-scope store, so the data is its own flag -- no fence, no separate flag hop; the
// receiver polls LOCAL memory.  Two receive areas alternate with the epoch's parity: a peer can run at most one collective
// ahead (it needs this rank's push to finish the current one).  The receive areas are fine-grained device memory exported
// with hipIpcGetMemHandle; the host exchanges the 64-byte handles (torch.distributed / any side channel) once.
// The epoch lives in device memory and is advanced by the kernels themselves, so a captured hipGraph replays correctly.
// Every spin is bounded; a give-up sets the communicator's error word.
//
// Round 4: (1) inside the decoder the PUSH half rides in the row-parallel GEMV's epilogue (EPI_TP_PUSH, w4_gemv.hpp: the lane's two fp32 row
// sums leave as granules straight from its registers -- no fp32 round trip through memory, and the transfer starts when a wave finishes,
// not when the launch does); the launch behind it (k_tp_allreduce<T, true>) only pulls, sums in rank order, rounds and adds the residual.
// (2) A second backend: PIE_COMM_RCCL runs the same collectives through RCCL (ncclAllReduce / ncclAllGather on the launch stream,
// capturable in the step's hipGraph) -- the comparator and fall-back for the first run on a real multi-GPU node.  RCCL's ring sums in its
// own order: every rank still gets identical bits, but they differ from the one-shot form's rank-order sum in the last fp32 bit.
//
// NOT measured on a multi-GPU node (this build pool hands out single-GPU boxes): the one-shot form is verified with two processes sharing
// one card (tests/test_gpu_tp.py), where "peer memory" is the same HBM; RCCL refuses two ranks on one device, so its backend is verified
// with a one-rank communicator (the whole tensor-parallel code path of the decoder, bit for bit against the unsharded decoder).
#include <cstring>
#include <new>
#include <vector>

#include <dlfcn.h>

#include "decoder.hpp"

constexpr int TP_MAX_WORLD = 8;
constexpr unsigned long long TP_SPIN_LIMIT = 200000000ull;  // s_memrealtime ticks (100 MHz): 2 s

// RCCL, loaded with dlopen at the first RCCL communicator (the copy already in the process -- torch's -- if there is one)
struct RcclId {
    char b[128];  // ncclUniqueId, passed by value
};
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *id) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, RcclId id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t st) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return PIE_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return pie::fail(PIE_E_STATE, std::string("RCCL backend: librccl.so cannot be loaded: ") + dlerror());
    RcclApi a;
    a.lib = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather) return pie::fail(PIE_E_STATE, "RCCL backend: librccl.so lacks an entry point");
    g_rccl = a;
    return PIE_OK;
}
#define PIE_RCCL_TRY(expr)                                                                                                                   \
    do {                                                                                                                                     \
        const int r_ = (expr);                                                                                                               \
        if (r_ != 0) return pie::fail(PIE_E_HIP, std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error")); \
    } while (0)
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0;  // rccl.h: ncclFloat32, ncclSum

struct pie_comm {
    int rank = 0, world = 1;
    int backend = PIE_COMM_IPC;
    void *rccl = nullptr;        // ncclComm_t (PIE_COMM_RCCL)
    float *gather = nullptr;     // PIE_COMM_RCCL: [4] this rank's (max, sum exp, argmax, -) | [world][4] everybody's
    size_t max_elems = 0;
    unsigned long long *recv = nullptr;       // [2][world][max_elems + 8] granules: slot r of parity p at (p * world + r) * stride
    unsigned long long *peer[TP_MAX_WORLD] = {};  // every rank's recv (own pointer at [rank]); host copies of the mapped pointers
    unsigned long long **peer_dev = nullptr;  // the same table in device memory
    unsigned *epoch = nullptr;                // device: collectives completed so far; [1] = error word
    bool connected = false;
};

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ void tp_push(unsigned long long *p, float v, unsigned e) {
    __hip_atomic_store((gu64 *)(unsigned long long)p, ((unsigned long long)e << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool tp_pull(const unsigned long long *p, unsigned e, float &v, unsigned long long t0, unsigned *err) {
    for (;;) {
        const unsigned long long g = __hip_atomic_load((const gu64 *)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(g >> 32) == e) {
            v = __builtin_bit_cast(float, (unsigned)g);
            return true;
        }
        // after one give-up the communicator is dead: later waits fall through at once instead of stacking 2 s each
        if (__builtin_amdgcn_s_memrealtime() - t0 > TP_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(err, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = 0.0f;
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// data[n] (this rank's fp32 partial) -> the sum over the ranks, in rank order:
//   resid == nullptr: written back to data (pie_allreduce_f32);
//   resid != nullptr: the row-parallel Linear's tail, h = T(h + T(sum)) -- the Linear's one rounding, then the residual add
//   (language.py:151,153; proxy_inference_engine_amd/tp.py: TPLlama._row_parallel).
// One thread per PAIR of elements, n / 2048 workgroups (4 for H = 8192): every thread pushes its two values to all peers, then pulls
// the same two positions of all ranks' slots.  The last workgroup to finish advances the epoch (arrival counter in epoch[2]).
// PUSHED: the producer (the row-parallel GEMV's EPI_TP_PUSH epilogue) has already pushed this rank's values; only the pull half runs here.
template <class T, bool PUSHED>
__global__ void __launch_bounds__(1024) k_tp_allreduce(unsigned long long *const *peers, unsigned *epoch, int rank, int world, size_t stride, float *data,
                                                       int n, u16 *resid) {
    const unsigned e = epoch[0] + 1;
    const size_t slot = ((size_t)(e & 1) * world + rank) * stride;
    const int i = 2 * (blockIdx.x * 1024 + threadIdx.x);
    if (i < n) {
        if (!PUSHED) {
            const float v0 = data[i], v1 = i + 1 < n ? data[i + 1] : 0.0f;
            for (int r = 0; r < world; ++r) {
                tp_push(peers[r] + slot + i, v0, e);
                if (i + 1 < n) tp_push(peers[r] + slot + i + 1, v1, e);
            }
        }
        const unsigned long long *mine = peers[rank] + (size_t)(e & 1) * world * stride;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        float s0 = 0.0f, s1 = 0.0f;
        for (int r = 0; r < world; ++r) {
            float a, b = 0.0f;
            tp_pull(mine + (size_t)r * stride + i, e, a, t0, epoch + 1);
            if (i + 1 < n) tp_pull(mine + (size_t)r * stride + i + 1, e, b, t0, epoch + 1);
            s0 = r ? s0 + a : a, s1 = r ? s1 + b : b;
        }
        if (resid) {  // n is even here (launcher)
            const u32 h2 = *reinterpret_cast<const u32 *>(resid + i);
            *reinterpret_cast<u32 *>(resid + i) = pack2<T>(lo_f32<T>(h2) + round_T<T>(s0), hi_f32<T>(h2) + round_T<T>(s1));
        } else {
            data[i] = s0;
            if (i + 1 < n) data[i + 1] = s1;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(epoch + 2, 1u) == gridDim.x - 1) {  // every workgroup read epoch[0] before its own arrival
            epoch[2] = 0;
            __hip_atomic_store(epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Global log-sum-exp and greedy token from every shard's (max, sum exp, first argmax + vocabulary offset) triple (ties: the lowest vocabulary
// index, like mx.argmax); advances the device-side decode state exactly as k_logits_finish does.  One thread.
__device__ __forceinline__ void tp_merge_publish(const float *gm, const float *gs, const float *ga, int world, const unsigned *err_word, float *lse_out, int *token,
                                                 DecState *state, int *history, int hist_cap) {
    float GM = -INFINITY;
    for (int r = 0; r < world; ++r) GM = fmaxf(GM, gm[r]);
    float S = 0.0f;
    int gtok = 0x7fffffff;
    for (int r = 0; r < world; ++r) {
        S += gs[r] > 0.0f ? gs[r] * expf(gm[r] - GM) : 0.0f;
        if (gm[r] == GM) gtok = min(gtok, __float_as_int(ga[r]));
    }
    *lse_out = GM + logf(S);
    // a collective of this step gave up (sticky error word): the sums are not sums -- the host sees token -1 instead of a plausible id
    if (err_word && __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) gtok = -1;
    *token = gtok;
    if (state) {
        const int next_pos = state->pos + 1;
        if (history && next_pos < hist_cap) history[next_pos] = gtok;
        state->token = gtok;
        state->pos = next_pos;
    }
}

// Vocabulary-parallel tail, part 1 (one workgroup of 256): merge this rank's per-wave partials of the lm_head GEMV into its triple, then
//   RCCL == false: exchange the triples with the peers as granules and publish the global log-sum-exp and token;
//   RCCL == true:  leave the triple in `gather` for ncclAllGather; k_tp_tail_merge publishes.
template <bool RCCL>
__global__ void __launch_bounds__(256) k_tp_tail_stats(unsigned long long *const *peers, unsigned *epoch, int rank, int world, size_t stride,
                                                       const LogitStat *stats, int n_stats, int vocab_offset, float *lse_out, int *token, DecState *state,
                                                       int *history, int hist_cap, float *gather) {
    __shared__ float s_max[4], s_sum[4];
    __shared__ int s_arg[4];
    LogitStat st[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = threadIdx.x + 256 * k;
        const uint4 raw = *reinterpret_cast<const uint4 *>(stats + (i < n_stats ? i : n_stats - 1));
        st[k].max = i < n_stats ? __uint_as_float(raw.x) : -INFINITY;
        st[k].sumexp = i < n_stats ? __uint_as_float(raw.y) : 0.0f;
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
    if (threadIdx.x == 0) {
        const float local_se = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        if (RCCL) {
            gather[0] = M, gather[1] = local_se, gather[2] = __int_as_float(tok + vocab_offset), gather[3] = 0.0f;
            return;
        }
        const unsigned e = epoch[0] + 1;
        // the three numbers ride in the last 8 granules of this rank's slot (behind the hidden-vector area)
        const size_t slot = ((size_t)(e & 1) * world + rank) * stride + (stride - 8);
        for (int r = 0; r < world; ++r) {
            tp_push(peers[r] + slot + 0, M, e);
            tp_push(peers[r] + slot + 1, local_se, e);
            tp_push(peers[r] + slot + 2, __int_as_float(tok + vocab_offset), e);
        }
        const unsigned long long *mine = peers[rank] + (size_t)(e & 1) * world * stride + (stride - 8);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        float gm[TP_MAX_WORLD], gs[TP_MAX_WORLD], ga[TP_MAX_WORLD];
        for (int r = 0; r < world; ++r) {
            tp_pull(mine + (size_t)r * stride + 0, e, gm[r], t0, epoch + 1);
            tp_pull(mine + (size_t)r * stride + 1, e, gs[r], t0, epoch + 1);
            tp_pull(mine + (size_t)r * stride + 2, e, ga[r], t0, epoch + 1);
        }
        tp_merge_publish(gm, gs, ga, world, epoch + 1, lse_out, token, state, history, hist_cap);
        epoch[0] = e;
    }
}
// PIE_COMM_RCCL: the gathered triples [world][4] -> the same publication
__global__ void k_tp_tail_merge(const float *gathered, int world, float *lse_out, int *token, DecState *state, int *history, int hist_cap) {
    float gm[TP_MAX_WORLD], gs[TP_MAX_WORLD], ga[TP_MAX_WORLD];
    for (int r = 0; r < world; ++r) gm[r] = gathered[4 * r], gs[r] = gathered[4 * r + 1], ga[r] = gathered[4 * r + 2];
    tp_merge_publish(gm, gs, ga, world, nullptr, lse_out, token, state, history, hist_cap);
}
// PIE_COMM_RCCL: the row-parallel Linear's tail after ncclAllReduce, h = T(h + T(sum)) (language.py:151,153)
template <class T>
__global__ void __launch_bounds__(256) k_tp_round_residual(const float *sum, int n, u16 *resid) {
    const int i = 2 * (blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const float2 s2 = *reinterpret_cast<const float2 *>(sum + i);
    const u32 h2 = *reinterpret_cast<const u32 *>(resid + i);
    *reinterpret_cast<u32 *>(resid + i) = pack2<T>(lo_f32<T>(h2) + round_T<T>(s2.x), hi_f32<T>(h2) + round_T<T>(s2.y));
}

// part 2: this rank's slice of the log-probabilities, logits - lse
template <class T>
__global__ void __launch_bounds__(256) k_tp_tail_logprobs(const u16 *logits, int V, const float *lse, float *logprobs) {
    const float l = *lse;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 2; i < V; i += gridDim.x * 512) {
        const u32 w = *reinterpret_cast<const u32 *>(logits + i);
        logprobs[i] = lo_f32<T>(w) - l;
        if (i + 1 < V) logprobs[i + 1] = hi_f32<T>(w) - l;
    }
}

}  // namespace

// pushed: the producing GEMV's EPI_TP_PUSH epilogue already sent this rank's values (one-shot backend only; tp_comm_push_args)
int tp_allreduce_launch(pie_comm *c, int dtype, float *data, int n, u16 *resid, hipStream_t st, bool pushed) {
    PIE_REQUIRE(c && c->connected, PIE_E_STATE, "tensor-parallel communicator is not connected (pie_comm_connect)");
    PIE_REQUIRE(n > 0 && (size_t)n <= c->max_elems && (!resid || n % 2 == 0), PIE_E_SHAPE, "tp all-reduce: vector longer than the communicator's slots");
    PIE_REQUIRE(data || (pushed && resid), PIE_E_ARG, "tp all-reduce: null data");
    if (c->backend == PIE_COMM_RCCL) {
        PIE_REQUIRE(!pushed && data, PIE_E_STATE, "tp all-reduce: the RCCL backend reduces the fp32 partial in memory");
        PIE_RCCL_TRY(g_rccl.AllReduce(data, data, (size_t)n, RCCL_FLOAT32, RCCL_SUM, c->rccl, st));
        if (!resid) return PIE_OK;
        const dim3 grid((unsigned)((n / 2 + 255) / 256));
        if (dtype == PIE_F16) hipLaunchKernelGGL(k_tp_round_residual<F16>, grid, dim3(256), 0, st, data, n, resid);
        else hipLaunchKernelGGL(k_tp_round_residual<BF16>, grid, dim3(256), 0, st, data, n, resid);
        PIE_LAUNCH_CHECK();
        return PIE_OK;
    }
    const size_t stride = c->max_elems + 8;
    const dim3 grid((unsigned)((n + 2047) / 2048));
#define TP_AR(TT, P) hipLaunchKernelGGL((k_tp_allreduce<TT, P>), grid, dim3(1024), 0, st, c->peer_dev, c->epoch, c->rank, c->world, stride, data, n, resid)
    if (dtype == PIE_F16) {
        if (pushed) TP_AR(F16, true);
        else TP_AR(F16, false);
    } else {
        if (pushed) TP_AR(BF16, true);
        else TP_AR(BF16, false);
    }
#undef TP_AR
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

// What the row-parallel GEMV's EPI_TP_PUSH epilogue needs; false when this communicator's collectives do not run on granules (RCCL)
bool tp_comm_push_args(const pie_comm *c, unsigned long long *const **peers, const unsigned **epoch, unsigned *stride) {
    if (!c || c->backend != PIE_COMM_IPC || !c->connected) return false;
    *peers = c->peer_dev, *epoch = c->epoch, *stride = (unsigned)(c->max_elems + 8);
    return true;
}

int tp_tail_launch(pie_comm *c, int dtype, const u16 *logits, int V_local, int vocab_offset, const LogitStat *stats, int n_stats, float *lse, float *logprobs,
                   int *token, DecState *state, int *history, int hist_cap, hipStream_t st) {
    PIE_REQUIRE(c && c->connected, PIE_E_STATE, "tensor-parallel communicator is not connected (pie_comm_connect)");
    PIE_REQUIRE(n_stats <= TAIL_MAX_STATS, PIE_E_SHAPE, "tp tail: too many partials");
    const size_t stride = c->max_elems + 8;
    if (c->backend == PIE_COMM_RCCL) {
        hipLaunchKernelGGL(k_tp_tail_stats<true>, dim3(1), dim3(256), 0, st, c->peer_dev, c->epoch, c->rank, c->world, stride, stats, n_stats, vocab_offset, lse, token,
                           state, history, hist_cap, c->gather);
        PIE_LAUNCH_CHECK();
        PIE_RCCL_TRY(g_rccl.AllGather(c->gather, c->gather + 4, 4, RCCL_FLOAT32, c->rccl, st));
        hipLaunchKernelGGL(k_tp_tail_merge, dim3(1), dim3(1), 0, st, c->gather + 4, c->world, lse, token, state, history, hist_cap);
    } else {
        hipLaunchKernelGGL(k_tp_tail_stats<false>, dim3(1), dim3(256), 0, st, c->peer_dev, c->epoch, c->rank, c->world, stride, stats, n_stats, vocab_offset, lse, token,
                           state, history, hist_cap, (float *)nullptr);
    }
    PIE_LAUNCH_CHECK();
    const dim3 grid((unsigned)((V_local + 511) / 512 < 64 ? (V_local + 511) / 512 : 64));
    if (dtype == PIE_F16) hipLaunchKernelGGL(k_tp_tail_logprobs<F16>, grid, dim3(256), 0, st, logits, V_local, lse, logprobs);
    else hipLaunchKernelGGL(k_tp_tail_logprobs<BF16>, grid, dim3(256), 0, st, logits, V_local, lse, logprobs);
    PIE_LAUNCH_CHECK();
    return PIE_OK;
}

int tp_comm_geometry(const pie_comm *c, int *rank, int *world, size_t *max_elems) {
    *rank = c->rank, *world = c->world, *max_elems = c->max_elems;
    return c->connected ? PIE_OK : PIE_E_STATE;
}

extern "C" {

int pie_comm_create(int rank, int world, size_t max_elems, pie_comm **out) {
    PIE_REQUIRE(out, PIE_E_ARG, "pie_comm_create: null pointer");
    PIE_REQUIRE(world >= 1 && world <= TP_MAX_WORLD && rank >= 0 && rank < world, PIE_E_ARG, "pie_comm_create: 1 <= world <= 8, 0 <= rank < world");
    PIE_REQUIRE(max_elems > 0 && max_elems <= (1u << 20), PIE_E_SHAPE, "pie_comm_create: max_elems out of range");
    pie_comm *c = new (std::nothrow) pie_comm();
    PIE_REQUIRE(c, PIE_E_HIP, "pie_comm_create: out of host memory");
    c->rank = rank, c->world = world, c->max_elems = max_elems;
    const size_t bytes = 2 * (size_t)world * (max_elems + 8) * sizeof(unsigned long long);
    // fine-grained: peers write it over xGMI while this GPU polls it (coarse-grained memory is only coherent at kernel boundaries)
    hipError_t e = hipExtMallocWithFlags((void **)&c->recv, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        // A peer's stores into coarse-grained memory are not guaranteed visible to a kernel that is polling it: every collective
        // would spin for its whole bounded wait and then reduce zeros.  Only a single-rank communicator (nobody writes remotely) may
        // fall back; with peers this is an error at creation, not a 2-second timeout per step later.
        if (world > 1) {
            (void)pie_comm_destroy(c);
            return pie::fail(PIE_E_HIP, "pie_comm_create: fine-grained device memory (hipDeviceMallocFinegrained) is unavailable; peers' stores would not be visible to a polling kernel");
        }
        e = hipMalloc((void **)&c->recv, bytes);
    }
    if (e != hipSuccess || hipMemset(c->recv, 0, bytes) != hipSuccess || hipMalloc((void **)&c->epoch, 16) != hipSuccess ||
        hipMemset(c->epoch, 0, 16) != hipSuccess || hipMalloc((void **)&c->peer_dev, sizeof(void *) * TP_MAX_WORLD) != hipSuccess) {
        (void)pie_comm_destroy(c);
        return pie::fail(PIE_E_HIP, "pie_comm_create: device allocation failed");
    }
    c->peer[rank] = c->recv;
    if (world == 1) {
        PIE_HIP_TRY(hipMemcpy(c->peer_dev, c->peer, sizeof(void *) * TP_MAX_WORLD, hipMemcpyHostToDevice));
        c->connected = true;
    }
    *out = c;
    return PIE_OK;
}

int pie_comm_rccl_unique_id(void *id128) {
    PIE_REQUIRE(id128, PIE_E_ARG, "pie_comm_rccl_unique_id: null pointer");
    int rc = rccl_load();
    if (rc) return rc;
    PIE_RCCL_TRY(g_rccl.GetUniqueId((RcclId *)id128));
    return PIE_OK;
}

int pie_comm_create_rccl(int rank, int world, size_t max_elems, const void *id128, pie_comm **out) {
    PIE_REQUIRE(out && id128, PIE_E_ARG, "pie_comm_create_rccl: null pointer");
    PIE_REQUIRE(world >= 1 && world <= TP_MAX_WORLD && rank >= 0 && rank < world, PIE_E_ARG, "pie_comm_create_rccl: 1 <= world <= 8, 0 <= rank < world");
    PIE_REQUIRE(max_elems > 0 && max_elems <= (1u << 20), PIE_E_SHAPE, "pie_comm_create_rccl: max_elems out of range");
    int rc = rccl_load();
    if (rc) return rc;
    pie_comm *c = new (std::nothrow) pie_comm();
    PIE_REQUIRE(c, PIE_E_HIP, "pie_comm_create_rccl: out of host memory");
    c->rank = rank, c->world = world, c->max_elems = max_elems, c->backend = PIE_COMM_RCCL;
    if (hipMalloc((void **)&c->epoch, 16) != hipSuccess || hipMemset(c->epoch, 0, 16) != hipSuccess ||
        hipMalloc((void **)&c->gather, sizeof(float) * 4 * (TP_MAX_WORLD + 1)) != hipSuccess) {
        (void)pie_comm_destroy(c);
        return pie::fail(PIE_E_HIP, "pie_comm_create_rccl: device allocation failed");
    }
    RcclId id;
    memcpy(&id, id128, sizeof id);
    const int r = g_rccl.CommInitRank(&c->rccl, world, id, rank);  // collective: every rank of the world calls it with the same id
    if (r != 0) {
        (void)pie_comm_destroy(c);
        return pie::fail(PIE_E_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
    }
    c->connected = true;
    *out = c;
    return PIE_OK;
}

int pie_comm_export(const pie_comm *c, void *handle64) {
    PIE_REQUIRE(c && c->backend == PIE_COMM_IPC, PIE_E_STATE, "pie_comm_export: only the one-shot (IPC) communicator has a receive area to export");
    PIE_REQUIRE(c && handle64, PIE_E_ARG, "pie_comm_export: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    PIE_HIP_TRY(hipIpcGetMemHandle(&h, c->recv));
    memcpy(handle64, &h, 64);
    return PIE_OK;
}

int pie_comm_connect(pie_comm *c, const void *handles) {
    PIE_REQUIRE(c && handles, PIE_E_ARG, "pie_comm_connect: null pointer");
    PIE_REQUIRE(c->backend == PIE_COMM_IPC, PIE_E_STATE, "pie_comm_connect: an RCCL communicator is connected by its creation");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + 64 * r, 64);
        void *p = nullptr;
        PIE_HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer[r] = (unsigned long long *)p;
    }
    PIE_HIP_TRY(hipMemcpy(c->peer_dev, c->peer, sizeof(void *) * TP_MAX_WORLD, hipMemcpyHostToDevice));
    c->connected = true;
    return PIE_OK;
}

int pie_comm_destroy(pie_comm *c) {
    if (!c) return PIE_OK;
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->rccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->rccl);
    if (c->gather) (void)hipFree(c->gather);
    if (c->recv) (void)hipFree(c->recv);
    if (c->epoch) (void)hipFree(c->epoch);
    if (c->peer_dev) (void)hipFree(c->peer_dev);
    delete c;
    return PIE_OK;
}

int pie_allreduce_f32(pie_comm *c, float *data, size_t n, void *stream) {
    PIE_REQUIRE(c && data, PIE_E_ARG, "pie_allreduce_f32: null pointer");
    return tp_allreduce_launch(c, PIE_BF16, data, (int)n, nullptr, (hipStream_t)stream, false);
}

int pie_comm_status(pie_comm *c, unsigned *error) {
    PIE_REQUIRE(c && error, PIE_E_ARG, "pie_comm_status: null pointer");
    PIE_HIP_TRY(hipDeviceSynchronize());
    PIE_HIP_TRY(hipMemcpy(error, c->epoch + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
    return PIE_OK;
}

}  // extern "C"
