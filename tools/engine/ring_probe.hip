// tools/ring_probe.hip -- developer probe of the LDS-DMA weight ring (csrc/ring.hpp): one loader wave + three consumer waves per
// CU stream R distinct [N, K] W4S matrices back to back (no dependency edges) and the row sums are compared bit for bit with a
// plain one-wave-per-row-pair kernel.  Answers, before the engine is built on it: does LDS-DMA reach destinations above 64 KiB,
// does the FULL / FREE handshake hold, and what does the ring stream at with int4 consumers.
//   ring_probe [--n N] [--k K] [--reps R] [--iters I]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ring_probe.hip -Iinclude -o tools/ring_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ring_gemv.hpp"  // build with -Iproxy_inference_engine_amd/csrc -Itools/engine

namespace pie {
void set_error(const std::string &) {}
int fail(int code, const std::string &) { return code; }
}  // namespace pie

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

__global__ void k_fill_units(u32 *w, size_t n_units, unsigned seed) {  // random codes, tame {scale, bias}
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // dword index
    if (i >= n_units * (W4S_UNIT_BYTES / 4)) return;
    unsigned x = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u ^ seed * 0x9E3779B9u;
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    const size_t in_unit = i % (W4S_UNIT_BYTES / 4);
    if (in_unit >= 512) {  // scale in [2^-7, 2^-6), bias in (-2^-4, 0]: bf16 bit patterns
        const unsigned sc = 0x3C00u | (x & 0x7Fu), bi = 0xBD00u | ((x >> 8) & 0x7Fu);
        x = sc | (bi << 16);
    }
    w[i] = x;
}
__global__ void k_fill_x(u16 *x, int K, unsigned seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16, h *= 0x7feb352du, h ^= h >> 15;
    const float f = ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f);
    const unsigned u = __float_as_uint(f);
    x[i] = (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

struct ProbeArgs {
    const char *w;       // R matrices back to back
    size_t mat_bytes;
    int n_pairs, ns, K, reps;
    const u16 *x;
    float *y;            // [reps][2 * n_pairs]
    unsigned *err;
    int abl;  // 1: consumers only poll and release (no LDS reads, no math); 2: every matrix re-read from the first one (cache-resident stream)
};

// the x image of the launched GEMV (w4_gemv.hpp: [8 pieces][groups | 1][16 B], pre-scaled) + group sums
template <class T>
__device__ void stage_x(char *smem, const GemvLds &L, const u16 *x, int K, int tid, int nt) {
    float *sxs = reinterpret_cast<float *>(smem + L.off_sx);
    const int n_pieces = K >> 3;
    for (int j = tid; j < ((n_pieces + 63) & ~63); j += nt) {
        const bool ok = j < n_pieces;
        const uint4 v = reinterpret_cast<const uint4 *>(x)[ok ? j : n_pieces - 1];
        float ps = ok ? sum8<T>(v) : 0.0f;
        ps = lanes8_sum(ps);
        if (ok) {
            *reinterpret_cast<uint4 *>(smem + ((size_t)(j & 7) * L.stride + (j >> 3)) * 16) = scale8<T>(v);
            if ((j & 7) == 0) sxs[j >> 3] = ps;
        }
    }
}

template <class T>
__global__ void __launch_bounds__(RING_WAVES * 64, 1) k_ring_probe(const ProbeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const GemvLds L = gemv_lds(a.K);
    const unsigned lds0 = lds_addr_of(smem);
    const unsigned img_bytes = (unsigned)((L.total + 15) & ~15);
    const unsigned ctl = lds0 + img_bytes, ring = ctl + 64 + RING_CONSUMERS * 2 * GEMV_MAX_RUN * 4;
    if (threadIdx.x < 16) lds_st(ctl + 4 * threadIdx.x, 0u);
    stage_x<T>(smem, L, a.x, a.K, threadIdx.x, RING_WAVES * 64);
    __syncthreads();
    const int n_cus = gridDim.x, cu = blockIdx.x;
    if (wave < RING_LOADERS) {
        auto next = [&](int w, int m, const char **p, unsigned *bytes) -> bool {
            if (a.abl & 16) {  // slot-interleaved: the chip's streams form one compact window sweeping through the matrix (two K slices only)
                const int S = n_cus * RING_CONSUMERS, n_slots = a.n_pairs / 2, kmax = (n_slots + S - 1) / S;
                const int rep = m / kmax, k = m % kmax, slot = k * S + ((a.abl & 32) ? w * n_cus + cu : cu * RING_CONSUMERS + w);
                if (rep >= a.reps) return false;
                *p = a.w + (size_t)rep * a.mat_bytes + (size_t)slot * RING_SLOT_BYTES;
                *bytes = slot < n_slots ? (unsigned)RING_SLOT_BYTES : 0u;
                return true;
            }
            if (m >= a.reps) return false;
            const RingRun r = ring_run(a.n_pairs, 1, n_cus, cu, w);
            *p = a.w + (size_t)((a.abl & 2) ? 0 : m) * a.mat_bytes + (size_t)r.first * a.ns * W4S_UNIT_BYTES;
            *bytes = (unsigned)(r.count * a.ns * W4S_UNIT_BYTES);
            return true;
        };
        const unsigned long long dl = __builtin_amdgcn_s_memrealtime() + 300000000ull;
        if (a.abl & 4) __builtin_amdgcn_s_setprio(3);
        if (wave == 0) ring_loader<0>(ring, ctl, lane, next, dl);
        if (RING_LOADERS > 1 && wave == 1) ring_loader<(RING_LOADERS > 1 ? 1 : 0)>(ring, ctl, lane, next, dl);
        return;
    }
    const int cw = wave - RING_LOADERS;
    RingCursor cur = ring_cursor(ring, ctl, cw);
    RingRun r = ring_run(a.n_pairs, 1, n_cus, cu, cw);
    const int S = n_cus * RING_CONSUMERS, sidx = (a.abl & 32) ? cw * n_cus + cu : cu * RING_CONSUMERS + cw;
    if (a.abl & 16) {
        const int n_slots = a.n_pairs / 2;
        r.first = 0, r.count = sidx < n_slots ? 2 * ((n_slots - sidx + S - 1) / S) : 0;
    }
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + 200000000ull;  // 2 s
    float *outp = reinterpret_cast<float *>(smem + img_bytes + 64) + cw * 2 * GEMV_MAX_RUN;
    bool ok = true;
    for (int rep = 0; rep < a.reps && r.count > 0 && ok; ++rep) {
        if (a.abl & 1) {
            const int n_units = r.count * a.ns;
            for (int u = 0; u < n_units && ok; u += RING_SLOT_UNITS) {
                ring_wait_slot(cur, deadline, ok);
                ring_release_slot(cur);
            }
            continue;
        }
        if (a.ns == 2 && !(a.abl & 8)) ok = ring_consume<T, 2>(cur, smem, lds0, smem, L, a.K, r.count, outp, lane, deadline);
        else ok = ring_consume<T, 0>(cur, smem, lds0, smem, L, a.K, r.count, outp, lane, deadline);
        const int gpair = (a.abl & 16) ? 2 * ((lane >> 1) * S + sidx) + (lane & 1) : r.first + lane;
        if (lane < r.count) *reinterpret_cast<float2 *>(a.y + (size_t)rep * 2 * a.n_pairs + 2 * gpair) = *reinterpret_cast<const float2 *>(outp + 2 * lane);
    }
    if (!ok && lane == 0) atomicAdd(a.err, 1u);
}

// reference: one wave per (matrix, row pair), units straight from global memory, the same arithmetic
template <class T>
__global__ void __launch_bounds__(256) k_ref(const ProbeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemvLds L = gemv_lds(a.K);
    stage_x<T>(smem, L, a.x, a.K, threadIdx.x, 256);
    __syncthreads();
    const float *sxs = reinterpret_cast<const float *>(smem + L.off_sx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_groups = a.K >> 6;
    for (size_t job = (size_t)blockIdx.x * 4 + wave; job < (size_t)a.reps * a.n_pairs; job += (size_t)gridDim.x * 4) {
        const int rep = (int)(job / a.n_pairs), pair = (int)(job % a.n_pairs);
        const char *base = a.w + (size_t)rep * a.mat_bytes + (size_t)pair * a.ns * W4S_UNIT_BYTES;
        float acc = 0.0f;
        for (int sl = 0; sl < a.ns; ++sl) {
            const char *p = base + (size_t)sl * W4S_UNIT_BYTES;
            const uint4 c0 = *reinterpret_cast<const uint4 *>(p + lane * 16), c1 = *reinterpret_cast<const uint4 *>(p + 1024 + lane * 16);
            const u32 sb = *reinterpret_cast<const u32 *>(p + 2048 + lane * 4);
            const int g = sl * 32 + (lane & 31);
            const bool gvalid = g < n_groups;
            const int gc = gvalid ? g : n_groups - 1;
            u32 xr[32];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint4 v = *reinterpret_cast<const uint4 *>(smem + ((size_t)q * L.stride + gc) * 16);
                xr[4 * q + 0] = v.x, xr[4 * q + 1] = v.y, xr[4 * q + 2] = v.z, xr[4 * q + 3] = v.w;
            }
            const float sx = sxs[gc];
            const float dd = w4s_unit_dot<T>(c0, c1, xr);
            const float pr = fmaf(lo_f32<T>(sb), dd * T::DSCALE - T::OFFSET * sx, hi_f32<T>(sb) * sx);
            acc += gvalid ? pr : 0.0f;
        }
        const float tot = half_wave_sum(acc);
        if ((lane & 31) == 31) a.y[(size_t)rep * 2 * a.n_pairs + 2 * pair + (lane >> 5)] = tot;
    }
}

int main(int argc, char **argv) {
    int N = 28672, K = 4096, reps = 8, iters = 5, abl = 0;
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : (char *)"0"; };
        if (s == "--n") N = atoi(next());
        else if (s == "--k") K = atoi(next());
        else if (s == "--reps") reps = atoi(next());
        else if (s == "--iters") iters = atoi(next());
        else if (s == "--abl") abl = atoi(next());
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    const int n_pairs = N / 2, ns = w4s_slices(K);
    const size_t mat_bytes = (size_t)n_pairs * ns * W4S_UNIT_BYTES;
    printf("ring probe: %d CUs, [%d, %d] x %d matrices, %.1f MB each; %d loaders, %d consumers x %d slots, %d fills in flight\n", n_cus, N, K, reps, mat_bytes / 1e6, RING_LOADERS, RING_CONSUMERS, RING_SLOTS, RING_INFLIGHT);
    char *w;
    u16 *x;
    float *y, *yref;
    unsigned *err;
    CK(hipMalloc(&w, mat_bytes * reps));
    CK(hipMalloc(&x, K * 2));
    CK(hipMalloc(&y, (size_t)reps * N * 4));
    CK(hipMalloc(&yref, (size_t)reps * N * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    CK(hipMemset(y, 0xFF, (size_t)reps * N * 4));
    const size_t n_units = (size_t)n_pairs * ns * reps, n_dw = n_units * (W4S_UNIT_BYTES / 4);
    hipLaunchKernelGGL(k_fill_units, dim3((unsigned)((n_dw + 255) / 256)), dim3(256), 0, 0, (u32 *)w, n_units, 12345u);
    hipLaunchKernelGGL(k_fill_x, dim3((K + 255) / 256), dim3(256), 0, 0, x, K, 777u);
    CK(hipDeviceSynchronize());
    ProbeArgs a = {w, mat_bytes, n_pairs, ns, K, reps, x, y, err, abl};
    const GemvLds L = gemv_lds(K);
    const unsigned lds = (unsigned)((L.total + 15) & ~15) + 64 + RING_CONSUMERS * 2 * GEMV_MAX_RUN * 4 + RING_BYTES;
    printf("dynamic LDS %u bytes (x image %d, ring %d)\n", lds, L.total, RING_BYTES);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_probe<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ref<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    ProbeArgs ar = a;
    ar.y = yref, ar.abl = 0;
    hipLaunchKernelGGL(k_ref<BF16>, dim3(n_cus * 4), dim3(256), (unsigned)L.total, 0, ar);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_ring_probe<BF16>, dim3(n_cus), dim3(RING_WAVES * 64), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  iter %d: %.3f ms  %.2f TB/s\n", it, ms, (double)mat_bytes * reps / ms / 1e9);
    }
    unsigned herr = 0;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<float> hy((size_t)reps * N), hr((size_t)reps * N);
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), yref, hr.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < hy.size(); ++i) bad += memcmp(&hy[i], &hr[i], 4) != 0;
    printf("give-ups %u, differing row sums %zu / %zu  (y[0] = %g, ref %g)\n", herr, bad, hy.size(), hy[0], hr[0]);
    printf(bad || herr ? "PROBE FAILED\n" : "probe ok\n");
    return bad || herr ? 1 : 0;
}
