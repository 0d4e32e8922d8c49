// tools/pilot_probe.hip -- measurements behind the weight pilot (developer tool, not part of the product).
//
// Questions, one section each:
//   A. Is the workgroup -> XCD map a fixed function of blockIdx (same for every launch, grid size and stream)?
//   B. Does a line touched (one dword per 128 B / per 64 B / all of it) by a workgroup on XCD x in kernel P serve a later kernel's
//      non-temporal streaming loads from the XCD's L2 -- across a kernel boundary -- and how much faster is the launched GEMV
//      (pie_qgemv_w4g64, the product kernel) when the head of its matrix was touched on the matching / on a different XCD?
//   C. What does a resident, polling helper kernel on a second stream cost a chain of dependent GEMV launches?
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pilot_probe.hip -Iinclude -Lproxy_inference_engine_amd/lib -lpie_hip \
//        -Wl,-rpath,'$ORIGIN/../proxy_inference_engine_amd/lib' -o tools/pilot_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pie_hip.h"

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)
#define PK(x)                                                                                  \
    do {                                                                                       \
        int rc_ = (x);                                                                         \
        if (rc_ != 0) {                                                                        \
            printf("pie error %d (%s) at %s:%d\n", rc_, pie_last_error(), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ unsigned hw_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

__global__ void k_where(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id() | (hw_id() << 8);
}

// reads `bytes` with plain loads: evicts L2 and the Infinity Cache (600 MB > 256 MiB)
__global__ void k_flush(const uint4 *p, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345679u) *sink = acc;
}

// Touches the first `head_chunks` chunks (chunk c = bytes [c * chunk_bytes, +chunk_bytes): what workgroup c % 256 of the GEMV streams in
// round c / 256) from workgroups whose blockIdx % 8 == (c + shift) % 8.  mode 0: one dword per 128 B, 1: one dword per 64 B, 2: every byte
// (16 B per lane).  Grid: 8 * ranks workgroups of 256 threads.
__global__ void __launch_bounds__(256) k_touch(const char *base, unsigned chunk_bytes, unsigned head_chunks, int shift, int mode, unsigned *sink) {
    const unsigned g = blockIdx.x & 7u, j = blockIdx.x >> 3, ranks = gridDim.x >> 3;
    const unsigned step = mode == 0 ? 128u : (mode == 1 ? 64u : 16u);
    const unsigned per_chunk = chunk_bytes / step;
    unsigned acc = 0;
    for (unsigned c8 = j; c8 * 8u < head_chunks; c8 += ranks) {
        const unsigned c = c8 * 8u + ((g + 8u - (unsigned)shift) & 7u);
        if (c >= head_chunks) continue;
        const char *p = base + (size_t)c * chunk_bytes;
        for (unsigned l = threadIdx.x; l < per_chunk; l += 256u) {
            if (mode == 2) {
                const uint4 v = *reinterpret_cast<const uint4 *>(p + (size_t)l * 16u);
                acc ^= v.x ^ v.w;
            } else {
                acc ^= *reinterpret_cast<const unsigned *>(p + (size_t)l * step);
            }
        }
    }
    if (acc == 0x12345679u) *sink = acc;
}

// the resident helper of section C: lane 0 of every workgroup polls a device word until it becomes non-zero (or 50 ms pass)
__global__ void __launch_bounds__(256) k_resident(const unsigned *flag, unsigned *sink, int sleep) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned polls = 0;
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            ++polls;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) break;
            for (int i = 0; i < sleep; ++i) __builtin_amdgcn_s_sleep(8);
        }
        sink[blockIdx.x] = polls;
    }
}

static void *dmalloc(size_t b) {
    void *p;
    CK(hipMalloc(&p, b));
    return p;
}

int main(int argc, char **argv) {
    int reps = 40;
    for (int i = 1; i < argc; ++i)
        if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    unsigned *sink = (unsigned *)dmalloc(4096 * 4);
    CK(hipMemset(sink, 0, 4096 * 4));

    // ------------------------------------------------------------------ A. workgroup -> XCD map
    {
        unsigned *d = (unsigned *)dmalloc(4096 * 4);
        std::vector<unsigned> h(4096);
        auto show = [&](const char *what, int grid, hipStream_t st) {
            hipLaunchKernelGGL(k_where, dim3(grid), dim3(512), 0, st, d);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h.data(), d, (size_t)grid * 4, hipMemcpyDeviceToHost));
            int ok = 0;
            const unsigned x0 = h[0] & 15u;
            for (int b = 0; b < grid; ++b) ok += ((h[b] & 15u) == ((x0 + (unsigned)b) & 7u));
            printf("A %-34s grid %4d: XCC of block 0..15:", what, grid);
            for (int b = 0; b < 16 && b < grid; ++b) printf(" %u", h[b] & 15u);
            printf("  | blocks on (xcc0 + b) %% 8: %d / %d\n", ok, grid);
        };
        for (int r = 0; r < 3; ++r) show("stream 1", 256, s1);
        show("stream 1", 64, s1);
        show("stream 1", 33, s1);
        show("stream 1 (after a 33-block grid)", 256, s1);
        show("stream 1", 257, s1);
        show("stream 1 (after a 257-block grid)", 256, s1);
        show("stream 2", 256, s2);
        show("stream 2", 64, s2);
        show("stream 1", 2048, s1);
        // distinct CUs for a one-workgroup-per-CU grid?
        hipLaunchKernelGGL(k_where, dim3(256), dim3(512), 0, s1, d);
        CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost));
        std::vector<unsigned> ids;
        for (int b = 0; b < 256; ++b) {
            const unsigned hw = h[b] >> 8;  // HW_ID: [3:0] wave, [5:4] simd, [11:8] cu, [12] sh, [15:13] se (gfx9)
            ids.push_back(((h[b] & 15u) << 16) | (hw & 0xFF00u));
        }
        std::sort(ids.begin(), ids.end());
        const size_t uniq = std::unique(ids.begin(), ids.end()) - ids.begin();
        printf("A 256 workgroups of 512 threads landed on %zu distinct (xcc, se, sh, cu)\n", uniq);
        CK(hipFree(d));
    }

    // ------------------------------------------------------------------ B. touched lines and the launched GEMV
    const size_t flush_bytes = (size_t)640 << 20;
    uint4 *flush = (uint4 *)dmalloc(flush_bytes);
    CK(hipMemset(flush, 1, flush_bytes));
    struct Shape {
        const char *name;
        int N, K;
        size_t head_bytes;  // touched prefix
    } shapes[] = {{"o_proj 4096x4096", 4096, 4096, (size_t)1 << 40}, {"qkv 6144x4096", 6144, 4096, (size_t)1 << 40}, {"gate|up 28672x4096", 28672, 4096, (size_t)16 << 20},
                  {"gate|up 28672x4096 (8 MB head)", 28672, 4096, (size_t)8 << 20}, {"down 4096x14336 (whole)", 4096, 14336, (size_t)1 << 40}};
    for (const Shape &sh : shapes) {
        const size_t wb = pie_w4s_bytes(sh.N, sh.K);
        char *w = (char *)dmalloc(wb);
        CK(hipMemset(w, 0x11, wb));
        unsigned short *x = (unsigned short *)dmalloc((size_t)sh.K * 2), *y = (unsigned short *)dmalloc((size_t)sh.N * 2);
        CK(hipMemset(x, 0, (size_t)sh.K * 2));
        const int ns = (sh.K + 2047) / 2048;
        const unsigned chunk = 8u * ns * 2304u;  // one workgroup's 8 row pairs of a round
        const unsigned n_chunks = (unsigned)(wb / chunk);
        const unsigned head_chunks = (unsigned)std::min<size_t>(n_chunks, sh.head_bytes / chunk);
        struct Var {
            const char *name;
            int touch, shift, mode, ranks;
        } vars[] = {{"cold (no touch)", 0, 0, 0, 8},          {"touch 1 dw/128B, matching XCD", 1, 0, 0, 8}, {"touch 1 dw/128B, XCD + 1", 1, 1, 0, 8},
                    {"touch 1 dw/128B, XCD + 4", 1, 4, 0, 8}, {"touch 1 dw/64B, matching XCD", 1, 0, 1, 8},  {"touch all bytes, matching XCD", 1, 0, 2, 8},
                    {"touch 1 dw/128B, matching, 32 ranks", 1, 0, 0, 32}, {"touch 1 dw/128B, matching, 2 ranks", 1, 0, 0, 2}};
        printf("B %s: %zu bytes, chunk %u B, touching %u of %u chunks (%.1f MB)\n", sh.name, wb, chunk, head_chunks, n_chunks, head_chunks * (double)chunk / 1e6);
        for (const Var &v : vars) {
            hipEvent_t e0, e1, t0, t1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
            std::vector<float> tg, tt;
            for (int r = 0; r < reps + 3; ++r) {
                hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, s1, flush, flush_bytes / 16, sink);
                CK(hipEventRecord(t0, s1));
                if (v.touch) hipLaunchKernelGGL(k_touch, dim3(8 * v.ranks), dim3(256), 0, s1, w, chunk, head_chunks, v.shift, v.mode, sink);
                CK(hipEventRecord(t1, s1));
                CK(hipEventRecord(e0, s1));
                PK(pie_qgemv_w4g64(x, 1, w, sh.N, sh.K, nullptr, y, PIE_BF16, s1));
                CK(hipEventRecord(e1, s1));
                CK(hipStreamSynchronize(s1));
                float a, b;
                CK(hipEventElapsedTime(&a, e0, e1));
                CK(hipEventElapsedTime(&b, t0, t1));
                if (r >= 3) tg.push_back(a * 1e3f), tt.push_back(b * 1e3f);
            }
            std::sort(tg.begin(), tg.end()), std::sort(tt.begin(), tt.end());
            printf("B   %-40s gemv median %7.2f us (min %7.2f, p90 %7.2f) | touch kernel median %7.2f us\n", v.name, tg[tg.size() / 2], tg[0], tg[tg.size() * 9 / 10], tt[tt.size() / 2]);
        }
        CK(hipFree(w)); CK(hipFree(x)); CK(hipFree(y));
    }

    // ------------------------------------------------------------------ C. a resident polling helper beside a chain of launches
    {
        const int N = 28672, K = 4096, CHAIN = 40;
        const size_t wb = pie_w4s_bytes(N, K);
        char *w = (char *)dmalloc(wb * 4);  // four matrices in turn: no launch re-reads what the previous one left in the caches
        CK(hipMemset(w, 0x11, wb * 4));
        unsigned short *x = (unsigned short *)dmalloc((size_t)K * 2), *y = (unsigned short *)dmalloc((size_t)N * 2);
        CK(hipMemset(x, 0, (size_t)K * 2));
        unsigned *flag = (unsigned *)dmalloc(256);
        struct HV {
            const char *name;
            int wgs, sleep;
        } hv[] = {{"no helper", 0, 0}, {"helper 8 wgs, sleep 4x8", 8, 4}, {"helper 64 wgs, sleep 4x8", 64, 4}, {"helper 64 wgs, sleep 1x8", 64, 1}, {"helper 256 wgs, sleep 4x8", 256, 4}};
        for (const HV &h : hv) {
            std::vector<float> t;
            for (int r = 0; r < 8; ++r) {
                CK(hipMemsetAsync(flag, 0, 4, s1));
                CK(hipStreamSynchronize(s1));
                if (h.wgs) hipLaunchKernelGGL(k_resident, dim3(h.wgs), dim3(256), 0, s2, flag, sink, h.sleep);
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0, s1));
                for (int i = 0; i < CHAIN; ++i) PK(pie_qgemv_w4g64(x, 1, w + (size_t)(i & 3) * wb, N, K, nullptr, y, PIE_BF16, s1));
                CK(hipEventRecord(e1, s1));
                CK(hipStreamSynchronize(s1));
                CK(hipMemsetAsync(flag, 1, 4, s3));
                CK(hipStreamSynchronize(s3));
                CK(hipStreamSynchronize(s2));
                float a;
                CK(hipEventElapsedTime(&a, e0, e1));
                if (r >= 2) t.push_back(a * 1e3f / CHAIN);
            }
            std::sort(t.begin(), t.end());
            printf("C %-28s chain of %d gate|up GEMVs: %7.2f us per launch (min %7.2f)\n", h.name, CHAIN, t[t.size() / 2], t[0]);
        }
    }
    return 0;
}
