// tools/kslice_probe.hip -- what would o_proj behind a SECOND XCD-local seam cost?  (the follow-up to tools/seam_probe; round 6 material)
//
// After the fused q|k|v + attention launch, XCD c holds the attention output of query heads 4 c .. 4 c + 3 = 512 of o_proj's 4096 input columns.  If o_proj's
// weights were laid out per XCD-slice of K, each XCD could multiply ITS 512 columns into all 4096 output rows right there (no launch boundary: an XCD-local
// seam) and leave an fp32 partial vector; the 8 partials would be summed by the next launch's prologue (gate|up: today it stages one 8 KB vector).
// Two quantities decide whether that pays against o_proj's own launch (4.9 us kernel + 3.9 us boundary):
//   A. the in-kernel latency of the K-slice phase: 256 workgroups x 8 waves; XCD class c = blockIdx.x % 8; each workgroup 128 output rows x 512 columns of int4
//      g=64 weights (36 KB of synthetic units: 8 rows x 8 groups per 64-lane unit, 32 B codes + 4 B {scale | bias} per lane) against 512 activations
//      staged in LDS; lanes of one row are reduced by DPP; 128 fp32 partials stored.  s_memrealtime from workgroup start to the partials' store.
//   B. what the consumer's prologue pays to sum 8 partial vectors (8 x 16 KB fp32 + the 8 KB residual) instead of staging 8 KB: in-kernel, entry -> sum in LDS.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kslice_probe.hip -o tools/kslice_probe && tools/kslice_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

static inline void ck_(hipError_t e, const char *file, int line) {
    if (e != hipSuccess) {
        std::fprintf(stderr, "%s:%d: %s\n", file, line, hipGetErrorString(e));
        std::exit(2);
    }
}
#define CK(e) ck_((e), __FILE__, __LINE__)

typedef unsigned u32;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
constexpr int H = 4096, KS = 512, ROWS_PER_WG = 128, NT = 512;

__device__ __forceinline__ float dot2(u32 a, u32 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false); }

// A: the K-slice phase.  w: [8 xcd][32 wg][2 units per wave x 8 waves][64 lanes][36 B] -- unit u of wave v: rows 16 v + 8 u + (lane >> 3), group lane & 7
__global__ void __launch_bounds__(NT) k_slice(const uint4 *w_codes, const u32 *w_sb, const unsigned short *act, float *part, unsigned long long *stamps) {
    __shared__ __attribute__((aligned(16))) u32 s_x[KS / 2 * 1];  // 512 activations as 256 packed pairs
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int c = blockIdx.x & 7, j = blockIdx.x >> 3, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // weights first (the long latency), then the activations through LDS
    const size_t unit0 = (((size_t)c * 32 + j) * 8 + wave) * 2;
    uint4 c0[2], c1[2];
    u32 sb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const size_t base = (unit0 + u) * 64 + lane;
        c0[u] = w_codes[base * 2], c1[u] = w_codes[base * 2 + 1];
        sb[u] = w_sb[base];
    }
    if (threadIdx.x < KS / 2) s_x[threadIdx.x] = reinterpret_cast<const u32 *>(act + (size_t)c * KS)[threadIdx.x];
    __syncthreads();
    const int g = lane & 7;
    u32 xr[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) xr[i] = s_x[g * 32 + i];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const u32 wd[8] = {c0[u].x, c0[u].y, c0[u].z, c0[u].w, c1[u].x, c1[u].y, c1[u].z, c1[u].w};
        float d[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const u32 w8 = wd[t] >> 8;
            d[0] = dot2(wd[t] & 0x000F000Fu, xr[4 * t], d[0]);
            d[1] = dot2(wd[t] & 0x00F000F0u, xr[4 * t + 1], d[1]);
            d[2] = dot2(w8 & 0x000F000Fu, xr[4 * t + 2], d[2]);
            d[3] = dot2(w8 & 0x00F000F0u, xr[4 * t + 3], d[3]);
        }
        float acc = ((d[0] + d[2]) + (d[1] + d[3]) * 0.0625f) * __builtin_bit_cast(float, sb[u] << 16);
        acc += __shfl_xor(acc, 1, 64), acc += __shfl_xor(acc, 2, 64), acc += __shfl_xor(acc, 4, 64);  // the 8 groups of a row
        if (g == 0) part[(size_t)c * H + (size_t)j * ROWS_PER_WG + wave * 16 + u * 8 + (lane >> 3)] = acc;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (threadIdx.x == 0) stamps[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

// B: the consumer's prologue.  PARTS = 8: h[i] + sum of 8 fp32 partials -> LDS image; PARTS = 0: h only (today's staging)
template <int PARTS>
__global__ void __launch_bounds__(NT) k_prologue(const unsigned short *h, const float *part, float *sink, unsigned long long *stamps) {
    __shared__ float s_h[H];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int i = threadIdx.x * 8;  // 8 elements per thread
    const uint4 hv = *reinterpret_cast<const uint4 *>(h + i);
    float v[8] = {__builtin_bit_cast(float, hv.x << 16), __builtin_bit_cast(float, hv.x & 0xFFFF0000u), __builtin_bit_cast(float, hv.y << 16), __builtin_bit_cast(float, hv.y & 0xFFFF0000u),
                  __builtin_bit_cast(float, hv.z << 16), __builtin_bit_cast(float, hv.z & 0xFFFF0000u), __builtin_bit_cast(float, hv.w << 16), __builtin_bit_cast(float, hv.w & 0xFFFF0000u)};
    float4 p0[PARTS ? PARTS : 1], p1[PARTS ? PARTS : 1];
#pragma unroll
    for (int c = 0; c < PARTS; ++c) {
        p0[c] = *reinterpret_cast<const float4 *>(part + (size_t)c * H + i);
        p1[c] = *reinterpret_cast<const float4 *>(part + (size_t)c * H + i + 4);
    }
#pragma unroll
    for (int c = 0; c < PARTS; ++c) {
        v[0] += p0[c].x, v[1] += p0[c].y, v[2] += p0[c].z, v[3] += p0[c].w;
        v[4] += p1[c].x, v[5] += p1[c].y, v[6] += p1[c].z, v[7] += p1[c].w;
    }
    float ss = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s_h[i + e] = v[e], ss += v[e] * v[e];
    __syncthreads();
    if (threadIdx.x == 0) stamps[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
    if (ss == 123.456f) sink[blockIdx.x] = s_h[(threadIdx.x * 7) & (H - 1)];
}

static void report(const char *name, std::vector<unsigned long long> &v) {
    std::sort(v.begin(), v.end());
    std::printf("%-70s per workgroup: median %.2f us, p10 %.2f, p90 %.2f, max %.2f\n", name, v[v.size() / 2] * 0.01, v[v.size() / 10] * 0.01, v[v.size() * 9 / 10] * 0.01,
                v.back() * 0.01);
}

int main() {
    uint4 *w_codes;
    u32 *w_sb;
    unsigned short *act, *h;
    float *part, *sink;
    unsigned long long *stamps;
    const size_t units = 8 * 32 * 8 * 2, lanes = units * 64;
    CK(hipMalloc((void **)&w_codes, lanes * 32)), CK(hipMalloc((void **)&w_sb, lanes * 4)), CK(hipMalloc((void **)&act, H * 2)), CK(hipMalloc((void **)&h, H * 2));
    CK(hipMalloc((void **)&part, 8 * H * 4)), CK(hipMalloc((void **)&sink, 256 * 4)), CK(hipMalloc((void **)&stamps, 256 * 8));
    CK(hipMemset(w_codes, 0x35, lanes * 32)), CK(hipMemset(w_sb, 0x3c, lanes * 4)), CK(hipMemset(act, 0x3c, H * 2)), CK(hipMemset(h, 0x3c, H * 2)), CK(hipMemset(part, 0, 8 * H * 4));
    char *flush;
    CK(hipMalloc((void **)&flush, 512u << 20));  // evict the weights between repetitions: the K-slice phase reads them cold, like the step does
    std::vector<unsigned long long> a, b8, b0, hs(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)), CK(hipEventCreate(&e1));
    float kern_a = 0, kern_b8 = 0, kern_b0 = 0;
    for (int rep = 0; rep < 12; ++rep) {
        float ms;
        CK(hipMemset(flush, rep, 512u << 20));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_slice, dim3(256), dim3(NT), 0, 0, w_codes, w_sb, act, part, stamps);
        CK(hipEventRecord(e1, 0)), CK(hipEventSynchronize(e1)), CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hs.data(), stamps, 256 * 8, hipMemcpyDeviceToHost));
        if (rep >= 2) a.insert(a.end(), hs.begin(), hs.end()), kern_a += ms;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_prologue<8>, dim3(256), dim3(NT), 0, 0, h, part, sink, stamps);
        CK(hipEventRecord(e1, 0)), CK(hipEventSynchronize(e1)), CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hs.data(), stamps, 256 * 8, hipMemcpyDeviceToHost));
        if (rep >= 2) b8.insert(b8.end(), hs.begin(), hs.end()), kern_b8 += ms;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_prologue<0>, dim3(256), dim3(NT), 0, 0, h, part, sink, stamps);
        CK(hipEventRecord(e1, 0)), CK(hipEventSynchronize(e1)), CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hs.data(), stamps, 256 * 8, hipMemcpyDeviceToHost));
        if (rep >= 2) b0.insert(b0.end(), hs.begin(), hs.end()), kern_b0 += ms;
    }
    report("A. K-slice phase (128 rows x 512 columns per workgroup, cold weights), start -> partials stored", a);
    report("B. prologue that sums 8 fp32 partial vectors + the residual, entry -> image in LDS", b8);
    report("   prologue that stages the residual only (today), entry -> image in LDS", b0);
    std::printf("(whole launches by HIP events, us: A %.1f, B8 %.1f, B0 %.1f -- standalone, each with its own dispatch)\n", kern_a * 100, kern_b8 * 100, kern_b0 * 100);
    return 0;
}
