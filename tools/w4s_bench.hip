// tools/w4s_bench.hip -- kernel-level microbenchmark for the W4S GEMV (developer tool, not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/w4s_bench.hip proxy_inference_engine_amd/csrc/w4_gemv.hip \
//        proxy_inference_engine_amd/csrc/{decoder,ops,prefill,vision}.hip proxy_inference_engine_amd/csrc/page_pool.cpp -ldl -o tools/w4s_bench
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../proxy_inference_engine_amd/csrc/w4_gemv.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// A. pure streaming of W4S units: what the access pattern alone can do.
template <int U, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_stream(const char *w, size_t n_units, u32 *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t unit0 = ((size_t)blockIdx.x * WAVES + wave) * U;
    uint4 c0[U], c1[U];
    u32 sb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t un = unit0 + u;
        un = un < n_units ? un : n_units - 1;
        const char *p = w + un * W4S_UNIT_BYTES;
        c0[u] = *reinterpret_cast<const uint4 *>(p + lane * 16);
        c1[u] = *reinterpret_cast<const uint4 *>(p + 1024 + lane * 16);
        sb[u] = *reinterpret_cast<const u32 *>(p + 2048 + lane * 4);
    }
    u32 acc = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= c0[u].x ^ c0[u].y ^ c0[u].z ^ c0[u].w ^ c1[u].x ^ c1[u].y ^ c1[u].z ^ c1[u].w ^ sb[u];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// B. streaming + the dequant/dot math on register-resident x (no prologue, no reduction)
template <int U, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_stream_dot(const char *w, size_t n_units, const u32 *x, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t unit0 = ((size_t)blockIdx.x * WAVES + wave) * U;
    uint4 c0[U], c1[U];
    u32 sb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t un = unit0 + u;
        un = un < n_units ? un : n_units - 1;
        const char *p = w + un * W4S_UNIT_BYTES;
        c0[u] = *reinterpret_cast<const uint4 *>(p + lane * 16);
        c1[u] = *reinterpret_cast<const uint4 *>(p + 1024 + lane * 16);
        sb[u] = *reinterpret_cast<const u32 *>(p + 2048 + lane * 4);
    }
    u32 xr[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) xr[i] = x[i * 64 + lane];
    float tot = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float d = w4s_unit_dot<BF16>(c0[u], c1[u], xr);
        tot += lo_f32<BF16>(sb[u]) * d + hi_f32<BF16>(sb[u]);
    }
    if (tot == 1.2345f) out[blockIdx.x] = tot;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    void start() { CK(hipEventRecord(a, 0)); }
    float stop() { CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

int main(int argc, char **argv) {
    const size_t W_BYTES = (size_t)1536 << 20;  // 1.5 GiB of weights: cycling through it defeats the 256 MiB Infinity Cache
    char *w; u32 *x; float *outf; u16 *y; u16 *xin, *normw, *resid;
    CK(hipMalloc(&w, W_BYTES));
    CK(hipMalloc(&x, 64 * 64 * 4)); CK(hipMalloc(&outf, 1 << 20)); CK(hipMalloc(&y, 1 << 20));
    CK(hipMalloc(&xin, 65536)); CK(hipMalloc(&normw, 65536)); CK(hipMalloc(&resid, 65536));
    {   // random fill (host side, small pattern repeated)
        std::vector<u32> h(1 << 20);
        for (auto &v : h) v = (u32)rand() * 2654435761u;
        for (size_t off = 0; off < W_BYTES; off += h.size() * 4) CK(hipMemcpy(w + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<u16> hx(32768, 0x3f80);
        CK(hipMemcpy(x, hx.data(), 64 * 64 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(xin, hx.data(), 65536, hipMemcpyHostToDevice));
        CK(hipMemcpy(normw, hx.data(), 65536, hipMemcpyHostToDevice));
        CK(hipMemset(resid, 0, 65536));
    }
    Timer t;
    auto report = [&](const char *name, size_t bytes_per_launch, int launches, float ms) {
        printf("%-44s %8.2f us/launch  %8.1f GB/s\n", name, 1e3 * ms / launches, bytes_per_launch * (double)launches / (ms * 1e-3) / 1e9);
    };
    // gate/up sized problem: 14336 pairs x 2 slices = 28672 units = 66 MB
    const size_t units = 28672, bytes = units * W4S_UNIT_BYTES;
    const int slots = (int)(W_BYTES / bytes), reps = 60;
#define RUN_STREAM(U, WAVES)                                                                                         \
    {                                                                                                                \
        int grid = (int)((units + (size_t)U * WAVES - 1) / ((size_t)U * WAVES));                                     \
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<U, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, w + (size_t)(i % slots) * bytes, units, (u32 *)outf); \
        CK(hipDeviceSynchronize()); t.start();                                                                       \
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<U, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, w + (size_t)(i % slots) * bytes, units, (u32 *)outf); \
        char nm[64]; snprintf(nm, 64, "stream        U=%d waves/WG=%d grid=%d", U, WAVES, grid); report(nm, bytes, reps, t.stop());         \
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream_dot<U, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, w + (size_t)(i % slots) * bytes, units, x, outf); \
        CK(hipDeviceSynchronize()); t.start();                                                                       \
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream_dot<U, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, w + (size_t)(i % slots) * bytes, units, x, outf); \
        snprintf(nm, 64, "stream+dot    U=%d waves/WG=%d grid=%d", U, WAVES, grid); report(nm, bytes, reps, t.stop());              \
    }
    if (argc > 1 && std::string(argv[1]) == "pmc") {
        // counter-collection mode (rocprofv3 --pmc ...): only the calibration stream and the cold gate/up product kernel,
        // so per-kernel averages are not mixed across shapes or cache states
        RUN_STREAM(1, 4)
        const int N = 28672, K = 4096;
        size_t b = (size_t)(N / 2) * w4s_slices(K) * W4S_UNIT_BYTES;
        int sl = (int)(W_BYTES / b);
        for (int i = 0; i < 64; ++i) {
            GemvArgs a = {};
            a.w = w + (size_t)(i % sl) * b, a.K = K, a.N = N, a.x = xin, a.norm_w = normw, a.eps = 1e-5f, a.y = y, a.resid = resid;
            if (w4s_gemv_launch(PIE_BF16, PRO_RMSNORM, EPI_SWIGLU, a, 1, 0)) { printf("launch failed: %s\n", pie_last_error()); exit(1); }
        }
        // the other per-layer shapes (distinct template instantiations, so their per-kernel averages stay separate):
        // down (NONE, RESIDUAL, K = 14336), o_proj-sized (NONE, RESIDUAL, K = 4096 -> NPT 1), q|k|v-sized (RMSNORM, STORE)
        struct PmcCase { int pro, epi, N, K; } pc[] = {{PRO_NONE, EPI_RESIDUAL, 4096, 14336}, {PRO_NONE, EPI_RESIDUAL, 4096, 4096}, {PRO_NONE, EPI_STORE, 6144, 4096}};
        for (auto &c : pc) {
            size_t cb = (size_t)(c.N / 2) * w4s_slices(c.K) * W4S_UNIT_BYTES;
            int csl = (int)(W_BYTES / cb);
            for (int i = 0; i < 64; ++i) {
                GemvArgs a = {};
                a.w = w + (size_t)(i % csl) * cb, a.K = c.K, a.N = c.N, a.x = xin, a.norm_w = normw, a.eps = 1e-5f, a.y = y, a.resid = resid;
                if (w4s_gemv_launch(PIE_BF16, c.pro, c.epi, a, 1, 0)) { printf("launch failed: %s\n", pie_last_error()); exit(1); }
            }
        }
        CK(hipDeviceSynchronize());
        return 0;
    }
    RUN_STREAM(1, 4) RUN_STREAM(2, 4) RUN_STREAM(4, 4) RUN_STREAM(8, 4) RUN_STREAM(4, 8) RUN_STREAM(2, 8) RUN_STREAM(8, 8) RUN_STREAM(4, 16) RUN_STREAM(16, 4)

    // product kernel variants through the launcher
    struct Case { const char *name; int pro, epi, N, K; };
    Case cases[] = {{"gemv store      N=28672 K=4096", PRO_NONE, EPI_STORE, 28672, 4096}, {"gemv rms+swiglu N=28672 K=4096", PRO_RMSNORM, EPI_SWIGLU, 28672, 4096},
                    {"gemv residual   N=4096  K=14336", PRO_NONE, EPI_RESIDUAL, 4096, 14336}, {"gemv residual   N=4096  K=4096", PRO_NONE, EPI_RESIDUAL, 4096, 4096},
                    {"gemv store      N=6144  K=4096", PRO_NONE, EPI_STORE, 6144, 4096}, {"gemv store      N=128256 K=4096", PRO_NONE, EPI_STORE, 128256, 4096}};
    for (auto &c : cases) {
        size_t b = (size_t)(c.N / 2) * w4s_slices(c.K) * W4S_UNIT_BYTES;
        int sl = (int)(W_BYTES / b);
        auto launch = [&](int i) {
            GemvArgs a = {};
            a.w = w + (size_t)(i % sl) * b, a.K = c.K, a.N = c.N, a.x = xin, a.norm_w = normw, a.eps = 1e-5f, a.y = y, a.resid = resid;
            if (w4s_gemv_launch(PIE_BF16, c.pro, c.epi, a, 1, 0)) { printf("launch failed: %s\n", pie_last_error()); exit(1); }
        };
        for (int i = 0; i < 3; ++i) launch(i);
        CK(hipDeviceSynchronize()); t.start();
        for (int i = 0; i < reps; ++i) launch(i);
        report(c.name, b, reps, t.stop());
    }
    // ablations of the product kernel on the gate/up shape (N=28672, K=4096)
    {
        const int N = 28672, K = 4096;
        size_t b = (size_t)(N / 2) * w4s_slices(K) * W4S_UNIT_BYTES;
        int sl = (int)(W_BYTES / b);
        auto mk = [&](int i) {
            GemvArgs a = {};
            a.w = w + (size_t)(i % sl) * b, a.K = K, a.N = N, a.x = xin, a.norm_w = normw, a.eps = 1e-5f, a.y = y, a.resid = resid;
            a.n_slices = w4s_slices(K), a.n_pairs = N / 2, a.n_waves = w4s_gemv_waves(N, K);
            a.full_rounds = a.n_pairs / a.n_waves, a.rem_pairs = a.n_pairs % a.n_waves, a.n_blocks = a.n_waves % GEMV_WAVES == 0 ? (a.n_waves / GEMV_WAVES) : 0;
            return a;
        };
        const unsigned lds = (unsigned)gemv_lds(K).total;
#define RUN_ABL(PRO, EPI, ABL, label)                                                                                     \
        {                                                                                                                  \
            dim3 grid((w4s_gemv_waves(N, K) + GEMV_WAVES - 1) / GEMV_WAVES), block(64 * GEMV_WAVES);                        \
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_w4s_gemv<BF16, PRO, EPI, 1, ABL>), grid, block, lds, 0, mk(i)); \
            CK(hipDeviceSynchronize()); t.start();                                                                         \
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_w4s_gemv<BF16, PRO, EPI, 1, ABL>), grid, block, lds, 0, mk(i)); \
            report(label, b, reps, t.stop());                                                                              \
        }
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 0, "abl rms+swiglu full")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 1, "abl rms+swiglu no-weight-loads")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 2, "abl rms+swiglu no-dot")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 4, "abl rms+swiglu no-x-staging")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 6, "abl rms+swiglu loads only")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 7, "abl rms+swiglu nothing")
        RUN_ABL(PRO_NONE, EPI_STORE, 0, "abl store full")
        RUN_ABL(PRO_NONE, EPI_STORE, 4, "abl store no-x-staging")
    }
    // ablations of the W2S stream (2-bit units, round 5) on the lm_head shape (N=128256, K=4096) and on gate/up: what bounds it?
    for (int shape = 0; shape < 2; ++shape) {
        const int N = shape ? 28672 : 128256, K = 4096;
        size_t b = (size_t)(N / 2) * w4s_slices(K) * W2S_UNIT_BYTES;
        int sl = (int)(W_BYTES / b);
        auto mk = [&](int i) {
            GemvArgs a = {};
            a.fmt = FMT_W2S;
            a.w = w + (size_t)(i % sl) * b, a.K = K, a.N = N, a.x = xin, a.norm_w = normw, a.eps = 1e-5f, a.y = y, a.resid = resid;
            a.n_slices = w4s_slices(K), a.n_pairs = N / 2, a.n_waves = w4s_gemv_waves(N, K);
            a.full_rounds = a.n_pairs / a.n_waves, a.rem_pairs = a.n_pairs % a.n_waves, a.n_blocks = a.n_waves % GEMV_WAVES == 0 ? (a.n_waves / GEMV_WAVES) : 0;
            return a;
        };
        const unsigned lds = (unsigned)gemv_lds(K).total;
#define RUN_ABL2(ABL, label)                                                                                              \
        {                                                                                                                  \
            dim3 grid((w4s_gemv_waves(N, K) + GEMV_WAVES - 1) / GEMV_WAVES), block(64 * GEMV_WAVES);                        \
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_w4s_gemv<BF16, PRO_NONE, EPI_STORE, 1, ABL, FMT_W2S>), grid, block, lds, 0, mk(i)); \
            CK(hipDeviceSynchronize()); t.start();                                                                         \
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_w4s_gemv<BF16, PRO_NONE, EPI_STORE, 1, ABL, FMT_W2S>), grid, block, lds, 0, mk(i)); \
            char nm[64]; snprintf(nm, 64, "w2s N=%d %s", N, label); report(nm, b, reps, t.stop());                                      \
        }
        RUN_ABL2(0, "full")
        {   // the same launch with TWO 8-wave workgroups per CU (4096 waves): does a second pair of waves per SIMD hide the per-unit serial chain?
            auto mk2 = [&](int i) {
                GemvArgs a = mk(i);
                a.n_waves = 4096;
                a.full_rounds = a.n_pairs / a.n_waves, a.rem_pairs = a.n_pairs % a.n_waves, a.n_blocks = a.n_waves / GEMV_WAVES;
                return a;
            };
            dim3 grid(4096 / GEMV_WAVES), block(64 * GEMV_WAVES);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_w4s_gemv<BF16, PRO_NONE, EPI_STORE, 1, 0, FMT_W2S>), grid, block, lds, 0, mk2(i));
            CK(hipDeviceSynchronize()); t.start();
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_w4s_gemv<BF16, PRO_NONE, EPI_STORE, 1, 0, FMT_W2S>), grid, block, lds, 0, mk2(i));
            char nm2[64]; snprintf(nm2, 64, "w2s N=%d full, 4096 waves", N); report(nm2, b, reps, t.stop());
        }
        RUN_ABL2(1, "no-weight-loads")
        RUN_ABL2(2, "no-dot")
        RUN_ABL2(4, "no-x-reads")
        RUN_ABL2(6, "loads only")
        RUN_ABL2(5, "dot only")
        RUN_ABL2(7, "nothing")
#undef RUN_ABL2
    }
    // ablations of the product kernel on the qkv shape (N=6144, K=4096)
    {
        const int N = 6144, K = 4096;
        size_t b = (size_t)(N / 2) * w4s_slices(K) * W4S_UNIT_BYTES;
        int sl = (int)(W_BYTES / b);
        auto mk = [&](int i) {
            GemvArgs a = {};
            a.w = w + (size_t)(i % sl) * b, a.K = K, a.N = N, a.x = xin, a.norm_w = normw, a.eps = 1e-5f, a.y = y, a.resid = resid;
            a.n_slices = w4s_slices(K), a.n_pairs = N / 2, a.n_waves = w4s_gemv_waves(N, K);
            a.full_rounds = a.n_pairs / a.n_waves, a.rem_pairs = a.n_pairs % a.n_waves, a.n_blocks = a.n_waves % GEMV_WAVES == 0 ? (a.n_waves / GEMV_WAVES) : 0;
            return a;
        };
        const unsigned lds = (unsigned)gemv_lds(K).total;
#undef RUN_ABL
#define RUN_ABL(PRO, EPI, ABL, label)                                                                                     \
        {                                                                                                                  \
            dim3 grid((w4s_gemv_waves(N, K) + GEMV_WAVES - 1) / GEMV_WAVES), block(64 * GEMV_WAVES);                        \
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_w4s_gemv<BF16, PRO, EPI, 1, ABL>), grid, block, lds, 0, mk(i)); \
            CK(hipDeviceSynchronize()); t.start();                                                                         \
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_w4s_gemv<BF16, PRO, EPI, 1, ABL>), grid, block, lds, 0, mk(i)); \
            report(label, b, reps, t.stop());                                                                              \
        }
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 0, "abl6144 rms+swiglu full")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 1, "abl6144 rms+swiglu no-weight-loads")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 2, "abl6144 rms+swiglu no-dot")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 4, "abl6144 rms+swiglu no-x-staging")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 6, "abl6144 rms+swiglu loads only")
        RUN_ABL(PRO_RMSNORM, EPI_SWIGLU, 7, "abl6144 rms+swiglu nothing")
        RUN_ABL(PRO_NONE, EPI_STORE, 0, "abl6144 store full")
        RUN_ABL(PRO_NONE, EPI_STORE, 4, "abl6144 store no-x-staging")
    }
    return 0;
}
