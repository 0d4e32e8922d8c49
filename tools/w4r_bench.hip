// tools/w4r_bench.hip -- the weight-streaming int4 GEMM (k_w4r_gemm, w4r_gemm.hpp) alone: parity against a plain reference kernel and
// against round 2's kernels (k_w4m_gemm up to 32 rows -- its multi-strip siblings, which this kernel replaced, are in git history -- and k_w4l2_gemm), and timing on the 8B model's layer shapes with cold weights (developer tool, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DW4R_ABL=<mask>] tools/w4r_bench.hip -o tools/w4r_bench[_<mask>]
//   tools/w4r_bench check            parity over a grid of (M, N, K) incl. ragged M, K splits and every epilogue
//   tools/w4r_bench time [M ...]     us per launch per shape, new vs old kernels
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../proxy_inference_engine_amd/csrc/w4m_gemm.hip"

int bias_any_launch(int, void *, const void *, int, int, hipStream_t) { return 0; }  // vision.hip's; not reached from here
static int g_knob_w4r = -1;
static bool g_plain = false;  // W4R_PLAIN=1: the plain conversion at every row count
#ifdef W4R_PROF
unsigned long long *g_w4r_prof = nullptr;
#endif
int pie_knob(int k) { return k == PIE_KNOB_W4R ? g_knob_w4r : -1; }
namespace pie {
int fail(int code, const std::string &msg) {
    std::fprintf(stderr, "error %d: %s\n", code, msg.c_str());
    return code;
}
}  // namespace pie

static u16 f2bf(float f) {
    u32 u;
    memcpy(&u, &f, 4);
    return (u16)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}
static float bf2f(u16 b) {
    u32 u = (u32)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// reference: y32[m][n] = sum_k x[m][k] * T(s q + b), straight from the W4M tiles, double accumulation; one thread per (m, n)
__global__ void k_ref(const char *w4m, const u16 *x, int M, int N, int K, float *y32) {
    const int nn = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (nn >= N) return;
    const int groups = K >> 6, nt = nn >> 5, n = nn & 31;
    double acc = 0.0;
    for (int g = 0; g < groups; ++g) {
        const char *tile = w4m + ((size_t)nt * groups + g) * W4M_TILE_BYTES;
        const u32 sbw = reinterpret_cast<const u32 *>(tile + 1024)[n];
        const float s = BF16::to_f32((u16)(sbw & 0xFFFF)), b = BF16::to_f32((u16)(sbw >> 16));
        for (int kh = 0; kh < 2; ++kh)
            for (int wd = 0; wd < 4; ++wd) {
                const u32 word = reinterpret_cast<const u32 *>(tile)[(32 * kh + n) * 4 + wd];
                for (int j = 0; j < 8; ++j) {
                    const u32 q = (word >> (4 * (j >> 1) + 16 * (j & 1))) & 15u;
                    const float wv = BF16::to_f32(BF16::from_f32(__fadd_rn(__fmul_rn(s, (float)q), b)));
                    acc += (double)wv * (double)BF16::to_f32(x[(size_t)m * K + 64 * g + 16 * wd + 8 * kh + j]);
                }
            }
    }
    y32[(size_t)m * N + nn] = (float)acc;
}

static std::vector<char> make_tiles(int N, int K, unsigned seed) {
    std::vector<char> w(w4m_bytes(N, K));
    srand(seed);
    const size_t tiles = w.size() / W4M_TILE_BYTES;
    for (size_t t = 0; t < tiles; ++t) {
        u32 *p = reinterpret_cast<u32 *>(w.data() + t * W4M_TILE_BYTES);
        for (int i = 0; i < 256; ++i) p[i] = ((u32)rand() << 16) ^ (u32)rand() ^ ((u32)rand() << 8);
        for (int i = 0; i < 32; ++i) {
            const float s = (0.002f + 0.01f * (rand() % 1000) / 1000.0f) * ((rand() & 1) ? 1.0f : -1.0f);
            const float b = -7.5f * s + 0.003f * ((rand() % 200) - 100) / 100.0f;
            p[256 + i] = (u32)f2bf(s) | ((u32)f2bf(b) << 16);
        }
    }
    return w;
}

struct Dev {
    char *w = nullptr;
    u16 *x = nullptr, *y = nullptr, *y_old = nullptr, *bias = nullptr;
    float *ref = nullptr, *ws = nullptr;
};

static int check_shape(int M, int N, int K, int epi, bool with_bias, bool no_slab_consumer) {
    const auto hw = make_tiles(N, K, 17 + M + N + K);
    std::vector<u16> hx((size_t)M * K), hb(N);
    for (auto &v : hx) v = f2bf((float)((rand() % 2001) - 1000) / 1000.0f);
    for (auto &v : hb) v = f2bf((float)((rand() % 2001) - 1000) / 4000.0f);
    Dev d;
    const size_t out_elems = (size_t)M * N;
    hipMalloc(&d.w, hw.size()), hipMalloc(&d.x, hx.size() * 2), hipMalloc(&d.y, out_elems * 2 + 64), hipMalloc(&d.ref, out_elems * 4), hipMalloc(&d.bias, N * 2);
    hipMemcpy(d.w, hw.data(), hw.size(), hipMemcpyHostToDevice), hipMemcpy(d.x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d.bias, hb.data(), N * 2, hipMemcpyHostToDevice);
    hipMemset(d.y, 0xFF, out_elems * 2 + 64);
    const size_t wsb = w4r_workspace_bytes(M, N, K);
    if (wsb) hipMalloc(&d.ws, wsb), hipMemset(d.ws, 0xFF, wsb);
    hipLaunchKernelGGL(k_ref, dim3((N + 127) / 128, M), dim3(128), 0, 0, d.w, d.x, M, N, K, d.ref);
    int slabs = 0;
    bool bias_done = false;
    const int rc = w4r_gemm_launch(PIE_BF16, d.w, d.x, M, N, K, d.y, d.ws, 0, epi, with_bias ? d.bias : nullptr, nullptr, no_slab_consumer ? nullptr : &slabs, &bias_done, g_plain);
    if (rc) return 1;
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("M=%d N=%d K=%d epi=%d: kernel FAULT: %s\n", M, N, K, epi, hipGetErrorString(hipGetLastError()));
        exit(2);
    }
    std::vector<float> ref(out_elems), got(out_elems);
    hipMemcpy(ref.data(), d.ref, out_elems * 4, hipMemcpyDeviceToHost);
    std::vector<u16> hy(out_elems);
    hipMemcpy(hy.data(), d.y, out_elems * 2, hipMemcpyDeviceToHost);
    const W4rPlan pl = w4r_plan(M, N, K, epi == W4R_STORE && d.ws);
    double max_err = 0.0, max_ref = 0.0;
    size_t bad = 0, off1 = 0, cnt = out_elems;
    if (slabs > 1) {  // sum the slabs in order, compare un-rounded
        std::vector<float> part((size_t)slabs * out_elems);
        hipMemcpy(part.data(), d.ws, part.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < out_elems; ++i) {
            float v = part[i];
            for (int z = 1; z < slabs; ++z) v += part[(size_t)z * out_elems + i];
            const double e = fabs((double)v - ref[i]);
            max_err = e > max_err ? e : max_err, max_ref = fabs(ref[i]) > max_ref ? fabs(ref[i]) : max_ref;
            if (e > 2e-4 * (1.0 + fabs(ref[i]))) ++bad;
        }
    } else if (epi == W4R_SWIGLU) {
        cnt = out_elems / 2;
        for (int m = 0; m < M; ++m)
            for (int i = 0; i < N / 2; ++i) {
                float g = bf2f(f2bf(ref[(size_t)m * N + 2 * i])), u = bf2f(f2bf(ref[(size_t)m * N + 2 * i + 1]));
                if (with_bias) g = bf2f(f2bf(g + bf2f(hb[2 * i]))), u = bf2f(f2bf(u + bf2f(hb[2 * i + 1])));
                const float want = bf2f(f2bf(bf2f(f2bf(g / (1.0f + expf(-g)))) * u));
                const float have = bf2f(hy[(size_t)m * (N / 2) + i]);
                const double e = fabs((double)have - want);
                max_err = e > max_err ? e : max_err, max_ref = fabs(want) > max_ref ? fabs(want) : max_ref;
                if (e > 0.02 * (0.02 + fabs(want))) ++bad;
                else if (have != want) ++off1;
            }
    } else {
        for (size_t i = 0; i < out_elems; ++i) {
            float want = bf2f(f2bf(ref[i]));
            const float mag = fabs(want);  // the rounding slack is one bf16 ulp of the Linear's output, whatever the bias then cancels
            if (with_bias && bias_done) want = bf2f(f2bf(want + bf2f(hb[i % N])));
            const float have = bf2f(hy[i]);
            const double e = fabs((double)have - want);
            max_err = e > max_err ? e : max_err, max_ref = fabs(want) > max_ref ? fabs(want) : max_ref;
            if (e > 0.0079 * (0.01 + (mag > fabs(want) ? mag : fabs(want)))) ++bad;  // one bf16 ulp of slack: fp32 summation order differs from the double reference
            else if (have != want) ++off1;
        }
    }
    // guard words behind y untouched?
    std::vector<u16> tail(32);
    hipMemcpy(tail.data(), d.y + (epi == W4R_SWIGLU ? out_elems / 2 : out_elems), 64, hipMemcpyDeviceToHost);
    bool overrun = false;
    if (slabs <= 1)
        for (auto v : tail) overrun |= v != 0xFFFF;
    printf("M=%3d N=%6d K=%6d epi=%d bias=%d  plan mb=%d kw=%d S=%d steps=%d slabs=%d: max|err| %.3g (max|ref| %.3g), off-by-rounding %zu / %zu, BAD %zu%s\n", M, N, K, epi,
           (int)with_bias, pl.mb, pl.kw, pl.S, pl.steps, slabs, max_err, max_ref, off1, cnt, bad, overrun ? "  OVERRUN" : "");
    hipFree(d.w), hipFree(d.x), hipFree(d.y), hipFree(d.ref), hipFree(d.bias);
    if (d.ws) hipFree(d.ws);
    return (bad || overrun) ? 1 : 0;
}

static double time_it(int iters, const std::function<void(int)> &f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < iters / 2; ++i) f(i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) f(i);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0), hipEventDestroy(e1);
    return ms * 1e3 / iters;
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "check";
    if (getenv("W4R_PLAIN")) g_plain = atoi(getenv("W4R_PLAIN")) != 0;
    if (getenv("W4R_KNOB")) g_knob_w4r = atoi(getenv("W4R_KNOB"));  // developer: plan variants behind PIE_KNOB_W4R values > 1
    if (mode == "check") {
        int fails = 0;
        const int Ms[] = {6, 17, 32, 33, 64, 65, 96, 100, 128, 129, 160, 161, 192, 200, 256};
        for (int M : Ms) {
            fails += check_shape(M, 256, 512, W4R_STORE, false, false);          // tiny: 2 column workgroups, K split
            fails += check_shape(M, 6144, 4096, W4R_STORE, M % 2 == 0, false);   // q|k|v shape, slabs
            fails += check_shape(M, 1024, 1024, W4R_SWIGLU, M % 3 == 0, false);
        }
        fails += check_shape(32, 4096, 14336, W4R_STORE, false, false);   // down: 7-step splits
        fails += check_shape(128, 4096, 14336, W4R_STORE, false, true);   // reduce launch instead of a slab consumer
        fails += check_shape(256, 4096, 4096, W4R_STORE, true, true);
        fails += check_shape(128, 28672, 4096, W4R_SWIGLU, false, false);  // gate|up
        fails += check_shape(48, 4128, 2048, W4R_STORE, true, false);      // 129 strips: a ragged last column workgroup
        fails += check_shape(9, 128256, 4096, W4R_STORE, false, false);    // lm_head
        for (int M : {7, 40, 100, 150, 200}) {
            fails += check_shape(M, 256, 704, W4R_STORE, false, false);   // 11 groups: a partial last step
            fails += check_shape(M, 1408, 1408, W4R_SWIGLU, false, false);  // 22 groups, 44 strips
            fails += check_shape(M, 512, 320, W4R_STORE, true, true);     // 5 groups
        }
        printf(fails ? "CHECK FAILED (%d shapes)\n" : "check ok\n", fails);
        return fails ? 1 : 0;
    }
#ifdef W4R_PROF
    if (mode == "prof") {  // timeline of a few workgroups of one gate|up launch
        const int M = argc > 2 ? atoi(argv[2]) : 32, N = 28672, K = 4096;
        const size_t bytes = w4m_bytes(N, K);
        char *w;
        u16 *x, *y;
        unsigned long long *prof;
        hipMalloc(&w, bytes * 5), hipMalloc(&x, 256 * K * 2), hipMalloc(&y, (size_t)256 * N * 2), hipMalloc(&prof, 8 * 8 * 32 * 4 * 8);
        const auto hw = make_tiles(N, K, 5);
        for (int i = 0; i < 5; ++i) hipMemcpy(w + bytes * i, hw.data(), bytes, hipMemcpyHostToDevice);
        hipMemset(x, 0x3c, 256 * K * 2), hipMemset(prof, 0, 8 * 8 * 32 * 4 * 8);
        extern unsigned long long *g_w4r_prof;
        for (int i = 0; i < 5; ++i) {
            g_w4r_prof = i == 4 ? prof : nullptr;
            w4r_gemm_launch(PIE_BF16, w + bytes * i, x, M, N, K, y, nullptr, 0, W4R_SWIGLU, nullptr, nullptr, nullptr, nullptr, g_plain);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(8 * 8 * 32 * 4);
        hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
        const int waves = 8;
        for (int wg = 0; wg < 4; ++wg) {
            const unsigned long long *b = h.data() + (size_t)wg * waves * 32 * 4;
            const unsigned long long t0 = b[0];
            printf("workgroup %d: per step [start, issue done, wait done, barrier done] in cycles since wave 0's step 0 (100 MHz? no: s_memtime = shader clock)\n", wg * 50);
            for (int wv = 0; wv < waves; wv += waves - 1)
                for (int v = 0; v < 20; ++v) {
                    const unsigned long long *q = b + ((size_t)wv * 32 + v) * 4;
                    if (!q[0]) break;
                    printf("  wave %d step %2d: %7lld %7lld %7lld %7lld   compute %5lld wait %5lld barrier %5lld\n", wv, v, (long long)(q[0] - t0), (long long)(q[1] - t0), (long long)(q[2] - t0),
                           (long long)(q[3] - t0), (long long)(q[1] - q[0]), (long long)(q[2] - q[1]), (long long)(q[3] - q[2]));
                }
        }
        return 0;
    }
#endif
    // ---- timing: cold weights (cycling through copies > the 256 MiB Infinity Cache)
    std::vector<int> Ms;
    for (int i = 2; i < argc; ++i) Ms.push_back(atoi(argv[i]));
    if (Ms.empty()) Ms = {8, 32, 64, 128, 256};
    struct Shape { const char *name; int N, K, epi; } shapes[] = {{"qkv", 6144, 4096, W4R_STORE}, {"o_proj", 4096, 4096, W4R_STORE}, {"gate_up", 28672, 4096, W4R_SWIGLU}, {"down", 4096, 14336, W4R_STORE}};
    u16 *x, *y;
    float *ws;
    hipMalloc(&x, 256 * 14336 * 2), hipMalloc(&y, (size_t)256 * 28672 * 2), hipMalloc(&ws, (size_t)16 * 256 * 6144 * 4);
    {
        std::vector<u16> hx((size_t)256 * 14336);
        for (auto &v : hx) v = f2bf((float)((rand() % 2001) - 1000) / 1000.0f);
        hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    }
    for (auto &s : shapes) {
        const size_t bytes = w4m_bytes(s.N, s.K);
        const int copies = (int)((size_t)(320u << 20) / bytes) + 2;
        char *w;
        hipMalloc(&w, bytes * copies);
        const auto hw = make_tiles(s.N, s.K, 5);
        for (int i = 0; i < copies; ++i) hipMemcpy(w + bytes * i, hw.data(), bytes, hipMemcpyHostToDevice);
        for (int M : Ms) {
            int slabs = 0;
            const double us_new = time_it(2 * copies, [&](int i) { w4r_gemm_launch(PIE_BF16, w + bytes * (i % copies), x, M, s.N, s.K, y, ws, 0, s.epi, nullptr, nullptr, &slabs, nullptr, g_plain); });
            double us_old;
            if (M <= 32) {
                us_old = time_it(2 * copies, [&](int i) { w4m_gemm_launch(PIE_BF16, w + bytes * (i % copies), x, M, s.N, s.K, y, 0, s.epi == W4R_SWIGLU ? 1 : 0, nullptr, nullptr); });
            } else {
                bool fused = false;
                int sl = 0;
                us_old = time_it(2 * copies, [&](int i) { w4l_gemm_launch(PIE_BF16, w + bytes * (i % copies), x, M, s.N, s.K, y, ws, 0, s.epi == W4R_SWIGLU ? (void *)y : nullptr, &fused, &sl); });
            }
            const W4rPlan pl = w4r_plan(M, s.N, s.K, s.epi == W4R_STORE);
            const double flop = 2.0 * M * s.N * s.K;
            printf("ABL=%d M=%3d %-8s N=%5d K=%5d  new %7.2f us (%5.2f TB/s, %5.2f PFLOP/s; mb=%d S=%d)   old %7.2f us   x%.2f\n", W4R_ABL, M, s.name, s.N, s.K, us_new,
                   bytes / us_new / 1e6, flop / us_new / 1e9, pl.mb, pl.S, us_old, us_old / us_new);
            fflush(stdout);
        }
        hipFree(w);
    }
    return 0;
}
