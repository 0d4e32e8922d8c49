// tools/w16_bench.hip -- the 16-bit many-row MFMA GEMM (k_w16l_gemm, w16_gemm.hpp) alone: parity against a plain double-precision
// reference kernel, and timing against hipBLASLt (the library GEMM it replaces; best of the heuristic's 8 candidates, as the product
// picked them) on the Llama-3-8B layer shapes and the Qwen2.5-VL-7B vision tower's shapes.  Developer tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DW16L_ABL=<mask>] tools/w16_bench.hip -o tools/w16_bench -lhipblaslt
//   tools/w16_bench check                 parity over a grid of (M, N, K) incl. ragged M / N / K, K splits, bias, SwiGLU, every tile plan
//   tools/w16_bench time [M ...]          us per launch and PFLOP/s per shape, own kernel vs hipBLASLt   (W16_PLAN="mb sw S" overrides the plan)
#include <hipblaslt/hipblaslt.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../proxy_inference_engine_amd/csrc/w4m_gemm.hip"

__global__ void k_bias_tool(u16 *y, const u16 *bias, int M, int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N) return;
    y[i] = BF16::from_f32(BF16::to_f32(y[i]) + BF16::to_f32(bias[i % N]));
}
int bias_any_launch(int, void *y, const void *bias, int M, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_bias_tool, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, st, (u16 *)y, (const u16 *)bias, M, N);
    return 0;
}
int pie_knob(int) { return -1; }
namespace pie {
int fail(int code, const std::string &msg) {
    std::fprintf(stderr, "error %d: %s\n", code, msg.c_str());
    return code;
}
}  // namespace pie

static inline void ck_(hipError_t e, const char *file, int line) {
    if (e != hipSuccess) {
        std::fprintf(stderr, "%s:%d: %s\n", file, line, hipGetErrorString(e));
        std::exit(2);
    }
}
#define CK(e) ck_((e), __FILE__, __LINE__)

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

// reference: ref[m][n] = sum_k x[m][k] w[n][k] in double, sumabs the same over |.|; x rows are Kx long, w rows K
__global__ void k_ref(const u16 *w, const u16 *x, int M, int N, int K, int Kx, float *ref, float *sumabs) {
    const int nn = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (nn >= N) return;
    double acc = 0.0, sa = 0.0;
    for (int k = 0; k < K; ++k) {
        const double p = (double)BF16::to_f32(w[(size_t)nn * K + k]) * (double)BF16::to_f32(x[(size_t)m * Kx + k]);
        acc += p, sa += fabs(p);
    }
    ref[(size_t)m * N + nn] = (float)acc, sumabs[(size_t)m * N + nn] = (float)sa;
}

static unsigned long long g_seed = 0x9E3779B97F4A7C15ull;
static float rnd() {  // uniform in [-1, 1)
    g_seed ^= g_seed << 13, g_seed ^= g_seed >> 7, g_seed ^= g_seed << 17;
    return (float)((g_seed >> 11) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

struct Problem {
    int M, N, K, Kx;
    u16 *w = nullptr, *x = nullptr, *y = nullptr, *bias = nullptr, *act = nullptr;
    void *w16m = nullptr, *ws = nullptr;
    float *ref = nullptr, *sumabs = nullptr;
};

static Problem make(int M, int N, int K, bool with_ref) {
    Problem p;
    p.M = M, p.N = N, p.K = K, p.Kx = 64 * ((K + 63) / 64);
    std::vector<u16> hw((size_t)N * K), hx((size_t)M * p.Kx, 0), hb(N);
    const bool zero = std::getenv("W16_ZERO") != nullptr;  // all-zero operands: the data-dependent share of the power
    for (auto &v : hw) v = f2bf(zero ? 0.0f : rnd() * 0.05f);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) hx[(size_t)m * p.Kx + k] = f2bf(zero ? 0.0f : rnd());
    for (auto &v : hb) v = f2bf(rnd() * 0.5f);
    CK(hipMalloc((void **)&p.w, hw.size() * 2)), CK(hipMalloc((void **)&p.x, hx.size() * 2)), CK(hipMalloc((void **)&p.y, (size_t)M * N * 2));
    CK(hipMalloc((void **)&p.bias, N * 2)), CK(hipMalloc((void **)&p.act, (size_t)M * N));
    CK(hipMemcpy(p.w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)), CK(hipMemcpy(p.x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(p.bias, hb.data(), N * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&p.w16m, w16m_size(N, K)));
    if (w16m_from_rows_launch(p.w, N, K, p.w16m, 0)) std::exit(2);
    const size_t wsb = with_ref ? (size_t)16 * M * N * 4 : w16l_workspace_bytes(M, N, K);  // (check: any overridden split)
    CK(hipMalloc(&p.ws, wsb ? wsb : 16));
    if (with_ref) {
        CK(hipMalloc((void **)&p.ref, (size_t)M * N * 4)), CK(hipMalloc((void **)&p.sumabs, (size_t)M * N * 4));
        hipLaunchKernelGGL(k_ref, dim3((N + 127) / 128, M), dim3(128), 0, 0, p.w, p.x, M, N, K, p.Kx, p.ref, p.sumabs);
    }
    CK(hipDeviceSynchronize());
    return p;
}
static void release(Problem &p) {
    void *ptrs[] = {p.w, p.x, p.y, p.bias, p.act, p.w16m, p.ws, p.ref, p.sumabs};
    for (void *q : ptrs)
        if (q) CK(hipFree(q));
}

static int check_one(int M, int N, int K, int mode, int mb, int sw, int S) {  // mode 0 store, 1 bias, 2 swiglu
    Problem p = make(M, N, K, true);
    g_w16_plan_override[0] = mb, g_w16_plan_override[1] = sw, g_w16_plan_override[2] = S;
    CK(hipMemset(p.y, 0xFF, (size_t)M * N * 2)), CK(hipMemset(p.act, 0xFF, (size_t)M * N));
    bool fused = false;
    const int rc = w16l_gemm_launch(PIE_BF16, p.w16m, p.x, 0, M, N, K, p.y, p.ws, 0, mode == 1 ? p.bias : nullptr, mode == 2 ? p.act : nullptr, mode == 2 ? &fused : nullptr, 0);
    CK(hipDeviceSynchronize());
    if (rc) std::exit(2);
    std::vector<float> ref((size_t)M * N), sa((size_t)M * N);
    std::vector<u16> y((size_t)M * N), hb(N);
    CK(hipMemcpy(ref.data(), p.ref, ref.size() * 4, hipMemcpyDeviceToHost)), CK(hipMemcpy(sa.data(), p.sumabs, sa.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), p.bias, N * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    double maxerr = 0;
    if (mode == 2 && fused) {
        std::vector<u16> act((size_t)M * (N / 2));
        CK(hipMemcpy(act.data(), p.act, act.size() * 2, hipMemcpyDeviceToHost));
        for (int m = 0; m < M; ++m)
            for (int i = 0; i < N / 2; ++i) {
                const float g = bf2f(f2bf(ref[(size_t)m * N + 2 * i])), u = bf2f(f2bf(ref[(size_t)m * N + 2 * i + 1]));
                const float want = bf2f(f2bf(bf2f(f2bf(g / (1.0f + expf(-g)))) * u));
                const float got = bf2f(act[(size_t)m * (N / 2) + i]);
                // a one-ulp difference of the rounded gate / up moves the product by up to ~2 ulp (and silu's slope <= 1.1)
                const float tol = 0x1p-6f * fabsf(want) + 0x1p-7f * (fabsf(g) + fabsf(u)) * fmaxf(1.0f, fabsf(u)) + 2e-3f * (sa[(size_t)m * N + 2 * i] + sa[(size_t)m * N + 2 * i + 1]) * fmaxf(1.0f, fabsf(u) + fabsf(g));
                const float err = fabsf(got - want);
                if (!(err <= tol)) ++bad;
                if (err > maxerr) maxerr = err;
            }
    } else {
        CK(hipMemcpy(y.data(), p.y, y.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < y.size(); ++i) {
            float want = ref[i];
            if (mode == 1) want = bf2f(f2bf(want)) + bf2f(hb[i % N]);
            const float got = bf2f(y[i]);
            const float tol = 0x1p-7f * (fabsf(want) + (mode == 1 ? fabsf(ref[i]) : 0.0f)) + 1e-3f * sa[i];
            const float err = fabsf(got - want);
            if (!(err <= tol)) ++bad;
            if (err > maxerr) maxerr = err;
        }
    }
    const W16Plan pl = w16_plan(M, N, K);
    std::printf("M=%5d N=%6d K=%6d mode=%d plan mb=%d sw=%d S=%d%s: max|err| %.3g, BAD %d\n", M, N, K, mode, pl.mb, pl.sw, pl.S, mode == 2 ? (fused ? " fused" : " NOT-fused") : "", maxerr, bad);
    std::fflush(stdout);
    release(p);
    g_w16_plan_override[0] = 0;
    return bad;
}

// ---- hipBLASLt, as the product used it: y[M, N] = x[M, K] . w[N, K]^T, best of the heuristic's 8 candidates
struct Lt {
    hipblasLtHandle_t h = nullptr;
    hipblasLtMatmulPreference_t pref = nullptr;
    void *ws = nullptr;
    size_t wsb = 64u << 20;
};
static Lt g_lt;
static double time_lt(const Problem &p, int reps) {
    if (!g_lt.h) {
        hipblasLtCreate(&g_lt.h), hipblasLtMatmulPreferenceCreate(&g_lt.pref);
        CK(hipMalloc(&g_lt.ws, g_lt.wsb));
        hipblasLtMatmulPreferenceSetAttribute(g_lt.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &g_lt.wsb, sizeof(g_lt.wsb));
    }
    hipblasLtMatmulDesc_t desc;
    hipblasLtMatrixLayout_t la, lb, lc;
    hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F);
    const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
    hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
    hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, p.K, p.N, p.K), hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, p.K, p.M, p.Kx), hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, p.N, p.M, p.N);
    hipblasLtMatmulHeuristicResult_t res[8];
    int n_res = 0;
    if (hipblasLtMatmulAlgoGetHeuristic(g_lt.h, desc, la, lb, lc, lc, g_lt.pref, 8, res, &n_res) != HIPBLAS_STATUS_SUCCESS || n_res == 0) return -1.0;
    const float one = 1.0f, zero = 0.0f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)), CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int i = 0; i < n_res; ++i) {
        bool ok = true;
        for (int r = 0; r < 2 && ok; ++r)
            ok = hipblasLtMatmul(g_lt.h, desc, &one, p.w, la, p.x, lb, &zero, p.y, lc, p.y, lc, &res[i].algo, g_lt.ws, g_lt.wsb, 0) == HIPBLAS_STATUS_SUCCESS;
        if (!ok) continue;
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) hipblasLtMatmul(g_lt.h, desc, &one, p.w, la, p.x, lb, &zero, p.y, lc, p.y, lc, &res[i].algo, g_lt.ws, g_lt.wsb, 0);
        CK(hipEventRecord(e1, 0)), CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / reps < best) best = ms / reps;
    }
    return best * 1e3;
}

static double time_own(const Problem &p, int reps, bool swiglu) {
    bool fused = false;
    for (int r = 0; r < 2; ++r) w16l_gemm_launch(PIE_BF16, p.w16m, p.x, 0, p.M, p.N, p.K, p.y, p.ws, 0, nullptr, swiglu ? p.act : nullptr, swiglu ? &fused : nullptr, 0);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)), CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) w16l_gemm_launch(PIE_BF16, p.w16m, p.x, 0, p.M, p.N, p.K, p.y, p.ws, 0, nullptr, swiglu ? p.act : nullptr, swiglu ? &fused : nullptr, 0);
    CK(hipEventRecord(e1, 0)), CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3;
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "check";
    if (const char *e = std::getenv("W16_BLOCK")) std::sscanf(e, "%d %d", &g_w16_block[0], &g_w16_block[1]);
    if (const char *e = std::getenv("W16_PLAN")) std::sscanf(e, "%d %d %d", &g_w16_plan_override[0], &g_w16_plan_override[1], &g_w16_plan_override[2]);
    if (mode == "check") {
        int bad = 0;
        const int plans[][3] = {{8, 2, 1}, {4, 2, 1}, {2, 2, 1}, {8, 1, 1}, {4, 1, 1}, {2, 1, 1}, {8, 2, 2}, {4, 1, 3}};
        for (auto &pl : plans) {
            bad += check_one(256, 256, 256, 0, pl[0], pl[1], pl[2]);
            bad += check_one(300, 520, 1216, 0, pl[0], pl[1], pl[2]);   // ragged everything; 19 groups
            bad += check_one(77, 1280, 3420, 1, pl[0], pl[1], pl[2]);   // the tower's odd width, with a bias
            if (pl[2] == 1) bad += check_one(130, 512, 512, 2, pl[0], pl[1], pl[2]);
        }
        bad += check_one(64, 214, 1176, 1, 0, 0, 0);   // any out_features: element-wise stores
        bad += check_one(5, 7, 64, 1, 0, 0, 0);
        bad += check_one(257, 100, 72, 1, 0, 0, 0);
        bad += check_one(512, 4096, 4096, 0, 0, 0, 0);
        bad += check_one(1000, 6840, 1280, 1, 0, 0, 0);
        bad += check_one(33, 1024, 2048, 0, 0, 0, 0);
        bad += check_one(512, 1024, 4096, 2, 0, 0, 0);
        bad += check_one(64, 4096, 14336, 0, 0, 0, 0);
        std::printf(bad ? "check FAILED (%d)\n" : "check ok\n", bad);
        return bad ? 1 : 0;
    }
    if (mode == "sweep") {  // every tile plan on one shape: tools/w16_bench sweep M N K
        const int M = std::atoi(argv[2]), N = std::atoi(argv[3]), K = std::atoi(argv[4]);
        Problem p = make(M, N, K, false);
        CK(hipFree(p.ws));
        CK(hipMalloc(&p.ws, (size_t)8 * M * N * 4));
        const W16Plan def = w16_plan(M, N, K);
        const double lt = time_lt(p, 30);
        std::printf("M=%d N=%d K=%d: default plan mb=%d sw=%d S=%d; hipBLASLt %.2f us\n", M, N, K, def.mb, def.sw, def.S, lt);
        for (int mb = 2; mb <= 8; mb *= 2)
            for (int sw = 1; sw <= 2; ++sw)
                for (int S = 1; S <= 8; ++S) {
                    if (S > 1 && 4 * S > (K + 63) / 64) continue;
                    if (32 * mb / 2 >= M + 32 * mb / 2 - 1 && mb > 2) continue;
                    g_w16_plan_override[0] = mb, g_w16_plan_override[1] = sw, g_w16_plan_override[2] = S;
                    if (w16_plan(M, N, K).S != S) continue;
                    const double us = time_own(p, 30, false);
                    std::printf("  mb=%d sw=%d S=%d: %8.2f us  x%.2f of hipBLASLt%s\n", mb, sw, S, us, lt / us, (mb == def.mb && sw == def.sw && S == def.S) ? "   <- default" : "");
                    std::fflush(stdout);
                }
        return 0;
    }
#ifdef W16L_PROF
    if (mode == "prof") {  // core clock and cycles per K step inside the kernel: tools/w16_bench prof M N K
        const int M = std::atoi(argv[2]), N = std::atoi(argv[3]), K = std::atoi(argv[4]);
        Problem p = make(M, N, K, false);
        const W16Plan pl = w16_plan(M, N, K);
        const int wgs = 8 * (((N + 128 * pl.sw - 1) / (128 * pl.sw) + 7) / 8) * ((M + 32 * pl.mb - 1) / (32 * pl.mb));
        CK(hipMalloc((void **)&g_w16_prof, (size_t)wgs * 32));
        for (int r = 0; r < 3; ++r) {
            CK(hipMemset(g_w16_prof, 0, (size_t)wgs * 32));
            w16l_gemm_launch(PIE_BF16, p.w16m, p.x, 0, M, N, K, p.y, p.ws, 0, nullptr, nullptr, nullptr, 0);
            CK(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> h((size_t)wgs * 4);
        CK(hipMemcpy(h.data(), g_w16_prof, h.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0;
        int cnt = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < wgs; ++b) {
            if (!h[4 * b + 2]) continue;
            cyc += (double)(h[4 * b + 2] - h[4 * b]), rt += (double)(h[4 * b + 3] - h[4 * b + 1]), ++cnt;
            if (h[4 * b + 1] < t0) t0 = h[4 * b + 1];
            if (h[4 * b + 3] > t1) t1 = h[4 * b + 3];
        }
        const int steps = ((K + 63) / 64 + 3) & ~3;
        std::printf("ABL=%d M=%d N=%d K=%d plan mb=%d sw=%d: %d workgroups, mean %.0f s_memtime ticks = %.2f us per workgroup -> %.1f ticks/us, %.0f ticks per K step (%d MFMAs per wave); kernel span %.1f us\n",
                    W16L_ABL, M, N, K, pl.mb, pl.sw, cnt, cyc / cnt, rt / cnt / 100.0, cyc / (rt / 100.0), cyc / cnt / steps, 4 * pl.mb * pl.sw, (double)(t1 - t0) / 100.0);
        return 0;
    }
#endif
    std::vector<int> Ms;
    for (int i = 2; i < argc; ++i) Ms.push_back(std::atoi(argv[i]));
    if (Ms.empty()) Ms = {512, 1024, 4096};
    struct Shape {
        const char *name;
        int N, K;
        bool swiglu;
    };
    const Shape shapes[] = {{"qkv", 6144, 4096, false},      {"o_proj", 4096, 4096, false},   {"gate_up", 28672, 4096, false}, {"gate_up+act", 28672, 4096, true},
                            {"down", 4096, 14336, false},    {"t_qkv", 3840, 1280, false},    {"t_proj", 1280, 2048, false},   {"t_gateup", 6840, 1280, false},
                            {"t_down", 1280, 3420, false},   {"t_merge0", 5120, 5120, false}, {"t_merge2", 3584, 5120, false}};
    for (int M : Ms)
        for (const Shape &s : shapes) {
            if (s.name[0] == 't' && s.name[2] == 'm' && M > 1024) continue;
            if (const char *only = std::getenv("W16_SHAPE"))
                if (std::string(s.name) != only) continue;
            Problem p = make(M, s.N, s.K, false);
            const double flop = 2.0 * M * s.N * s.K;
            const int reps = flop > 5e10 ? 10 : 50;
            const double own = time_own(p, reps, s.swiglu);
            const double lt = s.swiglu ? -1.0 : time_lt(p, reps);
            const W16Plan pl = w16_plan(M, s.N, s.K);
            std::printf("ABL=%d M=%5d %-12s N=%6d K=%6d  own %9.2f us (%5.3f PFLOP/s; mb=%d sw=%d S=%d)   hipBLASLt %9.2f us (%5.3f)   own/lt x%.2f\n", W16L_ABL, M, s.name, s.N, s.K, own,
                        flop / own / 1e9, pl.mb, pl.sw, pl.S, lt, lt > 0 ? flop / lt / 1e9 : 0.0, lt > 0 ? lt / own : 0.0);
            std::fflush(stdout);
            release(p);
        }
    return 0;
}
