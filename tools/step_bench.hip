// tools/step_bench.hip -- C++ driver of the decode step through the C ABI (developer / profiling tool, not part of the product).
//
// Builds a synthetic Llama-shaped int4 g=64 model (random weights, quantised and repacked on the device through
// pie_quantize_w4g64 / pie_repack_w4g64), then replays pie_decoder_step.  Because it is a plain binary it can sit directly
// after `rocprofv3 ... --` (kernel trace or --pmc passes on the PRODUCT step; a Python host crashed the profiler in round 1).
//   step_bench [--model 8b|70b|tiny] [--layers N] [--steps K] [--warmup W] [--ctx P] [--cap C] [--graph 0|1] [--check N]
//              [--prefill N [--prefill-reps R]]
// --check N: runs N steps eagerly and N as the replayed graph from the same state and compares logits / logprobs / tokens / hidden state
//            bit for bit.  (The persistent one-launch step this tool used to compare against lives under tools/engine/ since round 4.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/step_bench.hip -Iinclude -Lproxy_inference_engine_amd/lib -lpie_hip \
//        -Wl,-rpath,'$ORIGIN/../proxy_inference_engine_amd/lib' -o tools/step_bench
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pie_hip.h"
extern "C" void *pie_debug_buffer(pie_decoder *d, int which);

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);      \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)
#define PK(x)                                                                        \
    do {                                                                             \
        int rc_ = (x);                                                               \
        if (rc_ != 0) {                                                              \
            printf("pie error %d (%s) at %s:%d\n", rc_, pie_last_error(), __FILE__, __LINE__); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef unsigned short u16;

// bf16 values in (-amp, amp) + offset from a counter hash (deterministic, no host traffic)
__global__ void k_fill_bf16(u16 *out, size_t n, unsigned seed, float amp, float offset) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u ^ seed * 0x9E3779B9u;
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    const float f = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * amp + offset;
    const unsigned u = __float_as_uint(f);
    out[i] = (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

static void fill(u16 *p, size_t n, unsigned seed, float amp, float offset = 0.0f) {
    hipLaunchKernelGGL(k_fill_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed, amp, offset);
    CK(hipGetLastError());
}

struct Geo {
    int H, I, heads, kv, D, V, L;
};

static void *dmalloc(size_t b) {
    void *p;
    CK(hipMalloc(&p, b));
    return p;
}

// one [N, K] Linear -> W4S (optionally row-mapped); scratch buffers are reused
struct Quant {
    u16 *w = nullptr;
    uint32_t *codes = nullptr;
    u16 *scales = nullptr, *biases = nullptr;
    size_t cap = 0;
    void reserve(size_t n_elems) {
        if (n_elems <= cap) return;
        if (w) { CK(hipFree(w)); CK(hipFree(codes)); CK(hipFree(scales)); CK(hipFree(biases)); }
        cap = n_elems;
        w = (u16 *)dmalloc(cap * 2), codes = (uint32_t *)dmalloc(cap), scales = (u16 *)dmalloc(cap / 32), biases = (u16 *)dmalloc(cap / 32);
    }
    int bits = 4;  // --bits 2: MLX int2 g=64 triplets on W2S units (every Linear; the embedding table stays 4-bit codes)
    void *pack(int N, int K, unsigned seed, float amp, const int32_t *row_map_dev) {
        reserve((size_t)N * K);
        fill(w, (size_t)N * K, seed, amp);
        if (bits == 2 || bits == 6) {
            PK(pie_quantize_g64(w, N, K, bits, PIE_BF16, codes, scales, biases, nullptr));
            void *packed2 = dmalloc(bits == 2 ? pie_w2s_bytes(N, K) : pie_w6s_bytes(N, K));
            PK((bits == 2 ? pie_repack_w2g64 : pie_repack_w6g64)(codes, scales, biases, N, K, row_map_dev, N, packed2, nullptr));
            return packed2;
        }
        PK(pie_quantize_w4g64(w, N, K, PIE_BF16, codes, scales, biases, nullptr));
        void *packed = dmalloc(pie_w4s_bytes(N, K));
        PK(pie_repack_w4g64(codes, scales, biases, N, K, row_map_dev, N, packed, nullptr));
        return packed;
    }
};

int main(int argc, char **argv) {
    Geo g = {4096, 14336, 32, 8, 128, 128256, 32};
    int steps = 50, warmup = 10, ctx = 128, cap = 512, check = 0, graph = 1, kv_splits = 0, prefill = 0, prefill_reps = 3, no_mega = 0, sync_every = 0, heads = -1, bits = 4;
    std::string mode = "both";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : (char *)"0"; };
        if (a == "--model") {
            std::string m = next();
            if (m == "70b") g = {8192, 28672, 64, 8, 128, 128256, 80};
            else if (m == "tiny") g = {512, 1536, 8, 2, 64, 4096, 4};
            else if (m == "3b") g = {3072, 8192, 24, 8, 128, 128256, 28};
        } else if (a == "--layers") g.L = atoi(next());
        else if (a == "--steps") steps = atoi(next());
        else if (a == "--bits") bits = atoi(next());                    // 2 / 6: W2S / W6S units (int2 / int6 g=64), 4: W4S
        else if (a == "--warmup") warmup = atoi(next());
        else if (a == "--ctx") ctx = atoi(next());
        else if (a == "--cap") cap = atoi(next());
        else if (a == "--mode") mode = next();
        else if (a == "--graph") graph = atoi(next());
        else if (a == "--check") check = atoi(next());
        else if (a == "--kv-splits") kv_splits = atoi(next());
        else if (a == "--heads") heads = atoi(next());                  // accepted and ignored (older scripts): the per-q-head attention plan moved to tools/engine/
        else if (a == "--sync-every") sync_every = atoi(next());       // under rocprofv3: bound the dispatches in flight (thousands of queued
                                                                       // graph nodes overran the profiler: SIGSEGV in its interception)
        else if (a == "--no-mega") no_mega = 1;                        // accepted and ignored (older scripts)
        else if (a == "--prefill") prefill = atoi(next());            // time pie_decoder_prefill of N tokens instead of decode steps
        else if (a == "--prefill-reps") prefill_reps = atoi(next());
    }
    if (prefill > 0 && cap < prefill) cap = (prefill + 255) / 256 * 256;
    if (ctx + steps + warmup + 2 > cap) steps = cap - ctx - warmup - 2 > 1 ? cap - ctx - warmup - 2 : 1;
    char name[64];
    int n_cus = 0;
    size_t hbm = 0;
    PK(pie_device_info(name, sizeof name, &n_cus, &hbm));
    {
        // through dlsym: &pie_hello in this executable is its PLT stub, which dladdr attributes to step_bench itself
        Dl_info di;
        void *sym = dlsym(RTLD_DEFAULT, "pie_hello");
        if (sym && dladdr(sym, &di) && di.dli_fname) printf("library %s (%s)\n", di.dli_fname, pie_version());
    }
    printf("device %s, %d CUs, %.0f GB; model H=%d I=%d heads=%d/%d D=%d V=%d L=%d; ctx %d cap %d\n", name, n_cus, hbm / 1e9, g.H, g.I, g.heads, g.kv,
           g.D, g.V, g.L, ctx, cap);

    const int QD = g.heads * g.D, KVD = g.kv * g.D, NQ = QD + 2 * KVD;
    std::vector<int32_t> qmap(NQ), gmap(2 * g.I);
    PK(pie_qkv_row_map(g.heads, g.kv, g.D, qmap.data()));
    PK(pie_gateup_row_map(g.I, gmap.data()));
    int32_t *qmap_d = (int32_t *)dmalloc(NQ * 4), *gmap_d = (int32_t *)dmalloc(2 * g.I * 4);
    CK(hipMemcpy(qmap_d, qmap.data(), NQ * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gmap_d, gmap.data(), 2 * g.I * 4, hipMemcpyHostToDevice));

    pie_decoder_config cfg = {};
    cfg.dtype = PIE_BF16, cfg.hidden = g.H, cfg.n_layers = g.L, cfg.n_heads = g.heads, cfg.n_kv_heads = g.kv, cfg.head_dim = g.D, cfg.inter = g.I,
    cfg.vocab = g.V, cfg.rms_eps = 1e-5f, cfg.weight_format = PIE_W_INT4_G64, cfg.kv_splits = kv_splits;
    pie_decoder *dec = nullptr;
    PK(pie_decoder_create(&cfg, &dec));

    Quant q;
    q.bits = bits == 2 || bits == 6 ? bits : 4;
    const int lin_fmt = bits == 2 ? PIE_W_INT2_G64 + 1 : (bits == 6 ? PIE_W_INT6_G64 + 1 : 0);  // per-matrix format code (0: the decoder's default, int4 g=64)
    const float amp = 1.7f / sqrtf((float)g.H);  // uniform(-a, a): std a / sqrt(3) ~ 1 / sqrt(H)
    for (int l = 0; l < g.L; ++l) {
        pie_layer_weights lw = {};
        u16 *n1 = (u16 *)dmalloc(g.H * 2), *n2 = (u16 *)dmalloc(g.H * 2);
        fill(n1, g.H, 1000 + l, 0.1f, 1.0f), fill(n2, g.H, 2000 + l, 0.1f, 1.0f);
        lw.attn_norm = n1, lw.mlp_norm = n2;
        lw.wqkv = q.pack(NQ, g.H, 10 * l + 1, amp, qmap_d);
        lw.wo = q.pack(g.H, QD, 10 * l + 2, amp, nullptr);
        lw.wgateup = q.pack(2 * g.I, g.H, 10 * l + 3, amp, gmap_d);
        lw.wdown = q.pack(g.H, g.I, 10 * l + 4, 1.7f / sqrtf((float)g.I), nullptr);
        lw.fmt_qkv = lw.fmt_o = lw.fmt_gateup = lw.fmt_down = lin_fmt;
        PK(pie_decoder_set_layer(dec, l, &lw));
    }
    pie_global_weights gw = {};
    {
        q.reserve((size_t)g.V * g.H);
        fill(q.w, (size_t)g.V * g.H, 777, 0.05f);
        uint32_t *ec = (uint32_t *)dmalloc((size_t)g.V * g.H / 2);
        u16 *es = (u16 *)dmalloc((size_t)g.V * g.H / 32), *eb = (u16 *)dmalloc((size_t)g.V * g.H / 32);
        PK(pie_quantize_w4g64(q.w, g.V, g.H, PIE_BF16, ec, es, eb, nullptr));
        gw.embed_codes = ec, gw.embed_scales = es, gw.embed_biases = eb;
        gw.lm_head = q.pack(g.V, g.H, 888, amp, nullptr);
        u16 *fn = (u16 *)dmalloc(g.H * 2);
        fill(fn, g.H, 999, 0.1f, 1.0f);
        gw.final_norm = fn;
        std::vector<float> fr(g.D / 2);
        for (int i = 0; i < g.D / 2; ++i) fr[i] = powf(500000.0f, (float)(2 * i) / (float)g.D);
        float *fd = (float *)dmalloc(g.D * 2);
        CK(hipMemcpy(fd, fr.data(), g.D * 2, hipMemcpyHostToDevice));
        gw.rope_freqs = fd;
    }
    gw.fmt_lm_head = lin_fmt;
    PK(pie_decoder_set_globals(dec, &gw));

    const size_t kv_bytes = (size_t)g.kv * cap * g.D * 2;
    std::vector<const void *> kp(g.L), vp(g.L);
    for (int l = 0; l < g.L; ++l) {
        u16 *k = (u16 *)dmalloc(kv_bytes), *v = (u16 *)dmalloc(kv_bytes);
        fill(k, kv_bytes / 2, 5000 + l, 1.0f), fill(v, kv_bytes / 2, 6000 + l, 1.0f);  // a "prompt" already in the caches
        kp[l] = k, vp[l] = v;
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    PK(pie_decoder_set_kv(dec, kp.data(), vp.data(), cap, st));
    u16 *logits = (u16 *)dmalloc((size_t)g.V * 2), *hidden = (u16 *)dmalloc(g.H * 2);
    float *logprobs = (float *)dmalloc((size_t)g.V * 4);
    int32_t *token = (int32_t *)dmalloc(4), *hist = (int32_t *)dmalloc(4 * 65536);
    PK(pie_decoder_bind_outputs(dec, logits, logprobs, token, hidden, hist, 65536));
    CK(hipDeviceSynchronize());
    (void)no_mega, (void)heads, (void)mode;
    struct Snap {
        std::vector<u16> logits, hidden;
        std::vector<float> logprobs;
        int token;
    };
    auto run = [&](int use_graph, int n, std::vector<Snap> *snaps) -> double {
        const int flags = PIE_STEP_LOGITS | (use_graph ? PIE_STEP_GRAPH : 0);
        PK(pie_decoder_set_state(dec, ctx, 1, st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        if (snaps) {
            for (int i = 0; i < n; ++i) {
                PK(pie_decoder_step(dec, flags, st));
                Snap s;
                s.logits.resize(g.V), s.hidden.resize(g.H), s.logprobs.resize(g.V);
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(s.logits.data(), logits, (size_t)g.V * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(s.hidden.data(), hidden, g.H * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(s.logprobs.data(), logprobs, (size_t)g.V * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&s.token, token, 4, hipMemcpyDeviceToHost));
                snaps->push_back(std::move(s));
            }
            return 0.0;
        }
        for (int i = 0; i < warmup; ++i) PK(pie_decoder_step(dec, flags, st));
        CK(hipStreamSynchronize(st));
        PK(pie_decoder_set_state(dec, ctx, 1, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) {
            PK(pie_decoder_step(dec, flags, st));
            if (sync_every > 0 && (i + 1) % sync_every == 0) CK(hipStreamSynchronize(st));
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err = 0;
        PK(pie_decoder_status(dec, &err));
        if (err) printf("  !! a bounded wait gave up: %u\n", err);
        return ms / n;
    };

    int rc = 0;
    if (prefill > 0) {  // the batched prompt path (many-row int4 MFMA GEMM, causal attention): for rocprofv3 --kernel-trace / --pmc passes
        std::vector<int32_t> ids(prefill);
        for (int i = 0; i < prefill; ++i) ids[i] = (int32_t)((1315423911u * (unsigned)(i + 1)) % (unsigned)g.V);
        int32_t *ids_d = (int32_t *)dmalloc((size_t)prefill * 4);
        CK(hipMemcpy(ids_d, ids.data(), (size_t)prefill * 4, hipMemcpyHostToDevice));
        for (int r = 0; r < prefill_reps + 1; ++r) {  // first pass: one-off tile repack of the weights
            PK(pie_decoder_set_state(dec, 0, -1, st));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            PK(pie_decoder_prefill(dec, ids_d, prefill, nullptr, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("prefill of %d tokens, pass %d%s: %8.3f ms  (%.0f tokens/s)\n", prefill, r, r ? "" : " (includes the one-off W4M repack)", ms, prefill / ms * 1e3);
        }
        PK(pie_decoder_destroy(dec));
        return 0;
    }
    if (check > 0) {
        std::vector<Snap> a, b;
        run(0, check, &a);
        run(1, check, &b);
        for (int i = 0; i < check; ++i) {
            size_t dl = 0, dh = 0, dp = 0;
            for (int j = 0; j < g.V; ++j) dl += a[i].logits[j] != b[i].logits[j], dp += memcmp(&a[i].logprobs[j], &b[i].logprobs[j], 4) != 0;
            for (int j = 0; j < g.H; ++j) dh += a[i].hidden[j] != b[i].hidden[j];
            printf("check step %d: token %d vs %d, differing logits %zu / %d, hidden %zu / %d, logprobs %zu\n", i, a[i].token, b[i].token, dl, g.V, dh, g.H, dp);
            if (dl || dh || dp || a[i].token != b[i].token) rc = 1;
        }
        printf(rc ? "CHECK FAILED\n" : "check ok: graph replay == eager launches (logits, log-probabilities, tokens, hidden state bit for bit)\n");
    }
    const double bytes = (double)pie_decoder_step_bytes(dec, ctx + steps / 2, 1);
    {
        const double ms = run(graph, steps, nullptr);
        printf("launch sequence : %8.3f ms/step  %7.1f tok/s  %6.2f TB/s  (%.1f %% of 8 TB/s)\n", ms, 1e3 / ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0 * 100);
    }
    if (void *ps = pie_debug_buffer(dec, 7)) {  // -DPIE_ATTN_PROF build of the library: stamps of the last attention launch
        unsigned long long t[32] = {};
        CK(hipMemcpy(t, ps, sizeof t, hipMemcpyDeviceToHost));
        const char *kn[4] = {"qkv", "o_proj", "gate_up", "down"};
        {  // -DPIE_GEMV_PROF=3 build: fine prologue stamps of workgroup 0 of the last launch of each GEMV (32-word slots from word 256)
            std::vector<unsigned long long> f(1024);
            CK(hipMemcpy(f.data(), ps, 8192, hipMemcpyDeviceToHost));
            for (int k = 0; k < 4; ++k) {
                const unsigned long long *q = f.data() + 256 + 32 * k + 8;
                if (q[0] && q[5] > q[0])
                    printf("%-8s prologue, workgroup 0, us after kernel entry: kernel arguments %.2f | activations arrived %.2f | norm reduced %.2f | image staged %.2f | end %.2f\n", kn[k],
                           (q[1] - q[0]) * 0.01, q[2] ? (q[2] - q[0]) * 0.01 : 0.0, q[3] ? (q[3] - q[0]) * 0.01 : 0.0, (q[4] - q[0]) * 0.01, (q[5] - q[0]) * 0.01);
            }
        }
        for (int k = 0; k < 4; ++k) {  // -DPIE_GEMV_PROF build: workgroup 0 of the last launch of each GEMV
            const unsigned long long *g4 = t + 16 + 4 * k;
            if (g4[0] && g4[3] > g4[0])
                printf("%-8s launch, workgroup 0, us after its start: x staged %.2f | stream done %.2f | epilogue done %.2f\n", kn[k], (g4[1] - g4[0]) * 0.01,
                       (g4[2] - g4[0]) * 0.01, (g4[3] - g4[0]) * 0.01);
        }
        {  // -DPIE_GEMV_PROF=2: start / end of every workgroup of the last gate/up launch
            std::vector<unsigned long long> w(64 + 512);
            CK(hipMemcpy(w.data(), (const unsigned long long *)ps + 24, w.size() * 8, hipMemcpyDeviceToHost));  // gate/up's stamp block starts at word 24
            for (int wg = 0; wg < 3; ++wg)
                if (w[32 + 8 * wg]) {
                    printf("  gate_up workgroup %d, stream end of its 8 waves, us after the launch's first start:", wg == 0 ? 0 : (wg == 1 ? 100 : 201));
                    for (int k = 0; k < 8; ++k) printf(" %.2f", (double)(long long)(w[32 + 8 * wg + k] - w[64]) * 0.01);
                    printf("\n");
                }
            unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
            int n = 0;
            for (int b = 0; b < 256; ++b)
                if (w[64 + 2 * b] && w[65 + 2 * b]) {
                    ++n;
                    s0 = std::min(s0, w[64 + 2 * b]), s1 = std::max(s1, w[64 + 2 * b]), e0 = std::min(e0, w[65 + 2 * b]), e1 = std::max(e1, w[65 + 2 * b]);
                }
            if (n) {
                printf("gate_up: %d workgroups; starts spread over %.2f us; first end %.2f us, last end %.2f us after the first start\n", n, (s1 - s0) * 0.01, (e0 - s0) * 0.01,
                       (e1 - s0) * 0.01);
                int hist[12] = {};
                for (int b = 0; b < 256; ++b)
                    if (w[65 + 2 * b]) {
                        int k = (int)((w[65 + 2 * b] - e0) * 0.01 / 0.25);
                        hist[k < 11 ? k : 11]++;
                    }
                {  // which workgroups end last, and the mean end per XCD (blockIdx % 8 under round-robin placement)
                    std::vector<std::pair<unsigned long long, int>> o;
                    double xs[8] = {}, xn[8] = {};
                    for (int b = 0; b < 256; ++b)
                        if (w[65 + 2 * b]) o.push_back({w[65 + 2 * b], b}), xs[b % 8] += (w[65 + 2 * b] - s0) * 0.01, xn[b % 8] += 1;
                    std::sort(o.begin(), o.end());
                    printf("  last 12 to end:");
                    for (size_t k = o.size() >= 12 ? o.size() - 12 : 0; k < o.size(); ++k) printf(" %d", o[k].second);
                    printf("\n  first 12 to end:");
                    for (size_t k = 0; k < 12 && k < o.size(); ++k) printf(" %d", o[k].second);
                    printf("\n  mean end per blockIdx %% 8:");
                    for (int k = 0; k < 8; ++k) printf(" %.2f", xn[k] ? xs[k] / xn[k] : 0.0);
                    printf("\n");
                }
                printf("  ends, 0.25 us bins after the first end:");
                for (int k = 0; k < 12; ++k) printf(" %d", hist[k]);
                printf("\n");
            }
        }
        const unsigned long long *q = t + 2;
        if (!q[0] && q[1] && t[16]) {  // the attention ran behind the q|k|v launch's seam: one timeline, from that launch's start (workgroup 0 = kv-group 0, split 0)
            const unsigned long long s0 = t[16];
            printf("fused q|k|v + attention, workgroup 0, us after its start: x staged %.2f | stream done %.2f | epilogue done %.2f | position read %.2f | released %.2f | "
                   "K/V rows + q landed %.2f | scoring done %.2f | streams in LDS %.2f | barrier %.2f | partials stored %.2f\n",
                   (t[17] - s0) * 0.01, (t[18] - s0) * 0.01, (t[19] - s0) * 0.01, (q[1] - s0) * 0.01, (q[7] - s0) * 0.01, (q[2] - s0) * 0.01, (q[3] - s0) * 0.01, (q[4] - s0) * 0.01,
                   (q[5] - s0) * 0.01, (q[6] - s0) * 0.01);
        }
        if (q[0] && q[6] > q[0])
            printf("attention launch, workgroup (0,0,0), us after its start: position arrived %.2f | first K/V rows %.2f | scoring done %.2f | wave merge %.2f | "
                   "barrier %.2f | partials stored %.2f\n", (q[1] - q[0]) * 0.01, (q[2] - q[0]) * 0.01, (q[3] - q[0]) * 0.01, (q[4] - q[0]) * 0.01, (q[5] - q[0]) * 0.01,
                   (q[6] - q[0]) * 0.01);
    }
    PK(pie_decoder_destroy(dec));
    return rc;
}
