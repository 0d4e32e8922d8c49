// tools/w4m_bench.hip -- times k_w4m_gemm alone on the 8B model's four layer shapes (developer tool, not part of the product).
// Build one binary per ablation mask:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DW4M_ABL=<mask> tools/w4m_bench.hip -o tools/w4m_bench_<mask>
#include <cstdio>
#include <string>
#include <vector>

#include "../proxy_inference_engine_amd/csrc/w4m_gemm.hip"

int bias_any_launch(int, void *, const void *, int, int, hipStream_t) { return 0; }  // vision.hip's; not reached from here
int pie_knob(int) { return -1; }  // decoder.hip's knob table: every knob at its default
namespace pie {
int fail(int code, const std::string &msg) {
    std::fprintf(stderr, "error %d: %s\n", code, msg.c_str());
    return code;
}
}  // namespace pie

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8;
    struct Shape { const char *name; int N, K; } shapes[] = {{"qkv", 6144, 4096}, {"o_proj", 4096, 4096}, {"gate_up", 28672, 4096}, {"down", 4096, 14336}};
    const int copies = 24;  // cycle through copies so every launch streams cold weights (24 x 66 MB > the 256 MiB Infinity Cache)
    u16 *x, *y;
    hipMalloc(&x, 32 * 14336 * 2), hipMalloc(&y, 32 * 28672 * 2);
    hipMemset(x, 0x3c, 32 * 14336 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (auto &s : shapes) {
        const size_t bytes = w4m_bytes(s.N, s.K);
        char *w;
        hipMalloc(&w, bytes * copies);
        hipMemset(w, 0x21, bytes * copies);
        for (int i = 0; i < copies; ++i) w4m_gemm_launch(PIE_BF16, w + bytes * i, x, M, s.N, s.K, y, nullptr, 0, nullptr, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 4;
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < copies; ++i) w4m_gemm_launch(PIE_BF16, w + bytes * i, x, M, s.N, s.K, y, nullptr, 0, nullptr, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / (reps * copies);
        std::printf("ABL=%d M=%d %-8s N=%5d K=%5d  %7.2f us  %6.2f TB/s\n", W4M_ABL, M, s.name, s.N, s.K, us, bytes / us / 1e6);
        hipFree(w);
    }
    return 0;
}
