// tools/launch_cost.hip -- cost of dependent kernel boundaries inside a replayed hipGraph (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_tiny(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void __launch_bounds__(512) k_wide(int *p, int n) {   // many WGs, trivial work, small dirty footprint
    extern __shared__ char smem[];
    if (threadIdx.x == 0) p[(blockIdx.x * 64) % n] = blockIdx.x + (int)smem[0] * 0;
}
__global__ void __launch_bounds__(512) k_stream(const uint4 *src, size_t n16, int *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) sink[0] = acc;
}
template <class F> int time_graph(const char *name, int n_kernels, F enqueue) {
    hipStream_t cs; CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n_kernels; ++i) enqueue(cs, i);
    CK(hipStreamEndCapture(cs, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, 0));
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.2f us per kernel (%d kernels per graph)\n", name, 1e3 * ms / reps / n_kernels, n_kernels);
    return 0;
}
int main() {
    int *p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    uint4 *big; const size_t BIG = (size_t)1 << 30; CK(hipMalloc(&big, BIG)); CK(hipMemset(big, 1, BIG));
    time_graph("tiny (1 WG x 64 threads)", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, p); });
    time_graph("wide 256 WG x 512 thr", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_wide, dim3(256), dim3(512), 0, s, p, 1 << 18); });
    time_graph("wide 512 WG x 512 thr, 12 KB LDS", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_wide, dim3(512), dim3(512), 12288, s, p, 1 << 18); });
    time_graph("wide 2048 WG x 512 thr", 200, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_wide, dim3(2048), dim3(512), 0, s, p, 1 << 18); });
    for (size_t mb : {9, 14, 33, 66}) {
        char nm[64]; snprintf(nm, 64, "stream %zu MB per kernel (cycling 1 GB)", mb);
        const size_t n16 = (mb << 20) / 16, slots = BIG / (mb << 20);
        time_graph(nm, 60, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_stream, dim3(1024), dim3(512), 0, s, big + (size_t)(i % slots) * n16, n16, p); });
    }
    return 0;
}
