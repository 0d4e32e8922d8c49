// tools/kernarg_probe.hip -- does kernarg preloading (first scalar arguments delivered in SGPRs at wave launch, gfx950) shorten the fixed
// cost of a small dependent launch?  A chain of N dependent launches inside a hipGraph, each reading a device "state" word and one element
// of the previous launch's output: (a) all arguments in one by-value struct (loaded with s_load at wave start: one memory hop before
// anything else can issue), (b) the same values as leading scalar arguments, compiled with -mllvm -amdgpu-kernarg-preload-count=16.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/kernarg_probe.hip -o tools/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Args { const int *state; const float *in; float *out; int n; float s; };
__global__ void __launch_bounds__(256) k_struct(const Args a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int pos = a.state[0];
    if (i < a.n) a.out[i] = a.in[(i + pos) % a.n] * a.s;
}
__global__ void __launch_bounds__(256) k_scalar(const int *state, const float *in, float *out, int n, float s) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int pos = state[0];
    if (i < n) out[i] = in[(i + pos) % n] * s;
}
int main() {
    const int n = 65536, chain = 512;
    float *a, *b;
    int *state;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&state, 64));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(state, 0, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < chain; ++i) {
            const float *in = i & 1 ? b : a; float *out = i & 1 ? a : b;
            if (mode == 0) { Args ar = {state, in, out, n, 1.0f}; hipLaunchKernelGGL(k_struct, dim3(n / 256), dim3(256), 0, st, ar); }
            else hipLaunchKernelGGL(k_scalar, dim3(n / 256), dim3(256), 0, st, (const int *)state, in, out, n, 1.0f);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 20;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.3f us per dependent launch\n", mode == 0 ? "by-value struct (s_load at wave start)" : "leading scalars (kernarg preload)  ", ms * 1e3 / (reps * chain));
    }
    return 0;
}
