// tools/aql_probe_kernels.hip -- device side of tools/aql_probe.cpp (developer tool): built to a code object with
//   hipcc --genco --offload-arch=gfx950 -O3 tools/aql_probe_kernels.hip -o tools/aql_probe_kernels.hsaco
// No blockDim / gridDim (they live in the hidden kernel arguments, which the probe does not fill).
#include <hip/hip_runtime.h>

extern "C" __global__ void __launch_bounds__(512) k_trivial(unsigned *p) {
    if (p && threadIdx.x == 100000u) p[0] = 1u;
}

// One link of a dependency chain shaped like a decode GEMV's staging: every workgroup reads the WHOLE input vector (4096 words, written
// by all 256 workgroups of the previous link), reduces it, and writes its own 16 words of the output vector.
template <bool SC1>
__device__ __forceinline__ void link_body(const unsigned *in, unsigned *out, unsigned salt) {
    __shared__ unsigned red[8];
    unsigned v[8], acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned *p = in + threadIdx.x + 512u * i;
        v[i] = SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i] * (2u * (threadIdx.x + 512u * i) + 1u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i];
    if (threadIdx.x < 16u) {
        const unsigned idx = blockIdx.x * 16u + threadIdx.x;
        const unsigned own = SC1 ? __hip_atomic_load(in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : in[idx];
        const unsigned r = tot * 1664525u + own + salt + threadIdx.x;
        if (SC1) __hip_atomic_store(out + idx, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[idx] = r;
    }
}
extern "C" __global__ void __launch_bounds__(512) k_link_plain(const unsigned *in, unsigned *out, unsigned salt) { link_body<false>(in, out, salt); }
extern "C" __global__ void __launch_bounds__(512) k_link_sc1(const unsigned *in, unsigned *out, unsigned salt) { link_body<true>(in, out, salt); }
