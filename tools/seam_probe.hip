// tools/seam_probe.hip -- VERDICT r4 #2's gate, measured: what does an XCD-LOCAL 32-workgroup hand-off cost inside one launch?
//
// The design it would enable: the q|k|v rows of one kv-group (768 rows) computed by the 32 workgroups that run on one XCD, which then -- without
// leaving the launch -- run that group's attention on what the others wrote.  Producers and consumers share ONE L2, so the hand-off needs no
// agent-scope release / acquire (L2 write-back + invalidate: ~2 us round trip and ~11 us fences in rounds 2-4): stores are write-through to the
// XCD's L2, the arrival counter is a workgroup-scope atomic (performed AT that L2), and the consumers read counter and data with sc1 loads
// (miss the per-CU L1, hit the shared L2).  Gate: build the fused kernel only if the seam costs <= 1.5 us.
//
// Probe: 256 resident workgroups (one per CU); the 32 with blockIdx.x % 8 == c run on one XCD (checked with XCC_ID).  Per epoch every workgroup
// (1) waits a pseudo-random skew (0 .. ~1 us: the producers do not finish together), (2) writes 256 dwords, waits for the stores' acknowledgement
// and adds 1 to its XCD's counter, (3) polls the counter until all 32 have arrived, (4) reads the 32 x 256 dwords of the whole group and checks
// them.  s_memrealtime (100 MHz, chip-wide) stamps: arrival, go, data.  Reported per XCD and epoch: go - last arrival and data - last arrival
// (max over the group's workgroups), stale reads, and the same with agent-scope release / acquire for comparison (mode 1).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/seam_probe.hip -o tools/seam_probe && tools/seam_probe
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

constexpr int EPOCHS = 64, GROUP = 32, NT = 256;

struct Stamp {
    unsigned long long arrive, go, data;
    unsigned xcc, stale;
};

// mode 0: XCD-local (workgroup-scope atomic at the L2, sc1 loads, no fences); mode 1: agent-scope release / acquire (what a cross-XCD seam needs)
template <int MODE>
__global__ void __launch_bounds__(NT) k_seam(unsigned *counters, unsigned *data, Stamp *stamps, unsigned seed) {
    const int c = blockIdx.x & 7, j = blockIdx.x >> 3;  // XCD class, member
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned *my_data = data + ((size_t)c * GROUP + j) * NT;
    unsigned rng = seed ^ (blockIdx.x * 2654435761u);
    for (int e = 1; e <= EPOCHS; ++e) {
        rng = rng * 1664525u + 1013904223u;
        const int skew = (rng >> 24) & 15;  // x 64 cycles
        for (int s = 0; s < skew; ++s) __builtin_amdgcn_s_sleep(1);
        // (2) produce
        const unsigned val = (unsigned)e * 0x10000u + (unsigned)(blockIdx.x * NT + threadIdx.x);
        if (MODE == 0) {
            my_data[threadIdx.x] = val;  // write-through to the XCD's L2
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): acknowledged
            asm volatile("" ::: "memory");
        } else {
            __hip_atomic_store(my_data + threadIdx.x, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        unsigned long long t_arrive = 0, t_go = 0;
        if (threadIdx.x == 0) {
            unsigned old;
            if (MODE == 0) old = __hip_atomic_fetch_add(counters + c * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else old = __hip_atomic_fetch_add(counters + c * 64, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" : : "v"(old) : "memory");  // (the returned value: the add has been performed)
            t_arrive = __builtin_amdgcn_s_memrealtime();
            // (3) wait for the group
            const unsigned want = (unsigned)(GROUP * e);
            if (MODE == 0) {
                while (__hip_atomic_load(counters + c * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(0);
            } else {
                while (__hip_atomic_load(counters + c * 64, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(0);
            }
            t_go = __builtin_amdgcn_s_memrealtime();
        }
        __syncthreads();
        // (4) consume: the whole group's rows
        unsigned stale = 0;
        for (int p = 0; p < GROUP; ++p) {
            const unsigned *src = data + ((size_t)c * GROUP + p) * NT + threadIdx.x;
            const unsigned v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: past the L1, from the L2
            stale += v != (unsigned)e * 0x10000u + (unsigned)(((p << 3) | c) * NT + threadIdx.x);
        }
        stale = __syncthreads_count(stale != 0);
        if (threadIdx.x == 0) {
            Stamp &st = stamps[(size_t)(e - 1) * gridDim.x + blockIdx.x];
            st.arrive = t_arrive, st.go = t_go, st.data = __builtin_amdgcn_s_memrealtime(), st.xcc = xcc, st.stale = stale;
        }
        // nobody overwrites its rows before everyone has read them: a second counter closes the epoch
        if (threadIdx.x == 0) {
            if (MODE == 0) __hip_atomic_fetch_add(counters + c * 64 + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(counters + c * 64 + 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counters + c * 64 + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(GROUP * e)) __builtin_amdgcn_s_sleep(0);
        }
        __syncthreads();
    }
}

template <int MODE>
static void run(const char *name) {
    const int grid = 8 * GROUP;
    unsigned *counters, *data;
    Stamp *stamps;
    CK(hipMalloc((void **)&counters, 8 * 64 * 4)), CK(hipMalloc((void **)&data, (size_t)grid * NT * 4)), CK(hipMalloc((void **)&stamps, sizeof(Stamp) * EPOCHS * grid));
    std::vector<double> go, dat;
    long stale = 0, mixed = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counters, 0, 8 * 64 * 4)), CK(hipMemset(data, 0, (size_t)grid * NT * 4));
        hipLaunchKernelGGL(k_seam<MODE>, dim3(grid), dim3(NT), 0, 0, counters, data, stamps, 12345u + rep);
        CK(hipDeviceSynchronize());
        std::vector<Stamp> h((size_t)EPOCHS * grid);
        CK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        for (int e = 4; e < EPOCHS; ++e)  // (first epochs: cold)
            for (int c = 0; c < 8; ++c) {
                unsigned long long last = 0, g = 0, d = 0;
                unsigned x0 = h[(size_t)e * grid + c].xcc;
                for (int j = 0; j < GROUP; ++j) {
                    const Stamp &s = h[(size_t)e * grid + (j << 3 | c)];
                    last = std::max(last, s.arrive), g = std::max(g, s.go), d = std::max(d, s.data);
                    stale += s.stale, mixed += s.xcc != x0;
                }
                go.push_back((double)(g - last) * 0.01), dat.push_back((double)(d - last) * 0.01);
            }
    }
    std::sort(go.begin(), go.end()), std::sort(dat.begin(), dat.end());
    auto q = [](const std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
    std::printf("%-44s last arrival -> every workgroup released: median %.2f us, p10 %.2f, p90 %.2f;  -> all 32 KB of the group read: median %.2f us, p10 %.2f, p90 %.2f;"
                "  stale reads %ld, groups spanning XCDs %ld  (%zu group-epochs)\n",
                name, q(go, 0.5), q(go, 0.1), q(go, 0.9), q(dat, 0.5), q(dat, 0.1), q(dat, 0.9), stale, mixed, go.size());
    CK(hipFree(counters)), CK(hipFree(data)), CK(hipFree(stamps));
}

int main() {
    run<0>("XCD-local (L2 atomic, sc1 loads, no fences)");
    run<1>("agent-scope release / acquire");
    return 0;
}
