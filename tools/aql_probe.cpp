// tools/aql_probe.cpp -- what does a dependent kernel boundary cost as a function of the AQL packet's fence scopes?  (developer tool)
//
// HIP decides the acquire / release fence scopes of every dispatch packet; this probe writes the packets itself (HSA runtime, one user
// queue, one doorbell per chain) so the scopes can be chosen: a chain of N dependent dispatches (barrier bit set) of
//   * a trivial kernel (256 workgroups x 512 threads), and
//   * a "link" kernel in which every workgroup reads the whole 16-KB vector the previous link's 256 workgroups wrote (the shape of a
//     decode GEMV's activation staging), with plain or sc1 (agent-scope, write-through / L1-bypassing) accesses,
// timed end to end and checked against a host replay of the chain: a stale read anywhere changes the final vector.
//
// Build: g++ -O2 -std=c++17 tools/aql_probe.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o tools/aql_probe
// Run:   tools/aql_probe tools/aql_probe_kernels.hsaco [chain length]
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#define HK(x)                                                                  \
    do {                                                                       \
        hsa_status_t s_ = (x);                                                 \
        if (s_ != HSA_STATUS_SUCCESS) {                                        \
            const char *m_ = nullptr;                                          \
            hsa_status_string(s_, &m_);                                        \
            printf("HSA error %d (%s) at line %d: %s\n", (int)s_, m_ ? m_ : "?", __LINE__, #x); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_amd_memory_pool_t g_dev_pool, g_karg_pool;
static bool g_have_dev = false, g_have_karg = false;

static hsa_status_t agent_cb(hsa_agent_t a, void *) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) g_gpu = a, g_have_gpu = true;
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) g_cpu = a, g_have_cpu = true;
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t dev_pool_cb(hsa_amd_memory_pool_t p, void *) {
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev) g_dev_pool = p, g_have_dev = true;
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t karg_pool_cb(hsa_amd_memory_pool_t p, void *) {
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_karg) g_karg_pool = p, g_have_karg = true;
    return HSA_STATUS_SUCCESS;
}

struct Kernel {
    uint64_t object = 0;
    uint32_t kernarg = 0, group = 0, priv = 0;
};
static Kernel get_kernel(hsa_executable_t exe, const char *name) {
    hsa_executable_symbol_t sym;
    HK(hsa_executable_get_symbol_by_name(exe, (std::string(name) + ".kd").c_str(), &g_gpu, &sym));
    Kernel k;
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "tools/aql_probe_kernels.hsaco";
    const int N = argc > 2 ? atoi(argv[2]) : 1000;
    HK(hsa_init());
    HK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu || !g_have_cpu) return printf("no GPU / CPU agent\n"), 1;
    HK(hsa_amd_agent_iterate_memory_pools(g_gpu, dev_pool_cb, nullptr));
    HK(hsa_amd_agent_iterate_memory_pools(g_cpu, karg_pool_cb, nullptr));
    if (!g_have_dev || !g_have_karg) return printf("no device / kernarg pool\n"), 1;
    char name[64] = {};
    hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name);
    uint64_t freq = 0;
    hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq);
    printf("agent %s, timestamp frequency %.1f MHz, chain length %d\n", name, freq / 1e6, N);

    std::ifstream f(path, std::ios::binary);
    if (!f) return printf("cannot open %s\n", path), 1;
    std::vector<char> co((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    hsa_code_object_reader_t reader;
    HK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &reader));
    hsa_executable_t exe;
    HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HK(hsa_executable_freeze(exe, nullptr));
    const Kernel k_triv = get_kernel(exe, "k_trivial"), k_plain = get_kernel(exe, "k_link_plain"), k_sc1 = get_kernel(exe, "k_link_sc1");
    printf("kernarg bytes: trivial %u, link %u; group bytes: link %u\n", k_triv.kernarg, k_plain.kernarg, k_plain.group);

    hsa_queue_t *q = nullptr;
    HK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    HK(hsa_amd_profiling_set_profiler_enabled(q, 1));

    const unsigned NW = 4096;
    unsigned *buf[2];
    for (int i = 0; i < 2; ++i) {
        HK(hsa_amd_memory_pool_allocate(g_dev_pool, NW * 4, 0, (void **)&buf[i]));
        HK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, buf[i]));  // large-BAR: the host initialises and reads the vectors directly
    }
    char *kargs;
    const size_t karg_stride = 256;
    HK(hsa_amd_memory_pool_allocate(g_karg_pool, karg_stride * (size_t)(N + 1), 0, (void **)&kargs));
    HK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kargs));
    memset(kargs, 0, karg_stride * (size_t)(N + 1));
    hsa_signal_t first_sig, last_sig;
    HK(hsa_signal_create(1, 0, nullptr, &first_sig));
    HK(hsa_signal_create(1, 0, nullptr, &last_sig));

    auto host_chain = [&](std::vector<unsigned> v, int n) {
        std::vector<unsigned> o(NW);
        for (int l = 0; l < n; ++l) {
            unsigned tot = 0;
            for (unsigned j = 0; j < NW; ++j) tot += v[j] * (2u * j + 1u);
            for (unsigned j = 0; j < NW; ++j) o[j] = tot * 1664525u + v[j] + (unsigned)l + (j & 15u);
            v.swap(o);
        }
        return v;
    };

    struct Case {
        const char *name;
        int kind;  // 0 trivial, 1 link plain, 2 link sc1
        int acq, rel, barrier;
        int wgs = 256, threads = 512;  // launch geometry (trivial kernel only)
    };
    const int NO = HSA_FENCE_SCOPE_NONE, AG = HSA_FENCE_SCOPE_AGENT, SY = HSA_FENCE_SCOPE_SYSTEM;
    const Case cases[] = {
        {"trivial, acquire agent  / release agent ", 0, AG, AG, 1}, {"trivial, acquire none   / release none  ", 0, NO, NO, 1}, {"trivial, acquire agent  / release none  ", 0, AG, NO, 1},
        {"trivial, acquire none   / release agent ", 0, NO, AG, 1}, {"trivial, acquire system / release system", 0, SY, SY, 1}, {"trivial, no barrier bit, no fences      ", 0, NO, NO, 0},
        {"link plain, acquire agent / release agent", 1, AG, AG, 1}, {"link plain, acquire none  / release none ", 1, NO, NO, 1}, {"link plain, acquire agent / release none ", 1, AG, NO, 1},
        {"link plain, acquire none  / release agent", 1, NO, AG, 1}, {"link sc1,   acquire agent / release agent", 2, AG, AG, 1}, {"link sc1,   acquire none  / release none ", 2, NO, NO, 1},
        {"link sc1,   acquire agent / release none ", 2, AG, NO, 1}, {"link sc1,   acquire none  / release agent", 2, NO, AG, 1},
        // what a dispatch costs by its size (agent fences, barrier bit): is the floor per packet or per wave?
        {"trivial, 256 wgs x 512 threads (2048 waves)", 0, AG, AG, 1, 256, 512}, {"trivial, 256 wgs x 256 threads (1024 waves)", 0, AG, AG, 1, 256, 256},
        {"trivial, 256 wgs x 128 threads ( 512 waves)", 0, AG, AG, 1, 256, 128}, {"trivial, 256 wgs x  64 threads ( 256 waves)", 0, AG, AG, 1, 256, 64},
        {"trivial, 128 wgs x 512 threads (1024 waves)", 0, AG, AG, 1, 128, 512}, {"trivial,  32 wgs x 256 threads ( 128 waves)", 0, AG, AG, 1, 32, 256},
        {"trivial,   1 wg  x  64 threads (   1 wave )", 0, AG, AG, 1, 1, 64}, {"trivial, 512 wgs x 512 threads (4096 waves)", 0, AG, AG, 1, 512, 512},
        {"trivial, 1024 wgs x 256 threads (4096 waves)", 0, AG, AG, 1, 1024, 256},
    };
    for (const Case &c : cases) {
        double best = 1e30, sum = 0.0, dev_best = 1e30;
        int bad_runs = 0;
        const int REPS = 7;
        for (int rep = 0; rep < REPS; ++rep) {
            std::vector<unsigned> init(NW);
            for (unsigned j = 0; j < NW; ++j) init[j] = j * 2654435761u + (unsigned)rep * 977u + 12345u;
            memcpy(buf[0], init.data(), NW * 4);
            memset(buf[1], 0, NW * 4);
            hsa_signal_store_relaxed(first_sig, 1);
            hsa_signal_store_relaxed(last_sig, 1);
            const Kernel &k = c.kind == 0 ? k_triv : (c.kind == 1 ? k_plain : k_sc1);
            const uint64_t base = hsa_queue_add_write_index_relaxed(q, (uint64_t)N);
            // the queue (4096 packets) is empty here: every chain is waited for before the next starts
            for (int i = 0; i < N; ++i) {
                char *ka = kargs + karg_stride * (size_t)i;
                if (c.kind == 0) {
                    void *p = nullptr;
                    memcpy(ka, &p, 8);
                } else {
                    const unsigned *in = buf[i & 1];
                    unsigned *out = buf[(i & 1) ^ 1];
                    const unsigned salt = (unsigned)i;
                    memcpy(ka, &in, 8), memcpy(ka + 8, &out, 8), memcpy(ka + 16, &salt, 4);
                }
                hsa_kernel_dispatch_packet_t *p = (hsa_kernel_dispatch_packet_t *)q->base_address + ((base + (uint64_t)i) & (q->size - 1));
                const int wgs = c.kind == 0 ? c.wgs : 256, thr = c.kind == 0 ? c.threads : 512;
                p->workgroup_size_x = (uint16_t)thr, p->workgroup_size_y = 1, p->workgroup_size_z = 1;
                p->grid_size_x = (uint32_t)(wgs * thr), p->grid_size_y = 1, p->grid_size_z = 1;
                p->private_segment_size = k.priv, p->group_segment_size = k.group;
                p->kernel_object = k.object;
                p->kernarg_address = ka;
                p->reserved2 = 0;
                p->completion_signal.handle = i == 0 ? first_sig.handle : (i == N - 1 ? last_sig.handle : 0);
                // the chain's ends are visible to the host: system scope on the first acquire and the last release
                const int acq = i == 0 ? SY : c.acq, rel = i == N - 1 ? SY : c.rel;
                const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((c.barrier || i == 0 || i == N - 1 ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                                   (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
                const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
                __atomic_store_n((uint32_t *)p, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
            }
            const auto t0 = std::chrono::steady_clock::now();
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(base + (uint64_t)N - 1));
            while (hsa_signal_wait_scacquire(last_sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {
            }
            const auto t1 = std::chrono::steady_clock::now();
            const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
            hsa_amd_profiling_dispatch_time_t tf, tl;
            HK(hsa_amd_profiling_get_dispatch_time(g_gpu, first_sig, &tf));
            HK(hsa_amd_profiling_get_dispatch_time(g_gpu, last_sig, &tl));
            const double dev_us = (double)(tl.end - tf.start) / (double)freq * 1e6;
            if (rep > 0) best = us < best ? us : best, sum += us, dev_best = dev_us < dev_best ? dev_us : dev_best;
            if (c.kind != 0) {
                const std::vector<unsigned> want = host_chain(init, N);
                const unsigned *got = buf[N & 1];
                unsigned wrong = 0;
                for (unsigned j = 0; j < NW; ++j) wrong += got[j] != want[j];
                bad_runs += wrong != 0;
            }
        }
        printf("%-44s %7.3f us per dispatch (host clock, best; mean %7.3f) | device timestamps %7.3f", c.name, best / N, sum / (REPS - 1) / N, dev_best / N);
        if (c.kind != 0) printf(" | chains with a wrong final vector: %d / %d", bad_runs, REPS);
        printf("\n");
    }
    hsa_queue_destroy(q);
    hsa_shut_down();
    return 0;
}
