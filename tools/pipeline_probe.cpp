// tools/pipeline_probe.cpp -- can a decode step's dependent launches overlap if they are dispatched WITHOUT the barrier bit and wait for
// their producers on device-side counters?  (developer tool)
//
// A step is a chain of 161 dependent stages shaped like the 8B model's: per layer {14.2, 0.6 (32 workgroups), 9.4, 66.1, 33.0} MB of
// input-independent "weights" streamed by 256 workgroups x 512 threads, every workgroup first reading the WHOLE 16-KB vector its
// producer's workgroups wrote (a GEMV's activation staging), + one 295.5-MB stage.  Written as raw AQL packets on one user queue:
//   serial     barrier bit + agent acquire / release fences on every packet (what HIP emits); plain loads and stores
//   pipelined  no barrier bit, no fences: stage i + 1 is dispatched while stage i runs, issues the head of its stream, then one thread
//              per workgroup polls stage i's completion counter; the vector moves with agent-scope (sc1) loads and stores
// Timed by the device timestamps of the first and last packet; the final vector is checked against a host replay (a stale read anywhere
// changes it).
//
// Build: g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ tools/pipeline_probe.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o tools/pipeline_probe
//        hipcc --offload-device-only --no-gpu-bundle-output --offload-arch=gfx950 -O3 tools/pipeline_probe_kernels.hip -o tools/pipeline_probe_kernels.hsaco
// Run:   tools/pipeline_probe tools/pipeline_probe_kernels.hsaco
#include <hip/hip_runtime_api.h>
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


struct StageArgs {  // tools/pipeline_probe_kernels.hip
    const void *w;
    unsigned long long n16;
    unsigned *counters;
    const unsigned *xin;
    unsigned *xout;
    unsigned *errors;
    unsigned idx, target, mode, salt, n_wg, per_wg;
    unsigned *flags;
    unsigned prod_wgs, epoch;
    const unsigned long long *xin64;
    unsigned long long *xout64;
};

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "tools/pipeline_probe_kernels.hsaco";
    const int layers = argc > 2 ? atoi(argv[2]) : 32;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    HK(hsa_init());
    HK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu || !g_have_cpu) return printf("no GPU / CPU agent\n"), 1;
    HK(hsa_amd_agent_iterate_memory_pools(g_gpu, dev_pool_cb, nullptr));
    HK(hsa_amd_agent_iterate_memory_pools(g_cpu, karg_pool_cb, nullptr));
    if (!g_have_dev || !g_have_karg) return printf("no device / kernarg pool\n"), 1;
    uint64_t freq = 0;
    hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq);
    std::ifstream f(path, std::ios::binary);
    if (!f) return printf("cannot open %s\n", path), 1;
    std::vector<char> co((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    hsa_code_object_reader_t reader;
    HK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &reader));
    hsa_executable_t exe;
    HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HK(hsa_executable_freeze(exe, nullptr));
    const Kernel k = get_kernel(exe, "k_stage");
    if (k.kernarg < sizeof(StageArgs)) return printf("kernarg size mismatch: %u < %zu\n", k.kernarg, sizeof(StageArgs)), 1;

    hsa_queue_t *q = nullptr;
    HK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    HK(hsa_amd_profiling_set_profiler_enabled(q, 1));

    // the stages of one step
    struct Stage {
        size_t bytes;
        unsigned wgs;
    };
    std::vector<Stage> stages;
    for (int l = 0; l < layers; ++l)
        for (Stage s : {Stage{14168064, 256}, Stage{544768, 32}, Stage{9437184, 256}, Stage{66068480, 256}, Stage{33030144, 256}}) stages.push_back(s);
    stages.push_back({295510016, 256});
    const int N = (int)stages.size();
    size_t total = 0;
    for (const Stage &s : stages) total += s.bytes;
    printf("%d stages, %.2f GB per step\n", N, total / 1e9);

    const size_t BIG = (size_t)5 << 30;  // every stage has its own bytes (no re-reads within a step: 4.2 GB pass through the 256 MiB Infinity Cache)
    char *big;  // from HIP's allocator, like the product's weights (the runtime's 2-MB fragments; a raw pool allocation streamed at 1 TB/s)
    if (hipMalloc((void **)&big, BIG) != hipSuccess || hipMemset(big, 1, BIG) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return printf("hipMalloc failed\n"), 1;
    const unsigned NW = 4096;
    // plain device memory (hipMalloc): a buffer the CPU may access is mapped uncached on the GPU, and 256 workgroups reading 16 KB of it cost 20 us
    unsigned *buf[2], *counters, *errors;
    for (int i = 0; i < 2; ++i)
        if (hipMalloc((void **)&buf[i], NW * 4) != hipSuccess) return printf("hipMalloc failed\n"), 1;
    if (hipMalloc((void **)&counters, (size_t)(N + 1) * 4) != hipSuccess || hipMalloc((void **)&errors, 64) != hipSuccess) return printf("hipMalloc failed\n"), 1;
    (void)hipMemset(counters, 0, (size_t)(N + 1) * 4);
    (void)hipMemset(errors, 0, 64);
    unsigned *flags;
    if (hipMalloc((void **)&flags, (size_t)(N + 1) * 256 * 4) != hipSuccess) return printf("hipMalloc failed\n"), 1;
    (void)hipMemset(flags, 0, (size_t)(N + 1) * 256 * 4);
    unsigned long long *buf64[2];
    for (int i = 0; i < 2; ++i)
        if (hipMalloc((void **)&buf64[i], NW * 8) != hipSuccess || hipMemset(buf64[i], 0, NW * 8) != hipSuccess) return printf("hipMalloc failed\n"), 1;
    (void)hipDeviceSynchronize();
    // kernel arguments in DEVICE memory (written through a staging copy): from the host kernarg pool every wave's scalar loads cross PCIe
    // and an EMPTY 256-workgroup kernel took 15-24 us
    char *kargs, *kargs_dev;
    const size_t karg_stride = 512;
    const bool dev_kargs = !(argc > 3 && atoi(argv[3]) == 0);
    HK(hsa_amd_memory_pool_allocate(g_karg_pool, karg_stride * (size_t)(N + 1), 0, (void **)&kargs));
    HK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kargs));
    if (hipMalloc((void **)&kargs_dev, karg_stride * (size_t)(N + 1)) != hipSuccess) return printf("hipMalloc failed\n"), 1;
    printf("kernel arguments in %s memory\n", dev_kargs ? "device" : "host (kernarg pool)");
    hsa_signal_t first_sig, last_sig;
    HK(hsa_signal_create(1, 0, nullptr, &first_sig));
    HK(hsa_signal_create(1, 0, nullptr, &last_sig));

    auto host_chain = [&](std::vector<unsigned> v) {
        std::vector<unsigned> o(NW);
        for (int l = 0; l < N; ++l) {
            unsigned tot = 0;
            const unsigned per = NW / stages[l].wgs;
            for (unsigned j = 0; j < NW; ++j) tot += v[j] * (2u * j + 1u);
            for (unsigned j = 0; j < NW; ++j) o[j] = tot * 1664525u + v[j] + (unsigned)l + (j % per);
            v.swap(o);
        }
        return v;
    };

    struct Case {
        const char *name;
        int barrier, fences;
        unsigned mode;
        const char *kernel = nullptr;  // another (empty) kernel instead of k_stage
    };
    const Case cases[] = {
        {"serial: barrier bit, agent fences, plain accesses            ", 1, 1, 0u},
        {"serial: barrier bit, agent fences, sc1 vector                ", 1, 1, 2u},
        {"serial: barrier bit, NO fences, sc1 vector                   ", 1, 0, 2u},
        {"pipelined: no barrier bit, counters, sc1 vector              ", 0, 0, 1u | 2u},
        {"pipelined: per-workgroup flags instead of the counter        ", 0, 0, 1u | 2u | 512u},
        {"pipelined: the vector's granules carry the epoch (no flag)   ", 0, 0, 1u | 2u | 1024u},
        {"pipelined: ... + release on the counter increment            ", 0, 0, 1u | 2u | 4u},
        {"pipelined: ... + release + acquire fence after the wait      ", 0, 0, 1u | 2u | 4u | 8u},
        {"pipelined: plain vector, release + acquire (the textbook form)", 0, 0, 1u | 4u | 8u},
        {"serial: barrier bit, agent fences, plain accesses (again)    ", 1, 1, 0u},
        {"serial, ablation: no vector read                             ", 1, 1, 16u},
        {"serial, ablation: no stream                                  ", 1, 1, 32u},
        {"serial, ablation: neither                                    ", 1, 1, 48u},
        {"serial, ablation: neither, no head loads                     ", 1, 1, 48u | 64u},
        {"serial, ablation: neither, no head loads, no publish         ", 1, 1, 48u | 64u | 128u},
        {"serial, ablation: ... and no barrier (an empty kernel)       ", 1, 1, 48u | 64u | 128u | 256u},
        {"empty kernel, 8-byte arguments, no LDS                       ", 1, 1, 16u | 128u, "k_empty_ptr"},
        {"empty kernel, 88-byte arguments, no LDS                      ", 1, 1, 16u | 128u, "k_empty_args88"},
        {"empty kernel, 88-byte arguments, 32 B of LDS                 ", 1, 1, 16u | 128u, "k_empty_args88_lds"},
        {"empty kernel, 280-byte arguments, no LDS                     ", 1, 1, 16u | 128u, "k_empty_args280"},
        {"empty kernel, 280-byte arguments, 40 KB of LDS               ", 1, 1, 16u | 128u, "k_empty_args280_lds40k"},
    };
    unsigned epoch = 0, flag_epoch = 0, tag_epoch = 1;  // completed repetitions with the counters / the flags in use (both are monotonic)
    for (const Case &c : cases) {
        const Kernel kc = c.kernel ? get_kernel(exe, c.kernel) : k;
        double best = 1e30, sum = 0.0, dev_best = 1e30, dev_sum = 0.0;
        int bad_runs = 0;
        const int REPS = 9;
        for (int rep = 0; rep < REPS; ++rep) {
            std::vector<unsigned> init(NW);
            for (unsigned j = 0; j < NW; ++j) init[j] = j * 2654435761u + (unsigned)rep * 977u + 12345u;
            (void)hipMemcpy(buf[0], init.data(), NW * 4, hipMemcpyHostToDevice);
            (void)hipMemset(buf[1], 0, NW * 4);
            if (c.mode & 1024u) {  // stage 0 reads granules tagged tag_epoch * 1024 + 0
                std::vector<unsigned long long> g(NW);
                for (unsigned j = 0; j < NW; ++j) g[j] = ((unsigned long long)(tag_epoch * 1024u) << 32) | init[j];
                (void)hipMemcpy(buf64[0], g.data(), NW * 8, hipMemcpyHostToDevice);
            }
            (void)hipDeviceSynchronize();
            hsa_signal_store_relaxed(first_sig, 1);
            hsa_signal_store_relaxed(last_sig, 1);
            const uint64_t base = hsa_queue_add_write_index_relaxed(q, (uint64_t)N);
            size_t off = 0;
            for (int i = 0; i < N; ++i) {
                StageArgs a;
                a.w = big + off, a.n16 = stages[i].bytes / 16, off += (stages[i].bytes + 4095) & ~(size_t)4095;
                a.counters = counters, a.xin = buf[i & 1], a.xout = buf[(i & 1) ^ 1], a.errors = errors;
                a.idx = (unsigned)i, a.target = i > 0 ? (epoch + 1) * stages[i - 1].wgs : 0, a.mode = c.mode, a.salt = (unsigned)i;
                a.n_wg = stages[i].wgs, a.per_wg = NW / stages[i].wgs;
                a.flags = flags, a.prod_wgs = i > 0 ? stages[i - 1].wgs : 0, a.epoch = flag_epoch + 1;
                a.xin64 = buf64[i & 1], a.xout64 = buf64[(i & 1) ^ 1];
                // tagged granules: stage i waits for ITS producer's tag and writes its own -- distinct per stage (the two buffers are reused every other stage) and per repetition
                if (c.mode & 1024u) a.epoch = tag_epoch * 1024u + (unsigned)i, a.target = tag_epoch * 1024u + (unsigned)i + 1u;
                memcpy(kargs + karg_stride * (size_t)i, &a, sizeof(a));
                hsa_kernel_dispatch_packet_t *p = (hsa_kernel_dispatch_packet_t *)q->base_address + ((base + (uint64_t)i) & (q->size - 1));
                p->workgroup_size_x = 512, p->workgroup_size_y = 1, p->workgroup_size_z = 1;
                p->grid_size_x = stages[i].wgs * 512u, p->grid_size_y = 1, p->grid_size_z = 1;
                p->private_segment_size = kc.priv, p->group_segment_size = kc.group;
                p->kernel_object = kc.object;
                p->kernarg_address = (dev_kargs ? kargs_dev : kargs) + karg_stride * (size_t)i;
                p->reserved2 = 0;
                p->completion_signal.handle = i == 0 ? first_sig.handle : (i == N - 1 ? last_sig.handle : 0);
                const int NO = HSA_FENCE_SCOPE_NONE, AG = HSA_FENCE_SCOPE_AGENT, SY = HSA_FENCE_SCOPE_SYSTEM;
                const int acq = i == 0 ? SY : (c.fences ? AG : NO), rel = i == N - 1 ? SY : (c.fences ? AG : NO);
                const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((c.barrier || i == 0 ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                                   (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
                const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
                __atomic_store_n((uint32_t *)p, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
            }
            if (dev_kargs) {
                (void)hipMemcpy(kargs_dev, kargs, karg_stride * (size_t)N, hipMemcpyHostToDevice);
                (void)hipDeviceSynchronize();
            }
            const auto t0 = std::chrono::steady_clock::now();
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(base + (uint64_t)N - 1));
            while (hsa_signal_wait_scacquire(last_sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {
            }
            const auto t1 = std::chrono::steady_clock::now();
            if (c.mode & 1024u) ++tag_epoch;
            else if ((c.mode & 513u) == 513u) ++flag_epoch;
            else if (c.mode & 1u) ++epoch;
            const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
            hsa_amd_profiling_dispatch_time_t tf, tl;
            HK(hsa_amd_profiling_get_dispatch_time(g_gpu, first_sig, &tf));
            HK(hsa_amd_profiling_get_dispatch_time(g_gpu, last_sig, &tl));
            const double dev_us = (double)(tl.end - tf.start) / (double)freq * 1e6;
            if (rep > 1) best = us < best ? us : best, sum += us, dev_best = dev_us < dev_best ? dev_us : dev_best, dev_sum += dev_us;
            const std::vector<unsigned> want = host_chain(init);
            std::vector<unsigned> got(NW);
            if (c.mode & 1024u) {
                std::vector<unsigned long long> g(NW);
                (void)hipMemcpy(g.data(), buf64[N & 1], NW * 8, hipMemcpyDeviceToHost);
                for (unsigned j = 0; j < NW; ++j) got[j] = (unsigned)g[j];
            } else
                (void)hipMemcpy(got.data(), buf[N & 1], NW * 4, hipMemcpyDeviceToHost);
            unsigned wrong = 0;
            for (unsigned j = 0; j < NW; ++j) wrong += got[j] != want[j];
            bad_runs += (wrong != 0 && !(c.mode & (16u | 128u)));
        }
        unsigned n_err = 0;
        (void)hipMemcpy(&n_err, errors, 4, hipMemcpyDeviceToHost);
        printf("%s %8.1f us per step (host clock, best; mean %8.1f) | device timestamps best %8.1f mean %8.1f | %.2f TB/s | wrong final vectors %d / %d | spin timeouts %u\n", c.name,
               best, sum / (REPS - 2), dev_best, dev_sum / (REPS - 2), total / dev_best / 1e6, bad_runs, REPS, n_err);
    }
    hsa_queue_destroy(q);
    hsa_shut_down();
    return 0;
}
