// Native thread stress of the KV page pool (csrc/page_pool.cpp), compiled by tests/test_page_pool.py with g++ (and
// with -fsanitize=thread).  The three scenarios are the reference's concurrency cases
// (tests/cpp/test_page_allocator.cpp:206-330), restated against the C ABI; rounds are repeated so the tagged free-list
// head sees many recycles of the same indices (the ABA pattern).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pie_hip.h"

namespace pie {
int fail(int code, const std::string &msg) {
    (void)msg;
    return code;
}
}  // namespace pie

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                  \
        }                                                                  \
    } while (0)

static pie_page_pool *make_pool(size_t pages) {
    pie_page_pool *p = nullptr;
    CHECK(pie_page_pool_create(pages, 4, 16, PIE_BF16, nullptr, &p) == PIE_OK);
    return p;
}

static void producers_consumer(size_t n_threads) {
    const size_t producers = n_threads - 1, per = 1024 / producers, total = per * producers;
    pie_page_pool *p = make_pool(total);
    std::vector<std::vector<uint32_t>> initial(producers);
    for (auto &v : initial)
        for (size_t i = 0; i < per; ++i) {
            uint32_t id;
            CHECK(pie_page_alloc(p, &id) == PIE_OK);
            v.push_back(id);
        }
    CHECK(pie_page_pool_num_free(p) == 0);
    std::atomic<bool> start{false};
    std::vector<uint32_t> got;
    std::vector<std::thread> ts;
    ts.emplace_back([&] {
        while (!start.load(std::memory_order_acquire)) std::this_thread::yield();
        for (size_t i = 0; i < total; ++i) {
            uint32_t id = 0;
            int rc = PIE_EXHAUSTED;
            for (size_t r = 0; r < 2000000 && (rc = pie_page_alloc(p, &id)) == PIE_EXHAUSTED; ++r) std::this_thread::yield();
            CHECK(rc == PIE_OK);
            got.push_back(id);
        }
    });
    for (size_t q = 0; q < producers; ++q)
        ts.emplace_back([&, q] {
            while (!start.load(std::memory_order_acquire)) std::this_thread::yield();
            for (uint32_t id : initial[q]) CHECK(pie_page_free(p, id) == PIE_OK);
        });
    start.store(true, std::memory_order_release);
    for (auto &t : ts) t.join();
    CHECK(got.size() == total);
    CHECK(std::set<uint32_t>(got.begin(), got.end()).size() == total);
    CHECK(pie_page_pool_num_free(p) == 0);
    pie_page_pool_destroy(p);
}

static void free_shared_page() {
    constexpr int refs = 10;
    pie_page_pool *p = make_pool(1);
    uint32_t id, n;
    CHECK(pie_page_alloc(p, &id) == PIE_OK);
    for (int i = 1; i < refs; ++i) CHECK(pie_page_add_ref(p, id) == PIE_OK);
    CHECK(pie_page_ref_count(p, id, &n) == PIE_OK && n == refs);
    std::atomic<bool> start{false};
    std::vector<std::thread> ts;
    for (int i = 0; i < refs; ++i)
        ts.emplace_back([&] {
            while (!start.load(std::memory_order_acquire)) std::this_thread::yield();
            CHECK(pie_page_free(p, id) == PIE_OK);
        });
    start.store(true, std::memory_order_release);
    for (auto &t : ts) t.join();
    CHECK(pie_page_pool_num_free(p) == 1);
    uint32_t again;
    CHECK(pie_page_alloc(p, &again) == PIE_OK && again == id);
    CHECK(pie_page_ref_count(p, id, &n) == PIE_OK && n == 1);
    pie_page_pool_destroy(p);
}

static void push_pop_stress(size_t n_threads, size_t ops) {
    constexpr size_t num_pages = 128, per_thread = 4;
    pie_page_pool *p = make_pool(num_pages);
    std::vector<std::vector<uint32_t>> local(n_threads);
    for (auto &v : local)
        for (size_t i = 0; i < per_thread; ++i) {
            uint32_t id;
            CHECK(pie_page_alloc(p, &id) == PIE_OK);
            v.push_back(id);
        }
    std::atomic<bool> start{false};
    std::vector<std::thread> ts;
    for (size_t tid = 0; tid < n_threads; ++tid)
        ts.emplace_back([&, tid] {
            while (!start.load(std::memory_order_acquire)) std::this_thread::yield();
            auto &owned = local[tid];
            for (size_t op = 0; op < ops && !owned.empty(); ++op) {
                const size_t idx = op % owned.size();
                CHECK(pie_page_free(p, owned[idx]) == PIE_OK);
                uint32_t id = 0;
                int rc = PIE_EXHAUSTED;
                for (int r = 0; r < 100 && (rc = pie_page_alloc(p, &id)) == PIE_EXHAUSTED; ++r) std::this_thread::yield();
                if (rc == PIE_OK) {
                    // a page handed out twice would show a reference count other than 1 here
                    uint32_t n = 0;
                    CHECK(pie_page_ref_count(p, id, &n) == PIE_OK && n == 1);
                    owned[idx] = id;
                } else {
                    owned.erase(owned.begin() + idx);
                }
            }
        });
    start.store(true, std::memory_order_release);
    for (auto &t : ts) t.join();
    std::set<uint32_t> held;
    for (auto &v : local)
        for (uint32_t id : v) {
            CHECK(held.insert(id).second);
            CHECK(pie_page_free(p, id) == PIE_OK);
        }
    CHECK(pie_page_pool_num_free(p) == num_pages);
    // every page must still be reachable exactly once
    std::set<uint32_t> all;
    uint32_t id;
    while (pie_page_alloc(p, &id) == PIE_OK) CHECK(all.insert(id).second);
    CHECK(all.size() == num_pages);
    pie_page_pool_destroy(p);
}

int main() {
    const size_t hw = std::max(4u, std::thread::hardware_concurrency());
    for (int round = 0; round < 5; ++round) {
        producers_consumer(hw);
        free_shared_page();
        push_pop_stress(hw, 20000);
    }
    std::puts("ok");
    return 0;
}
