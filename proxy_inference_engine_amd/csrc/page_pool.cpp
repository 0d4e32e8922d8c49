// page_pool.cpp -- native KV page pool: the re-design of pie_core's PageAllocator / KVPage
// (/root/reference/src/pie_core/include/engine/page_allocator.hpp:17-72, page.hpp:14-123, src/engine/page_allocator.cpp)
// for MI355X.
//
// Same contract -- pages of 64 token slots with [tokens, kv_heads, head_dim] key and value blocks, reference counts,
// a LIFO free list that starts as 0, 1, 2, ... (page_allocator.cpp:52-63), nullopt on exhaustion, out-of-range ids
// rejected -- pinned by the reference's own unit tests, restated in tests/test_page_pool.py case by case.
// Different construction: the pages are not 2 x N separately allocated arrays but slices of ONE caller-owned HBM slab
// (page p = bytes [p * page_bytes, (p + 1) * page_bytes): K block then V block), so a paged attention kernel addresses a
// page with one multiply and the pool itself never touches device memory (it also runs without a GPU); and the free
// list is an index stack whose head carries a modification tag, so the pop's compare-exchange cannot suffer the ABA
// problem the pointer-based Treiber stack of the reference is exposed to.
#include <atomic>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/pie_hip.h"

namespace pie {
int fail(int code, const std::string &msg);
}

namespace {
constexpr uint32_t NIL = 0xFFFFFFFFu;
struct PageMeta {
    std::atomic<uint32_t> ref_count{0};
    std::atomic<uint64_t> num_tokens{0};
    std::atomic<uint32_t> next{NIL};  // free-list link (valid while the page is free)
};
}  // namespace

struct pie_page_pool {
    size_t num_pages = 0;
    int num_heads = 0, head_dim = 0, elem_bytes = 2;
    size_t page_bytes = 0;
    char *slab = nullptr;  // caller-owned device memory, may be null (bookkeeping only)
    std::unique_ptr<PageMeta[]> pages;
    std::atomic<uint64_t> head{0};  // low 32 bits: index of the top free page (NIL = empty), high 32 bits: tag
    std::atomic<size_t> num_free{0};
};

static inline uint64_t pack(uint32_t idx, uint32_t tag) { return ((uint64_t)tag << 32) | idx; }

static void push_free(pie_page_pool *p, uint32_t id) {
    uint64_t old = p->head.load(std::memory_order_relaxed);
    do {
        p->pages[id].next.store((uint32_t)old, std::memory_order_relaxed);
    } while (!p->head.compare_exchange_weak(old, pack(id, (uint32_t)(old >> 32) + 1), std::memory_order_release, std::memory_order_relaxed));
    p->num_free.fetch_add(1, std::memory_order_relaxed);
}

static uint32_t pop_free(pie_page_pool *p) {
    uint64_t old = p->head.load(std::memory_order_acquire);
    for (;;) {
        const uint32_t idx = (uint32_t)old;
        if (idx == NIL) return NIL;
        const uint32_t next = p->pages[idx].next.load(std::memory_order_relaxed);
        if (p->head.compare_exchange_weak(old, pack(next, (uint32_t)(old >> 32) + 1), std::memory_order_acquire, std::memory_order_acquire)) {
            p->num_free.fetch_sub(1, std::memory_order_relaxed);
            return idx;
        }
    }
}

#define POOL_REQUIRE(cond, code, msg) \
    do {                              \
        if (!(cond)) return pie::fail((code), (msg)); \
    } while (0)

extern "C" {

// int8 pages (the reference's own storage, page.hpp:25-32): int8 K block, int8 V block, then the per-head fp16 scales of K and of V
// ([num_heads, 1] each); rounded up to 256 bytes so every page starts on a cache-line pair
size_t pie_page_i8_bytes(int num_kv_heads, int head_dim) {
    if (num_kv_heads <= 0 || head_dim <= 0) return 0;
    const size_t raw = 2 * (size_t)PIE_PAGE_TOKENS * num_kv_heads * head_dim + 4 * (size_t)num_kv_heads;
    return (raw + 255) & ~(size_t)255;
}

size_t pie_page_pool_slab_bytes(size_t num_pages, int num_kv_heads, int head_dim, int dtype) {
    if (num_pages == 0 || num_kv_heads <= 0 || head_dim <= 0 || (dtype != PIE_BF16 && dtype != PIE_F16 && dtype != PIE_I8)) return 0;
    if (dtype == PIE_I8) return num_pages * pie_page_i8_bytes(num_kv_heads, head_dim);
    return num_pages * 2 * (size_t)PIE_PAGE_TOKENS * num_kv_heads * head_dim * 2;
}

int pie_page_pool_create(size_t num_pages, int num_kv_heads, int head_dim, int dtype, void *slab, pie_page_pool **out) {
    POOL_REQUIRE(out, PIE_E_ARG, "pie_page_pool_create: null output pointer");
    POOL_REQUIRE(num_pages > 0, PIE_E_ARG, "PageAllocator must be initialized with num_pages > 0.");
    POOL_REQUIRE(num_pages < NIL, PIE_E_ARG, "pie_page_pool_create: too many pages");
    POOL_REQUIRE(num_kv_heads > 0, PIE_E_ARG, "num_heads must be positive.");
    POOL_REQUIRE(head_dim > 0, PIE_E_ARG, "head_dim must be positive.");
    POOL_REQUIRE(dtype == PIE_BF16 || dtype == PIE_F16 || dtype == PIE_I8, PIE_E_ARG, "pie_page_pool_create: dtype must be PIE_BF16, PIE_F16 or PIE_I8");
    pie_page_pool *p = new (std::nothrow) pie_page_pool();
    POOL_REQUIRE(p, PIE_E_HIP, "pie_page_pool_create: out of host memory");
    p->num_pages = num_pages, p->num_heads = num_kv_heads, p->head_dim = head_dim;
    p->elem_bytes = dtype == PIE_I8 ? 1 : 2;
    p->page_bytes = dtype == PIE_I8 ? pie_page_i8_bytes(num_kv_heads, head_dim) : 2 * (size_t)PIE_PAGE_TOKENS * num_kv_heads * head_dim * 2;
    p->slab = (char *)slab;
    p->pages.reset(new (std::nothrow) PageMeta[num_pages]);
    if (!p->pages) {
        delete p;
        return pie::fail(PIE_E_HIP, "pie_page_pool_create: out of host memory");
    }
    for (size_t i = 0; i < num_pages; ++i) p->pages[i].next.store(i + 1 < num_pages ? (uint32_t)(i + 1) : NIL, std::memory_order_relaxed);
    p->head.store(pack(0, 0), std::memory_order_relaxed);  // first allocation returns page 0, then 1, ... (page_allocator.cpp:52-63)
    p->num_free.store(num_pages, std::memory_order_release);
    *out = p;
    return PIE_OK;
}

int pie_page_pool_destroy(pie_page_pool *p) {
    delete p;
    return PIE_OK;
}

size_t pie_page_pool_size(const pie_page_pool *p) { return p ? p->num_pages : 0; }
size_t pie_page_pool_num_free(const pie_page_pool *p) { return p ? p->num_free.load(std::memory_order_acquire) : 0; }

static int check_id(const pie_page_pool *p, uint32_t id, const char *who) {
    POOL_REQUIRE(p, PIE_E_ARG, std::string(who) + ": null pool");
    POOL_REQUIRE(id < p->num_pages, PIE_E_RANGE,
                 "Page ID " + std::to_string(id) + " is out of range for pool size " + std::to_string(p->num_pages));
    return PIE_OK;
}

int pie_page_alloc(pie_page_pool *p, uint32_t *page_id) {
    POOL_REQUIRE(p && page_id, PIE_E_ARG, "pie_page_alloc: null pointer");
    const uint32_t id = pop_free(p);
    if (id == NIL) return PIE_EXHAUSTED;  // std::nullopt: not an error, no message
    p->pages[id].ref_count.store(1, std::memory_order_release);
    p->pages[id].num_tokens.store(0, std::memory_order_release);
    *page_id = id;
    return PIE_OK;
}

int pie_page_free(pie_page_pool *p, uint32_t id) {
    if (int rc = check_id(p, id, "pie_page_free")) return rc;
    // decrement unless already zero: the reference only asserts ("dec_ref on free page", page.hpp:88-91); a release build of it
    // would push the page on the free list twice -- here a double free is an error and the pool stays intact
    uint32_t c = p->pages[id].ref_count.load(std::memory_order_acquire);
    do {
        POOL_REQUIRE(c != 0, PIE_E_STATE, "pie_page_free: page " + std::to_string(id) + " is not allocated (double free)");
    } while (!p->pages[id].ref_count.compare_exchange_weak(c, c - 1, std::memory_order_acq_rel, std::memory_order_acquire));
    if (c == 1) push_free(p, id);
    return PIE_OK;
}

int pie_page_add_ref(pie_page_pool *p, uint32_t id) {
    if (int rc = check_id(p, id, "pie_page_add_ref")) return rc;
    uint32_t c = p->pages[id].ref_count.load(std::memory_order_acquire);
    do {  // "add_ref on free page" (page.hpp:79-82): refused, a free page may be handed to someone else at any moment
        POOL_REQUIRE(c != 0, PIE_E_STATE, "pie_page_add_ref: page " + std::to_string(id) + " is not allocated");
    } while (!p->pages[id].ref_count.compare_exchange_weak(c, c + 1, std::memory_order_acq_rel, std::memory_order_acquire));
    return PIE_OK;
}

int pie_page_ref_count(const pie_page_pool *p, uint32_t id, uint32_t *count) {
    if (int rc = check_id(p, id, "pie_page_ref_count")) return rc;
    POOL_REQUIRE(count, PIE_E_ARG, "pie_page_ref_count: null pointer");
    *count = p->pages[id].ref_count.load(std::memory_order_acquire);
    return PIE_OK;
}

int pie_page_num_tokens(const pie_page_pool *p, uint32_t id, size_t *n) {
    if (int rc = check_id(p, id, "pie_page_num_tokens")) return rc;
    POOL_REQUIRE(n, PIE_E_ARG, "pie_page_num_tokens: null pointer");
    *n = (size_t)p->pages[id].num_tokens.load(std::memory_order_acquire);
    return PIE_OK;
}

int pie_page_set_num_tokens(pie_page_pool *p, uint32_t id, size_t n) {
    if (int rc = check_id(p, id, "pie_page_set_num_tokens")) return rc;
    POOL_REQUIRE(n <= PIE_PAGE_TOKENS, PIE_E_ARG, "pie_page_set_num_tokens: a page holds at most 64 tokens");
    p->pages[id].num_tokens.store(n, std::memory_order_release);
    return PIE_OK;
}

int pie_page_ptrs(const pie_page_pool *p, uint32_t id, void **k, void **v) {
    if (int rc = check_id(p, id, "pie_page_ptrs")) return rc;
    POOL_REQUIRE(k && v, PIE_E_ARG, "pie_page_ptrs: null pointer");
    POOL_REQUIRE(p->slab, PIE_E_STATE, "pie_page_ptrs: the pool was created without a slab");
    *k = p->slab + (size_t)id * p->page_bytes;
    *v = p->slab + (size_t)id * p->page_bytes + (size_t)PIE_PAGE_TOKENS * p->num_heads * p->head_dim * p->elem_bytes;
    return PIE_OK;
}

int pie_page_scale_ptrs(const pie_page_pool *p, uint32_t id, void **k_scale, void **v_scale) {
    if (int rc = check_id(p, id, "pie_page_scale_ptrs")) return rc;
    POOL_REQUIRE(k_scale && v_scale, PIE_E_ARG, "pie_page_scale_ptrs: null pointer");
    POOL_REQUIRE(p->slab, PIE_E_STATE, "pie_page_scale_ptrs: the pool was created without a slab");
    POOL_REQUIRE(p->elem_bytes == 1, PIE_E_STATE, "pie_page_scale_ptrs: only int8 pages carry scales");
    char *s = p->slab + (size_t)id * p->page_bytes + 2 * (size_t)PIE_PAGE_TOKENS * p->num_heads * p->head_dim;
    *k_scale = s;
    *v_scale = s + 2 * (size_t)p->num_heads;
    return PIE_OK;
}

}  // extern "C"
