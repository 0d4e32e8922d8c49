"""The reference's PageAllocator unit tests (tests/cpp/test_page_allocator.cpp:60-330), restated case by case against
the native pool behind the C ABI (include/pie_hip.h, csrc/page_pool.cpp) through its Python mirror.  No GPU: the pool
is host bookkeeping over a caller-owned slab, created here without one.  ctypes releases the GIL inside every call, so
the threaded cases do race in the native code; tests/cpp/page_pool_stress.cpp repeats them with std::thread (and under
ThreadSanitizer)."""
import os
import subprocess
import threading
from pathlib import Path

import pytest

from proxy_inference_engine_amd.cache.kv_cache.paged import TOKEN_CAPACITY_PER_PAGE, PageAllocator

ROOT = Path(__file__).resolve().parents[1]
DEFAULT_NUM_HEADS, DEFAULT_HEAD_DIM = 4, 16                                   # test_page_allocator.cpp:20-21
LARGE_POOL_SIZE, SMALL_POOL_SIZE, TINY_POOL_SIZE, SINGLE_PAGE_POOL = 1024, 10, 4, 1   # :22-25


def make_allocator(pages):
    return PageAllocator(pages, DEFAULT_NUM_HEADS, DEFAULT_HEAD_DIM)


def allocate_pages(alloc, n):
    ids = []
    for i in range(n):
        pid = alloc.allocate_page()
        assert pid is not None, f"alloc failed @{i}"
        ids.append(pid)
    return ids


def free_pages(alloc, ids):
    for pid in ids:
        alloc.free_page(pid)


def test_constructor_valid_args():                                           # :60-62
    make_allocator(TINY_POOL_SIZE)


def test_constructor_invalid_args():                                         # :64-71
    with pytest.raises(ValueError, match="num_pages > 0"):
        PageAllocator(0, DEFAULT_NUM_HEADS, DEFAULT_HEAD_DIM)
    with pytest.raises(ValueError, match="num_heads must be positive"):
        PageAllocator(TINY_POOL_SIZE, 0, DEFAULT_HEAD_DIM)
    with pytest.raises(ValueError, match="head_dim must be positive"):
        PageAllocator(TINY_POOL_SIZE, DEFAULT_NUM_HEADS, 0)


def test_exhaust_and_refill():                                               # :76-91
    alloc = make_allocator(TINY_POOL_SIZE)
    assert alloc.size() == TINY_POOL_SIZE
    assert alloc.get_num_free_pages() == TINY_POOL_SIZE
    ids = allocate_pages(alloc, TINY_POOL_SIZE)
    assert alloc.get_num_free_pages() == 0
    assert alloc.allocate_page() is None
    free_pages(alloc, ids)
    assert alloc.get_num_free_pages() == TINY_POOL_SIZE
    assert alloc.allocate_page() is not None
    assert alloc.get_num_free_pages() == TINY_POOL_SIZE - 1


def test_exhaustion_returns_none():                                          # :93-99
    alloc = make_allocator(2)
    assert alloc.allocate_page() is not None
    assert alloc.allocate_page() is not None
    assert alloc.allocate_page() is None
    assert alloc.get_num_free_pages() == 0


def test_edge_case_single_page():                                            # :101-116
    alloc = make_allocator(SINGLE_PAGE_POOL)
    assert alloc.size() == SINGLE_PAGE_POOL
    pid = alloc.allocate_page()
    assert pid == 0
    assert alloc.allocate_page() is None
    alloc.free_page(pid)
    assert alloc.get_num_free_pages() == 1
    assert alloc.allocate_page() == 0


def test_fresh_pool_allocates_in_index_order():                              # page_allocator.cpp:52-63
    alloc = make_allocator(SMALL_POOL_SIZE)
    assert allocate_pages(alloc, SMALL_POOL_SIZE) == list(range(SMALL_POOL_SIZE))


def test_lifo_order():                                                       # :121-132
    alloc = make_allocator(SMALL_POOL_SIZE)
    first = allocate_pages(alloc, SMALL_POOL_SIZE)
    free_pages(alloc, first[::-1])
    assert alloc.get_num_free_pages() == SMALL_POOL_SIZE
    second = allocate_pages(alloc, SMALL_POOL_SIZE)
    assert second == first
    assert alloc.get_num_free_pages() == 0


def test_single_thread_ref_counting():                                       # :137-153
    alloc = make_allocator(SINGLE_PAGE_POOL)
    pid = alloc.allocate_page()
    assert alloc.get_page(pid).get_ref_count() == 1
    alloc.add_ref(pid)
    alloc.add_ref(pid)
    assert alloc.get_page(pid).get_ref_count() == 3
    alloc.free_page(pid)
    alloc.free_page(pid)
    assert alloc.get_page(pid).get_ref_count() == 1
    assert alloc.get_num_free_pages() == 0
    alloc.free_page(pid)
    assert alloc.get_num_free_pages() == 1


def test_explicit_add_ref():                                                 # :155-163
    alloc = make_allocator(SINGLE_PAGE_POOL)
    pid = alloc.allocate_page()
    alloc.add_ref(pid)
    assert alloc.get_page(pid).get_ref_count() == 2
    with pytest.raises(IndexError, match="out of range"):
        alloc.add_ref(999)


def test_get_page():                                                         # :168-190 (const and non-const accessors)
    alloc = make_allocator(2)
    id1, id2 = alloc.allocate_page(), alloc.allocate_page()
    assert alloc.get_page(id1).page_id() == id1
    assert alloc.get_page(id2).page_id() == id2
    assert alloc.get_page(id1).get_ref_count() == 1
    assert alloc.get_page(id1).capacity() == TOKEN_CAPACITY_PER_PAGE == 64
    with pytest.raises(IndexError):
        alloc.get_page(999)


def test_num_tokens_reset_on_allocation():                                   # page.hpp:69,100-103; page_allocator.cpp:74-76
    alloc = make_allocator(SINGLE_PAGE_POOL)
    pid = alloc.allocate_page()
    page = alloc.get_page(pid)
    assert page.num_tokens() == 0
    page.set_num_tokens(37)
    assert page.num_tokens() == 37
    with pytest.raises(ValueError):
        page.set_num_tokens(65)
    alloc.free_page(pid)
    assert alloc.allocate_page() == pid
    assert alloc.get_page(pid).num_tokens() == 0


def test_invalid_id_throws():                                                # :195-201
    alloc = make_allocator(5)
    for bad in (5, 100):
        with pytest.raises(IndexError):
            alloc.get_page(bad)
    with pytest.raises(IndexError):
        alloc.free_page(5)
    with pytest.raises(IndexError):
        alloc.add_ref(5)


def test_double_free_and_add_ref_on_free_page_are_errors():
    """The reference asserts on both (page.hpp:79-91); here they are reported and leave the pool consistent."""
    alloc = make_allocator(TINY_POOL_SIZE)
    pid = alloc.allocate_page()
    alloc.free_page(pid)
    with pytest.raises(RuntimeError, match="double free"):
        alloc.free_page(pid)
    with pytest.raises(RuntimeError, match="not allocated"):
        alloc.add_ref(pid)
    assert alloc.get_num_free_pages() == TINY_POOL_SIZE
    assert sorted(allocate_pages(alloc, TINY_POOL_SIZE)) == list(range(TINY_POOL_SIZE))     # every page exactly once


def test_key_cache_needs_device_storage():
    alloc = make_allocator(2)
    with pytest.raises(RuntimeError, match="without device storage"):
        alloc.get_page(alloc.allocate_page()).key_cache()


def _run(threads):
    gate = threading.Event()
    ts = [threading.Thread(target=lambda f=f: (gate.wait(), f())) for f in threads]
    for t in ts:
        t.start()
    gate.set()
    for t in ts:
        t.join()


def test_concurrent_alloc_free_producers_consumer():                         # :206-254
    num_threads = max(2, os.cpu_count() or 2)
    num_producers = num_threads - 1
    pages_per_p = LARGE_POOL_SIZE // num_producers
    total = pages_per_p * num_producers
    alloc = make_allocator(total)
    initial = [allocate_pages(alloc, pages_per_p) for _ in range(num_producers)]
    assert alloc.get_num_free_pages() == 0
    got, failed = [], []

    def consumer():
        import time
        for i in range(total):
            pid = None
            deadline = time.monotonic() + 120.0           # bounded by time, not by spins: producers may be descheduled
            while time.monotonic() < deadline:
                pid = alloc.allocate_page()
                if pid is not None:
                    break
                time.sleep(0)
            if pid is None:
                failed.append(i)
                return
            got.append(pid)

    _run([consumer] + [lambda p=p: free_pages(alloc, initial[p]) for p in range(num_producers)])
    assert not failed, f"consumer failed @{failed}"
    assert len(got) == total and len(set(got)) == total
    assert alloc.get_num_free_pages() == 0


def test_concurrent_free_shared_page():                                      # :256-279
    refs = 10
    alloc = make_allocator(SINGLE_PAGE_POOL)
    pid = alloc.allocate_page()
    for _ in range(1, refs):
        alloc.add_ref(pid)
    assert alloc.get_page(pid).get_ref_count() == refs
    _run([lambda: alloc.free_page(pid)] * refs)
    assert alloc.get_num_free_pages() == 1
    assert alloc.allocate_page() == pid
    assert alloc.get_page(pid).get_ref_count() == 1


def test_high_contention_push_pop_stress():                                  # :282-330
    num_pages, ops_per_thr, pages_per_thread = 128, 2000, 4
    num_threads = max(4, os.cpu_count() or 4)
    alloc = make_allocator(num_pages)
    local = [allocate_pages(alloc, pages_per_thread) for _ in range(num_threads)]

    def worker(tid):
        owned = local[tid]
        for op in range(ops_per_thr):
            if not owned:
                break
            idx = op % len(owned)
            alloc.free_page(owned[idx])
            np_ = None
            for _ in range(100):
                np_ = alloc.allocate_page()
                if np_ is not None:
                    break
            if np_ is not None:
                owned[idx] = np_
            else:
                del owned[idx]

    _run([lambda t=t: worker(t) for t in range(num_threads)])
    for vec in local:
        for pid in vec:
            if alloc.get_page(pid).get_ref_count() == 1:
                alloc.free_page(pid)
    assert alloc.get_num_free_pages() == num_pages, "leak detected"


@pytest.mark.parametrize("sanitize", [False, True], ids=["native", "tsan"])
def test_native_thread_stress(tmp_path, sanitize):
    """std::thread versions of the three concurrency cases, compiled from csrc/page_pool.cpp itself; once plainly and
    once under ThreadSanitizer (CPU build: sanitizers are not available on the GPU pool)."""
    exe = tmp_path / "page_pool_stress"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", str(ROOT / "tests/cpp/page_pool_stress.cpp"),
           str(ROOT / "proxy_inference_engine_amd/csrc/page_pool.cpp"), "-o", str(exe)]
    if sanitize:
        cmd.insert(1, "-fsanitize=thread")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and sanitize and "tsan" in r.stderr.lower():
        pytest.skip("libtsan not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    if sanitize and r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ok" in r.stdout


def test_int8_page_geometry_and_oracle_quantiser():
    """The reference page's own storage (page.hpp:25-32): int8 K / V blocks [64, heads, head_dim] + float16 scales [heads, 1] per block.
    Geometry of the native pool for that dtype (no GPU needed) and the oracle's definition of the arithmetic the HIP kernels follow."""
    import numpy as np
    import torch
    from oracle import pie_oracle as po
    from proxy_inference_engine_amd import _ffi
    lib = _ffi.load()
    for heads, dim in ((8, 128), (2, 64), (1, 128)):
        pb = lib.pie_page_i8_bytes(heads, dim)
        assert pb % 256 == 0 and 0 <= pb - (2 * 64 * heads * dim + 4 * heads) < 256
        assert lib.pie_page_pool_slab_bytes(5, heads, dim, _ffi.PIE_I8) == 5 * pb
    pool = PageAllocator(4, 2, 64, dtype=torch.int8)          # bookkeeping only: the same allocator contract for int8 pages
    assert pool.page_bytes == lib.pie_page_i8_bytes(2, 64) and pool.allocate_page() == 0 and pool.get_num_free_pages() == 3
    x = np.array([[-300.0, -2.5, -1.5, -0.5, 0.49, 0.5, 1.5, 2.5, 126.5, 127.5]], np.float32)
    assert po.kv_i8_quantize(x, np.ones(1, np.float16)).tolist() == [[-127, -2, -2, 0, 0, 0, 2, 2, 126, 127]]       # half-even, clamp
    q = po.kv_i8_quantize(x, np.array([0.5], np.float16))
    assert q.tolist() == [[-127, -5, -3, -1, 1, 1, 3, 5, 127, 127]]
    assert np.array_equal(po.kv_i8_dequantize(q, np.array([0.5], np.float16)), q.astype(np.float32) * 0.5)
    rng = np.random.default_rng(0)
    y = rng.standard_normal((7, 3, 16)).astype(np.float32)
    s = np.full((7, 3), 1 / 32, np.float16)
    err = np.abs(po.kv_i8_dequantize(po.kv_i8_quantize(y, s), s) - y)
    assert err[np.abs(y) <= 127 / 32].max() <= 0.5 / 32 + 1e-7                                                          # half a step inside the range
