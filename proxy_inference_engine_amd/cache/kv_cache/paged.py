"""Paged KV storage: host mirror of pie_core's PageAllocator / KVPage over the native pool in libpie_hip.so
(reference: src/pie_core/include/engine/page_allocator.hpp:17-72, include/engine/page.hpp:14-123,
src/engine/page_allocator.cpp:8-157).

Method names, return values and error behaviour follow the reference class (std::optional -> None,
std::invalid_argument -> ValueError, std::out_of_range -> IndexError).  The pages are slices of one HBM slab
(`PageAllocator.slab`, uint8) owned by this object; `device=None` keeps bookkeeping only, so the allocator also runs
where no GPU is visible (the pool never touches device memory itself)."""
from __future__ import annotations

import ctypes as C

import torch

from ... import _ffi as _f

TOKEN_CAPACITY_PER_PAGE = 64  # page.hpp:14


class KVPage:
    """View of one page (page.hpp:18-123): id, reference count, token count and its K / V blocks."""

    __slots__ = ("_pool", "_id")

    def __init__(self, pool: "PageAllocator", page_id: int):
        self._pool, self._id = pool, page_id

    def page_id(self) -> int:
        return self._id

    def capacity(self) -> int:
        return TOKEN_CAPACITY_PER_PAGE

    def get_ref_count(self) -> int:
        n = C.c_uint32()
        _f.check(_f.load().pie_page_ref_count(self._pool._h, self._id, C.byref(n)))
        return n.value

    def num_tokens(self) -> int:
        n = C.c_size_t()
        _f.check(_f.load().pie_page_num_tokens(self._pool._h, self._id, C.byref(n)))
        return n.value

    def set_num_tokens(self, n: int) -> None:
        _f.check(_f.load().pie_page_set_num_tokens(self._pool._h, self._id, n))

    def _block(self, which: int, layer: int) -> torch.Tensor:
        p = self._pool
        if p.slab is None:
            raise RuntimeError("this PageAllocator was created without device storage (device=None)")
        half = p.page_bytes // 2
        raw = p.slab[layer, self._id * p.page_bytes + which * half: self._id * p.page_bytes + (which + 1) * half]
        # stored head-major [heads, 64, head_dim]; presented in the reference's logical order
        return raw.view(p.dtype).view(p.num_heads, TOKEN_CAPACITY_PER_PAGE, p.head_dim).permute(1, 0, 2)

    def key_cache(self, layer: int = 0) -> torch.Tensor:
        """[64, num_heads, head_dim] view into the slab (page.hpp:29)."""
        return self._block(0, layer)

    def value_cache(self, layer: int = 0) -> torch.Tensor:
        return self._block(1, layer)


class PageAllocator:
    """Fixed pool of KV pages with a lock-free LIFO free list (page_allocator.hpp:17-72)."""

    def __init__(self, num_pages: int, num_heads: int, head_dim: int, dtype: torch.dtype = torch.bfloat16,
                 device: torch.device | str | None = None, num_layers: int = 1):
        """num_layers > 1: one slab plane per decoder layer ([num_layers, slab bytes]); a page id names the same slice
        of every plane, so one block table serves all layers."""
        self._h = C.c_void_p()
        lib = _f.load()
        code = _f.dtype_code(dtype)
        if num_pages < 0 or num_heads < 0 or head_dim < 0:
            raise ValueError("PageAllocator: negative argument")
        self.slab = None
        nbytes = lib.pie_page_pool_slab_bytes(num_pages, num_heads, head_dim, code)
        if device is not None and nbytes:
            self.slab = torch.zeros(num_layers, nbytes, dtype=torch.uint8, device=device)  # mx::zeros pages (page.hpp:29-30)
        _f.check(lib.pie_page_pool_create(num_pages, num_heads, head_dim, code,
                                          C.c_void_p(self.slab.data_ptr() if self.slab is not None else None), C.byref(self._h)))
        self.num_heads, self.head_dim, self.dtype, self.num_layers = num_heads, head_dim, dtype, num_layers
        self.page_bytes = nbytes // num_pages

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _f.load().pie_page_pool_destroy(h)

    def allocate_page(self) -> int | None:
        """A free page id with ref count 1 and no tokens, or None when the pool is exhausted (page_allocator.cpp:68-79)."""
        out = C.c_uint32()
        rc = _f.load().pie_page_alloc(self._h, C.byref(out))
        if rc == 1:  # PIE_EXHAUSTED
            return None
        _f.check(rc)
        return out.value

    def free_page(self, page_id: int) -> None:
        """Drops one reference; the page returns to the free list at zero (page_allocator.cpp:81-87)."""
        _f.check(_f.load().pie_page_free(self._h, _id(page_id)))

    def add_ref(self, page_id: int) -> None:
        _f.check(_f.load().pie_page_add_ref(self._h, _id(page_id)))

    def get_page(self, page_id: int) -> KVPage:
        pid = _id(page_id)
        if pid >= self.size():
            raise IndexError(f"Page ID {page_id} is out of range for pool size {self.size()}")  # page_allocator.cpp:110-117
        return KVPage(self, pid)

    def size(self) -> int:
        return _f.load().pie_page_pool_size(self._h)

    def get_num_free_pages(self) -> int:
        return _f.load().pie_page_pool_num_free(self._h)


def _id(page_id: int) -> int:
    if page_id < 0 or page_id > 0xFFFFFFFF:
        raise IndexError(f"Page ID {page_id} is out of range")
    return int(page_id)
