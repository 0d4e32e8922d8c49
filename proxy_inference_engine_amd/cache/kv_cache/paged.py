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
from . import BaseCache

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
        half = TOKEN_CAPACITY_PER_PAGE * p.num_heads * p.head_dim * p.dtype.itemsize  # (an int8 page carries its scales after the two blocks)
        raw = p.slab[layer, self._id * p.page_bytes + which * half: self._id * p.page_bytes + (which + 1) * half]
        # stored head-major [heads, 64, head_dim]; presented in the reference's logical order
        return raw.view(p.dtype).view(p.num_heads, TOKEN_CAPACITY_PER_PAGE, p.head_dim).permute(1, 0, 2)

    def key_cache(self, layer: int = 0) -> torch.Tensor:
        """[64, num_heads, head_dim] view into the slab (page.hpp:29)."""
        return self._block(0, layer)

    def value_cache(self, layer: int = 0) -> torch.Tensor:
        return self._block(1, layer)

    def _scale(self, which: int, layer: int) -> torch.Tensor:
        p = self._pool
        if p.dtype != torch.int8:
            raise RuntimeError("only int8 pages carry scales (page.hpp:25-32)")
        if p.slab is None:
            raise RuntimeError("this PageAllocator was created without device storage (device=None)")
        off = self._id * p.page_bytes + 2 * TOKEN_CAPACITY_PER_PAGE * p.num_heads * p.head_dim + which * 2 * p.num_heads
        return p.slab[layer, off: off + 2 * p.num_heads].view(torch.float16).view(p.num_heads, 1)

    def key_cache_scale(self, layer: int = 0) -> torch.Tensor:
        """[num_heads, 1] float16 view into the slab (page.hpp:31,74); ones until set."""
        return self._scale(0, layer)

    def value_cache_scale(self, layer: int = 0) -> torch.Tensor:
        return self._scale(1, layer)


class PageAllocator:
    """Fixed pool of KV pages with a lock-free LIFO free list (page_allocator.hpp:17-72)."""

    def __init__(self, num_pages: int, num_heads: int, head_dim: int, dtype: torch.dtype = torch.bfloat16,
                 device: torch.device | str | None = None, num_layers: int = 1):
        """num_layers > 1: one slab plane per decoder layer ([num_layers, slab bytes]); a page id names the same slice
        of every plane, so one block table serves all layers."""
        self._h = C.c_void_p()
        lib = self._lib = _f.load()
        code = _f.PIE_I8 if dtype == torch.int8 else _f.dtype_code(dtype)  # int8: the reference page's own storage (page.hpp:25-32)
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
        if dtype == torch.int8 and self.slab is not None:  # key_cache_scale_ / value_cache_scale_ = mx::ones (page.hpp:31-32)
            for layer in range(num_layers):
                _f.check(lib.pie_page_i8_set_scales(C.c_void_p(self.slab[layer].data_ptr()), num_pages, num_heads, head_dim, None, num_pages, None, None,
                                                    _f.stream()))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        lib = getattr(self, "_lib", None)   # kept on the object: module globals may already be gone at interpreter shutdown
        if h and lib is not None:
            try:
                lib.pie_page_pool_destroy(h)
            except Exception:  # noqa: BLE001
                pass

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


class PagedSequence:
    """One sequence's view of the page pool: its block table (logical block j -> page id), kept on the host and mirrored
    in a device int32 tensor the kernels read (BatchDetails.consolidated_block_table, batch_details.hpp:52-66), and its
    length.  Shared by the per-layer PagedKVCache objects of a model: a page id names the same slice of every layer's
    slab plane.  Growing takes pages from the allocator -- nothing is copied, unlike ReusableKVCache's reallocation."""

    def __init__(self, allocator: PageAllocator, max_blocks: int = 16):
        if allocator.slab is None:
            raise ValueError("PagedSequence needs a PageAllocator with device storage")
        self.allocator = allocator
        self.pages: list[int] = []
        self.offset = 0
        self.table = torch.zeros(max(1, max_blocks), dtype=torch.int32, device=allocator.slab.device)

    @property
    def max_blocks(self) -> int:
        return self.table.numel()

    @property
    def capacity(self) -> int:
        return self.max_blocks * TOKEN_CAPACITY_PER_PAGE

    def reserve(self, needed: int) -> None:
        """Pages (and table room) for `needed` more positions; raises RuntimeError when the pool is exhausted."""
        blocks = (self.offset + needed + TOKEN_CAPACITY_PER_PAGE - 1) // TOKEN_CAPACITY_PER_PAGE
        if blocks > self.max_blocks:  # a longer table is a new device buffer: the decoder re-plans its attention launch
            table = torch.zeros(max(blocks, 2 * self.max_blocks), dtype=torch.int32, device=self.table.device)
            table[:self.max_blocks] = self.table
            self.table = table
        first = len(self.pages)
        while len(self.pages) < blocks:
            pid = self.allocator.allocate_page()
            if pid is None:
                for p in self.pages[first:]:
                    self.allocator.free_page(p)
                del self.pages[first:]
                raise RuntimeError(f"KV page pool exhausted ({self.allocator.size()} pages of {TOKEN_CAPACITY_PER_PAGE} tokens)")
            self.pages.append(pid)
        if len(self.pages) > first:
            self.table[first:len(self.pages)] = torch.tensor(self.pages[first:], dtype=torch.int32)
            self._mark_tokens()

    def advance(self, n: int) -> None:
        self.offset += n
        self._mark_tokens()

    def _mark_tokens(self) -> None:
        for j, pid in enumerate(self.pages):  # KVPage.num_tokens (page.hpp:69,100-103)
            self.allocator.get_page(pid).set_num_tokens(max(0, min(TOKEN_CAPACITY_PER_PAGE, self.offset - j * TOKEN_CAPACITY_PER_PAGE)))

    def truncate(self, length: int) -> None:
        """Keeps the first `length` positions and returns the pages behind them to the pool."""
        length = max(0, min(length, self.offset))
        keep = (length + TOKEN_CAPACITY_PER_PAGE - 1) // TOKEN_CAPACITY_PER_PAGE
        for pid in self.pages[keep:]:
            self.allocator.free_page(pid)
        del self.pages[keep:]
        self.offset = length
        self._mark_tokens()

    def release(self) -> None:
        self.truncate(0)

    def fork(self) -> "PagedSequence":
        """A second sequence with the same history: full pages are shared by reference count (add_ref,
        page_allocator.cpp:89-92), the partly filled last page is copied so both can keep appending."""
        other = PagedSequence(self.allocator, self.max_blocks)
        full = self.offset // TOKEN_CAPACITY_PER_PAGE
        for pid in self.pages[:full]:
            self.allocator.add_ref(pid)
            other.pages.append(pid)
        if self.offset % TOKEN_CAPACITY_PER_PAGE:
            pid = self.allocator.allocate_page()
            if pid is None:
                other.release()
                raise RuntimeError("KV page pool exhausted")
            pb, src = self.allocator.page_bytes, self.pages[full]
            self.allocator.slab[:, pid * pb:(pid + 1) * pb] = self.allocator.slab[:, src * pb:(src + 1) * pb]
            other.pages.append(pid)
        other.offset = self.offset
        if other.pages:
            other.table[:len(other.pages)] = torch.tensor(other.pages, dtype=torch.int32)
        other._mark_tokens()
        return other

    def __del__(self):
        try:
            if getattr(self, "allocator", None) is not None and getattr(self.allocator, "_h", None):
                self.release()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class PagedKVCache(BaseCache):
    """Per-layer cache object over a PagedSequence: what the reference's stub stands for ("a key-value cache that uses
    blocks of memory ... and a page manager", cache/kv_cache/paged.py:6-14).  `page_manager` is the shared PagedSequence;
    the decode kernels append and read the rows themselves through the block table, this class carries the protocol the
    engine and PromptCache use (offset, reuse, trim, state for persistence)."""

    def __init__(self, page_manager: PagedSequence, layer: int = 0):
        self.page_manager = page_manager
        self.layer = layer
        self.step = TOKEN_CAPACITY_PER_PAGE

    @property
    def offset(self) -> int:
        return self.page_manager.offset

    @property
    def capacity(self) -> int:
        return self.page_manager.capacity

    def reuse(self, new_prompt_length: int, common_prefix_length: int) -> None:
        """Trim to the common prefix (reusable.py:44-94); room for the new prompt is taken page by page later."""
        if self.layer == 0:
            self.page_manager.truncate(common_prefix_length)

    def is_trimmable(self) -> bool:
        return True

    def trim(self, n: int) -> int:
        n = min(self.offset, n)
        if self.layer == 0:
            self.page_manager.truncate(self.offset - n)
        return n

    def _rows(self, which: int) -> torch.Tensor:
        seq, a = self.page_manager, self.page_manager.allocator
        half = TOKEN_CAPACITY_PER_PAGE * a.num_heads * a.head_dim * a.dtype.itemsize  # (int8 pages: the raw codes; their scales follow the blocks)
        if not seq.pages:
            return torch.zeros((1, a.num_heads, 0, a.head_dim), dtype=a.dtype, device=a.slab.device)
        blocks = [a.slab[self.layer, p * a.page_bytes + which * half: p * a.page_bytes + (which + 1) * half].view(a.dtype).view(
            a.num_heads, TOKEN_CAPACITY_PER_PAGE, a.head_dim) for p in seq.pages]
        return torch.cat(blocks, dim=1)[:, :seq.offset].unsqueeze(0).contiguous()

    @property
    def state(self):
        """(keys, values) [1, n_kv_heads, offset, head_dim], gathered from the pages (the layout save_cache stores)."""
        return self._rows(0), self._rows(1)

    @state.setter
    def state(self, v):
        keys, values = v
        seq, a = self.page_manager, self.page_manager.allocator
        if a.dtype == torch.int8:
            raise NotImplementedError("restoring rows into int8 pages is not supported (their scales are the pool's)")
        n = keys.shape[2]
        if self.layer == 0 or seq.offset != n:
            seq.truncate(0)
            seq.reserve(n)
            seq.offset = n
            seq._mark_tokens()
        half = a.page_bytes // 2
        for j, p in enumerate(seq.pages):
            rows = min(TOKEN_CAPACITY_PER_PAGE, n - j * TOKEN_CAPACITY_PER_PAGE)
            for which, src in ((0, keys), (1, values)):
                dst = a.slab[self.layer, p * a.page_bytes + which * half: p * a.page_bytes + (which + 1) * half].view(a.dtype).view(
                    a.num_heads, TOKEN_CAPACITY_PER_PAGE, a.head_dim)
                dst[:, :rows] = src[0, :, j * TOKEN_CAPACITY_PER_PAGE: j * TOKEN_CAPACITY_PER_PAGE + rows].to(a.slab.device)

    @property
    def meta_state(self):
        return ""

    @meta_state.setter
    def meta_state(self, v):
        pass

    def update_and_fetch(self, keys: torch.Tensor, values: torch.Tensor):
        raise NotImplementedError("the decode kernels append to the pages themselves (pie_decoder_set_paged_kv)")

    def to_quantized(self, group_size: int = 64, bits: int = 4):
        return self
