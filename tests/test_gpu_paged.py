"""Paged KV on the GPU: the page pool's slab, pie_paged_kv_append and pie_paged_attn_decode.
The reference has only placeholders for the kernel (src/pie_core/src/layers/attention.cpp:71-83,
src/kernels/paged_attention.metal:6-23), so the parity statement is the domain property: attention over pages equals
attention over the same rows gathered contiguously -- checked against the CPU oracle's sdpa and, bit for bit where the
split geometry coincides, against the contiguous decode kernel."""
import numpy as np
import pytest
import torch

from oracle import pie_oracle as po
from tests._util import assert_dot_close, to_bits, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from proxy_inference_engine_amd import hip_ops
    return hip_ops


def _tdt(dt):
    return torch.bfloat16 if dt == "bfloat16" else torch.float16


def _fill(ops, alloc, rng, lens, Hkv, D, dt, max_blocks):
    """Allocates pages for sequences of the given lengths in an interleaved order (so every table is scattered), appends
    the rows token by token in batches through pie_paged_kv_append; returns (block_table, k_rows, v_rows per sequence)."""
    B = len(lens)
    table = np.zeros((B, max_blocks), np.int32)
    need = [(n + 63) // 64 for n in lens]
    for j in range(max(need)):
        for s in rng.permutation(B):
            if j < need[s]:
                pid = alloc.allocate_page()
                assert pid is not None
                table[s, j] = pid
    ks = [po.round_T(rng.standard_normal((n, Hkv, D)), dt) for n in lens]
    vs = [po.round_T(rng.standard_normal((n, Hkv, D)), dt) for n in lens]
    bt = torch.from_numpy(table).cuda()
    for t in range(max(lens)):
        kb = np.zeros((B, Hkv, D), np.float32)
        vb = np.zeros((B, Hkv, D), np.float32)
        pos = np.full(B, -1, np.int32)
        for s in range(B):
            if t < lens[s]:
                kb[s], vb[s], pos[s] = ks[s][t], vs[s][t], t
        ops.paged_kv_append(to_dev(po.to_bits(kb, dt), dt), to_dev(po.to_bits(vb, dt), dt), alloc.slab[0], alloc.size(), bt,
                            torch.from_numpy(pos).cuda())
    return bt, table, ks, vs


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("Hq,Hkv,D,lens", [
    (8, 2, 128, [1, 63, 64, 65, 300]),
    (32, 8, 128, [777, 5, 1500]),
    (6, 2, 64, [129, 0, 64, 31]),          # an idle slot (context 0) in the batch
    (4, 4, 64, [200]),
])
def test_paged_attention_vs_oracle(ops, dt, Hq, Hkv, D, lens):
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    rng = np.random.default_rng(Hq * 100 + len(lens))
    max_blocks = (max(lens) + 63) // 64 + 1
    n_pages = sum((n + 63) // 64 for n in lens) + 3
    alloc = PageAllocator(n_pages, Hkv, D, dtype=_tdt(dt), device="cuda")
    alloc.slab.view(torch.int16).fill_(0x7BFF if dt == "float16" else 0x7F00)    # poison: stale rows are huge (f16 65504 / bf16 1.7e38)
    bt, table, ks, vs = _fill(ops, alloc, rng, lens, Hkv, D, dt, max_blocks)
    B = len(lens)
    q = po.round_T(rng.standard_normal((B, Hq, D)), dt)
    got = ops.paged_attention_decode(to_dev(po.to_bits(q, dt), dt), alloc.slab[0], n_pages, bt,
                                     torch.tensor(lens, dtype=torch.int32, device="cuda"), Hkv, D ** -0.5)
    got = to_bits(got)
    for s, n in enumerate(lens):
        if n == 0:
            assert not got[s].any(), "an idle slot must produce zeros"
            continue
        k = np.ascontiguousarray(ks[s].transpose(1, 0, 2))       # [Hkv, n, D]
        v = np.ascontiguousarray(vs[s].transpose(1, 0, 2))
        want = po.sdpa(q[s][:, None, :], k, v, D ** -0.5, None, dt, True, T=n)
        # a weighted mean that cancels to nearly zero has a tiny ulp of its own: the ulp is taken at max(|want|, max / 128)
        assert_dot_close(po.from_bits(got[s], dt), po.round_T(want, dt), dt, max_frac=0.03, what=f"paged seq {s} len {n} {Hq}/{Hkv} D{D} {dt}")
    # the pages hold exactly the appended rows, in the reference's logical [64, heads, head_dim] view (page.hpp:29-30)
    s = int(np.argmax(lens))
    page = alloc.get_page(int(table[s, 0]))
    rows = min(64, lens[s])
    assert np.array_equal(to_bits(page.key_cache()[:rows].contiguous()), po.to_bits(ks[s][:rows], dt))
    assert np.array_equal(to_bits(page.value_cache()[:rows].contiguous()), po.to_bits(vs[s][:rows], dt))


def test_paged_equals_contiguous_kernel_bitwise(ops):
    """Same rows, same split geometry (4 splits): the paged and the contiguous decode kernels must agree bit for bit."""
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    dt, Hq, Hkv, D, n = "bfloat16", 32, 8, 128, 250
    rng = np.random.default_rng(11)
    # 16 x 8 x 4 = 512 workgroups -> 4 splits, the contiguous op's choice for 128 <= T < 512
    lens = [n] * 16
    alloc = PageAllocator(16 * 4 + 1, Hkv, D, device="cuda")
    bt, table, ks, vs = _fill(ops, alloc, rng, lens, Hkv, D, dt, 4)
    q = po.round_T(rng.standard_normal((16, Hq, D)), dt)
    got = ops.paged_attention_decode(to_dev(po.to_bits(q, dt), dt), alloc.slab[0], alloc.size(), bt,
                                     torch.tensor(lens, dtype=torch.int32, device="cuda"), Hkv, D ** -0.5)
    for s in (0, 7, 15):
        k = np.zeros((Hkv, 256, D), np.float32)
        v = np.zeros((Hkv, 256, D), np.float32)
        k[:, :n], v[:, :n] = ks[s].transpose(1, 0, 2), vs[s].transpose(1, 0, 2)
        ref = ops.scaled_dot_product_attention(to_dev(po.to_bits(q[s], dt), dt).view(1, Hq, 1, D), to_dev(po.to_bits(k, dt), dt).view(1, Hkv, 256, D),
                                               to_dev(po.to_bits(v, dt), dt).view(1, Hkv, 256, D), D ** -0.5, T=n)
        assert np.array_equal(to_bits(got[s]).reshape(-1), to_bits(ref).reshape(-1))


def test_shared_prefix_pages(ops):
    """Two sequences sharing their first pages by reference count (add_ref): both read the same rows; freeing one keeps
    the shared pages alive for the other (page_allocator.cpp:81-92)."""
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    dt, Hq, Hkv, D = "bfloat16", 8, 2, 128
    rng = np.random.default_rng(5)
    alloc = PageAllocator(8, Hkv, D, device="cuda")
    bt, table, ks, vs = _fill(ops, alloc, rng, [128], Hkv, D, dt, 3)          # the shared 2-page prefix
    shared = [int(table[0, 0]), int(table[0, 1])]
    for pid in shared:
        alloc.add_ref(pid)
    tail_a, tail_b = alloc.allocate_page(), alloc.allocate_page()
    table2 = np.array([shared + [tail_a], shared + [tail_b]], np.int32)
    bt2 = torch.from_numpy(table2).cuda()
    ka, kb = (po.round_T(rng.standard_normal((1, Hkv, D)), dt) for _ in range(2))
    va, vb = (po.round_T(rng.standard_normal((1, Hkv, D)), dt) for _ in range(2))
    ops.paged_kv_append(to_dev(po.to_bits(np.concatenate([ka, kb]), dt), dt), to_dev(po.to_bits(np.concatenate([va, vb]), dt), dt),
                        alloc.slab[0], alloc.size(), bt2, torch.tensor([128, 128], dtype=torch.int32, device="cuda"))
    q = po.round_T(rng.standard_normal((2, Hq, D)), dt)
    got = to_bits(ops.paged_attention_decode(to_dev(po.to_bits(q, dt), dt), alloc.slab[0], alloc.size(), bt2,
                                             torch.tensor([129, 129], dtype=torch.int32, device="cuda"), Hkv, D ** -0.5))
    for s, (kt, vt) in enumerate(((ka, va), (kb, vb))):
        k = np.ascontiguousarray(np.concatenate([ks[0], kt]).transpose(1, 0, 2))
        v = np.ascontiguousarray(np.concatenate([vs[0], vt]).transpose(1, 0, 2))
        want = po.sdpa(q[s][:, None, :], k, v, D ** -0.5, None, dt, True, T=129)
        assert_dot_close(po.from_bits(got[s], dt), po.round_T(want, dt), dt, max_frac=0.03, what=f"shared prefix seq {s}")
    free0 = alloc.get_num_free_pages()
    for pid in shared + [tail_a]:                                             # sequence A ends
        alloc.free_page(pid)
    assert alloc.get_num_free_pages() == free0 + 1                            # only its private tail came back
    assert all(alloc.get_page(pid).get_ref_count() == 1 for pid in shared)


def test_paged_argument_errors(ops):
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    alloc = PageAllocator(2, 2, 128, device="cuda")
    q = torch.zeros(1, 8, 128, dtype=torch.bfloat16, device="cuda")
    bt = torch.zeros(1, 1, dtype=torch.int32, device="cuda")
    with pytest.raises(TypeError):
        ops.paged_attention_decode(q, alloc.slab[0], 2, bt.long(), torch.ones(1, dtype=torch.int32, device="cuda"), 2, 1.0)
    with pytest.raises(ValueError):
        ops.paged_attention_decode(q, alloc.slab[0], 3, bt, torch.ones(1, dtype=torch.int32, device="cuda"), 2, 1.0)   # slab too small
    with pytest.raises(ValueError):
        ops.paged_attention_decode(q, alloc.slab[0], 2, bt, torch.ones(2, dtype=torch.int32, device="cuda"), 2, 1.0)
