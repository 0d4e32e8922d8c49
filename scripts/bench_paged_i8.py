"""Paged decode attention over int8 pages vs T pages (op level): B sequences x T cached positions, Llama-3-8B head geometry.
   python scripts/bench_paged_i8.py   (on the GPU box)"""
import numpy as np
import torch

from proxy_inference_engine_amd import hip_ops as ops
from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator

Hq, Hkv, D = 32, 8, 128
for B, T in ((1, 8192), (8, 2048), (32, 2048), (64, 4096)):
    nb = (T + 63) // 64
    n_pages = B * nb
    res = {}
    for name, dt in (("bf16", torch.bfloat16), ("int8", torch.int8)):
        alloc = PageAllocator(n_pages, Hkv, D, dtype=dt, device="cuda")
        alloc.slab.view(torch.int8).random_(-100, 100)
        if dt == torch.int8:
            ops.page_i8_set_scales(alloc.slab[0], n_pages, Hkv, D, torch.full((Hkv,), 1 / 32, dtype=torch.float16, device="cuda"),
                                   torch.full((Hkv,), 1 / 32, dtype=torch.float16, device="cuda"))
        else:
            alloc.slab.view(torch.bfloat16).normal_()
        bt = torch.from_numpy(np.random.default_rng(0).permutation(n_pages).astype(np.int32).reshape(B, nb)).cuda()
        ctx = torch.full((B,), T, dtype=torch.int32, device="cuda")
        q = torch.randn(B, Hq, D, device="cuda").to(torch.bfloat16)
        f = ops.paged_attention_decode_i8 if dt == torch.int8 else ops.paged_attention_decode
        for _ in range(5):
            f(q, alloc.slab[0], n_pages, bt, ctx, Hkv, D ** -0.5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f(q, alloc.slab[0], n_pages, bt, ctx, Hkv, D ** -0.5)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        kv_bytes = B * T * Hkv * D * 2 * (1 if dt == torch.int8 else 2)
        res[name] = (us, kv_bytes / us / 1e6)
    print(f"B={B:3d} T={T:5d}: bf16 pages {res['bf16'][0]:8.1f} us ({res['bf16'][1]:5.2f} TB/s)   int8 pages {res['int8'][0]:8.1f} us ({res['int8'][1]:5.2f} TB/s)   "
          f"x{res['bf16'][0] / res['int8'][0]:.2f}")
