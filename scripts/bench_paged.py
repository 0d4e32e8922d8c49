"""Times pie_paged_attn_decode against the contiguous decode attention on the same rows (HBM GB/s of K+V read).

    python scripts/bench_paged.py [--batch 8] [--ctx 8192] [--iters 50]
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from proxy_inference_engine_amd import hip_ops as ops  # noqa: E402
from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator  # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-dim", type=int, default=128)
    args = ap.parse_args()
    B, T, Hq, Hkv, D = args.batch, args.ctx, args.heads, args.kv_heads, args.head_dim
    blocks = (T + 63) // 64
    n_pages = B * blocks
    alloc = PageAllocator(n_pages, Hkv, D, device="cuda")
    alloc.slab.view(torch.bfloat16).normal_()
    rng = np.random.default_rng(0)
    table = rng.permutation(n_pages).astype(np.int32).reshape(B, blocks)       # fully scattered pages
    bt = torch.from_numpy(table).cuda()
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    q = torch.randn(B, Hq, D, device="cuda").bfloat16()
    t_paged = timed(lambda: ops.paged_attention_decode(q, alloc.slab[0], n_pages, bt, lens, Hkv, D ** -0.5), args.iters)
    kv_bytes = B * T * Hkv * D * 2 * 2
    k = torch.randn(1, Hkv, T, D, device="cuda").bfloat16()
    v = torch.randn(1, Hkv, T, D, device="cuda").bfloat16()
    q1 = q[:1].view(1, Hq, 1, D).contiguous()
    t_contig = timed(lambda: ops.scaled_dot_product_attention(q1, k, v, D ** -0.5, T=T), args.iters)
    print(json.dumps({"batch": B, "ctx": T, "paged_us": round(t_paged * 1e6, 1), "paged_GBps": round(kv_bytes / t_paged / 1e9, 1),
                      "contiguous_1seq_us": round(t_contig * 1e6, 1), "contiguous_1seq_GBps": round(kv_bytes / B / t_contig / 1e9, 1)}))


if __name__ == "__main__":
    main()
