"""Times the vision tower (Qwen2.5-VL-7B geometry: 32 blocks, hidden 1280, 16 heads of 80, SwiGLU 3420, merger -> 3584) on
synthetic pixels: ms per image and achieved dense-GEMM + attention TFLOP/s.

    python scripts/bench_vision.py [--grid 32x32 ...] [--iters 5]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from proxy_inference_engine_amd.models.intern.vision import VisionConfig, VisionModel  # noqa: E402


def synth(cfg: VisionConfig, dtype=torch.bfloat16, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(0)
    w = {}
    H, I, O, unit = cfg.hidden_size, cfg.intermediate_size, cfg.out_hidden_size, cfg.spatial_merge_size ** 2

    def lin(name, n, k, bias=True):
        w[name + ".weight"] = (torch.randn(n, k, generator=g, device=dev) / k ** 0.5).to(dtype)
        if bias:
            w[name + ".bias"] = (torch.randn(n, generator=g, device=dev) * 0.1).to(dtype)

    kin = cfg.in_channels * cfg.temporal_patch_size * cfg.patch_size ** 2
    w["vision_tower.patch_embed.proj.weight"] = (torch.randn(H, kin, generator=g, device=dev) / kin ** 0.5).to(dtype).view(
        H, cfg.in_channels, cfg.temporal_patch_size, cfg.patch_size, cfg.patch_size)
    for i in range(cfg.depth):
        p = f"vision_tower.blocks.{i}."
        w[p + "norm1.weight"] = torch.ones(H, dtype=dtype, device=dev)
        w[p + "norm2.weight"] = torch.ones(H, dtype=dtype, device=dev)
        lin(p + "attn.qkv", 3 * H, H), lin(p + "attn.proj", H, H)
        lin(p + "mlp.gate_proj", I, H), lin(p + "mlp.up_proj", I, H), lin(p + "mlp.down_proj", H, I)
    w["vision_tower.merger.ln_q.weight"] = torch.ones(H, dtype=dtype, device=dev)
    lin("vision_tower.merger.mlp.0", H * unit, H * unit), lin("vision_tower.merger.mlp.2", O, H * unit)
    return w


def flops(cfg: VisionConfig, grid, model: VisionModel):
    t, h, wd = grid
    N = t * h * wd
    H, I, D, nh = cfg.hidden_size, cfg.intermediate_size, cfg.hidden_size // cfg.num_heads, cfg.num_heads
    gemm = 2 * N * (cfg.in_channels * cfg.temporal_patch_size * cfg.patch_size ** 2) * H
    gemm += cfg.depth * 2 * N * (3 * H * H + H * H + 3 * H * I)
    gemm += 2 * (N // 4) * (4 * H * 4 * H + 4 * H * cfg.out_hidden_size)
    _, cu = model.get_window_index([grid])
    cu = sorted(set(cu))
    win = sum((b - a) ** 2 for a, b in zip(cu[:-1], cu[1:]))
    full = t * (h * wd) ** 2
    n_full = len(cfg.fullatt_block_indexes)
    attn = 4 * nh * D * (n_full * full + (cfg.depth - n_full) * win)       # useful flops (head_dim 80, not the padded 128)
    return gemm, attn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", nargs="*", default=["16x16", "32x32", "64x64"])
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph of the tower")
    args = ap.parse_args()
    cfg = VisionConfig(depth=32, hidden_size=1280, intermediate_size=3420, out_hidden_size=3584, num_heads=16)
    model = VisionModel(cfg, synth(cfg))
    for gs in args.grid:
        h, w = (int(v) for v in gs.split("x"))
        grid = (1, h, w)
        N = h * w
        pix = torch.randn(N, 1176, device="cuda").bfloat16()
        for _ in range(3):
            model(pix, [grid], graph=args.graph)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            model(pix, [grid], graph=args.graph)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / args.iters
        gemm, attn = flops(cfg, grid, model)
        print(json.dumps({"grid": gs, "graph": args.graph, "patches": N, "image_tokens": N // 4, "ms": round(ms, 3), "gemm_TFLOP": round(gemm / 1e12, 3),
                          "attn_TFLOP": round(attn / 1e12, 4), "TFLOP/s": round((gemm + attn) / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
