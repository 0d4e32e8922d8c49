#!/usr/bin/env python
"""Prompt-processing (prefill) throughput of the Llama-3-8B-shaped int4 model on one MI355X: time to first token.

    python scripts/bench_prefill.py [--prompts 128,1024,4096] [--layers 32]

Prints one line per prompt length: batched prefill (dequantise-to-T + hipBLASLt GEMMs, knob prefill_min at its default) against
the iterated-decode-step prompt path (prefill_min = 1000000).  Developer tool; bench.py stays the decode metric.
"""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompts", default="128,1024,4096")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--dense", action="store_true", help="unquantised bf16 weights (BASELINE.json configs[2])")
    ap.add_argument("--iterated-max", type=int, default=1024, help="longest prompt also timed through iterated decode steps")
    args = ap.parse_args()
    from proxy_inference_engine_amd import _ffi
    from proxy_inference_engine_amd.models.llama import Model, ModelArgs
    from proxy_inference_engine_amd.models.utils import LLAMA3_8B, synthetic_checkpoint

    cfg = dict(LLAMA3_8B, num_hidden_layers=args.layers)
    if args.dense:
        cfg["quantization"] = None
    model = Model(ModelArgs(**cfg), synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16))
    torch.cuda.empty_cache()
    for L in [int(x) for x in args.prompts.split(",")]:
        ids = torch.randint(0, cfg["vocab_size"], (L,), generator=torch.Generator().manual_seed(L)).cuda()
        out = {}
        for mode, env in (("batched", None), ("iterated", 1000000)):
            if mode == "iterated" and L > args.iterated_max:
                continue
            _ffi.set_knob("prefill_min", env)
            best = None
            for rep in range(3):
                cache = model.make_cache()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tok, _, _ = model.step(ids, cache)
                tok.item()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            out[mode] = best
        _ffi.set_knob("prefill_min", None)
        line = f"prompt {L:6d}: batched {1e3 * out['batched']:9.2f} ms = {L / out['batched']:9.0f} tok/s"
        if "iterated" in out:
            line += f" | iterated {1e3 * out['iterated']:9.2f} ms = {L / out['iterated']:7.0f} tok/s | speed-up {out['iterated'] / out['batched']:.1f}x"
        print(line, flush=True)


if __name__ == "__main__":
    main()
