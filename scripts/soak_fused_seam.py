#!/usr/bin/env python
"""Soak of the fused q|k|v + attention launch (w4_gemv.hpp FUSE): a 32-layer Llama-3-8B-shaped int4 model decodes N greedy steps twice -- with the
attention behind the XCD-local seam and as two launches (knob fuse_attn = 0) -- and every step's token and logits must be identical bit for bit.
A stale read through the seam (a race) would show up as a mismatch; so would a give-up of its bounded wait (pie_decoder_status).

    python scripts/soak_fused_seam.py [--steps 3000] [--prompt 100]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--prompt", type=int, default=100)
    args = ap.parse_args()
    from proxy_inference_engine_amd import _ffi
    from proxy_inference_engine_amd.models.llama import Model, ModelArgs
    from proxy_inference_engine_amd.models.utils import LLAMA3_8B, synthetic_checkpoint

    cfg = dict(LLAMA3_8B)
    weights = synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16)
    prompt = torch.randint(0, cfg["vocab_size"], (args.prompt,), generator=torch.Generator().manual_seed(7)).cuda()
    runs = {}
    for mode in (0, None):
        _ffi.set_knob("fuse_attn", mode)
        model = Model(ModelArgs(**cfg), weights)
        cache = model.make_cache()
        tok, _, logits = model.step(prompt, cache)
        toks, sums = [], []
        # stay inside the short-cache plan (capacity <= 1024): restart from the prompt when the context nears it
        for i in range(args.steps):
            if cache[0].offset >= 1000:
                cache = model.make_cache()
                tok, _, logits = model.step(prompt, cache)
            tok, _, logits = model.step(tok, cache)
            toks.append(tok.clone())
            sums.append(logits.view(torch.int16).to(torch.int64).sum().reshape(1))
        err = C.c_uint(1)
        _ffi.check(_ffi.load().pie_decoder_status(model._dec, C.byref(err)))
        runs[mode] = (torch.cat(toks).cpu(), torch.cat(sums).cpu(), model.graph_launches(True), err.value)
        del model, cache
    a, b = runs[0], runs[None]
    bad_t = int((a[0] != b[0]).sum())
    bad_l = int((a[1] != b[1]).sum())
    print(f"{args.steps} steps: launches per step {a[2]} (two launches) vs {b[2]} (fused); token mismatches {bad_t}, logit-checksum mismatches {bad_l}; "
          f"status {a[3]:#x} / {b[3]:#x}")
    sys.exit(1 if bad_t or bad_l or a[3] or b[3] or a[2] == b[2] else 0)


if __name__ == "__main__":
    main()
