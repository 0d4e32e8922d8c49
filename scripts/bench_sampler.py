#!/usr/bin/env python
"""Latency of one sampling call at the Llama-3 vocabulary (V = 128256): the fused HIP kernel vs the torch-op restatement it replaced."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from proxy_inference_engine_amd import hip_ops  # noqa: E402
from proxy_inference_engine_amd.samplers import categorical, min_p, top_k, top_p  # noqa: E402

V = 128256
x = torch.log_softmax(torch.randn(1, V, device="cuda") * 3, dim=-1)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def torch_top_p(lp, p, t):
    probs = torch.softmax(lp.float() * (1 / t), dim=-1)
    sp, si = torch.sort(probs, dim=-1)
    cum = torch.cumsum(sp, dim=-1)
    tp = torch.where(cum > 1 - p, sp, torch.zeros_like(sp))
    tok = categorical.sample_from_logits(torch.log(tp).cpu()).to(lp.device)[..., None]
    return si.gather(-1, tok.long())


for name, hip, ref in (("top_p 0.9", lambda: hip_ops.sample(x, "top_p", 1.0, p=0.9), None),
                       ("min_p 0.05", lambda: hip_ops.sample(x, "min_p", 1.0, p=0.05, k=1), None),
                       ("top_k 40", lambda: hip_ops.sample(x, "top_k", 1.0, k=40), None),
                       ("categorical", lambda: hip_ops.sample(x, "categorical", 1.0), None)):
    print(f"{name:12s}: HIP kernel {timed(hip):7.1f} us per call", flush=True)
srt = timed(lambda: torch.sort(x, dim=-1))
print(f"for scale: torch.sort of one [1, {V}] fp32 row alone {srt:7.1f} us")


def device_us(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("device time per call (HIP events around 50 back-to-back calls):")
for name, hip in (("top_p 0.9", lambda: hip_ops.sample(x, "top_p", 1.0, p=0.9)), ("min_p 0.05", lambda: hip_ops.sample(x, "min_p", 1.0, p=0.05, k=1)),
                  ("top_k 40", lambda: hip_ops.sample(x, "top_k", 1.0, k=40)), ("categorical", lambda: hip_ops.sample(x, "categorical", 1.0))):
    print(f"  {name:12s}: {device_us(hip):7.1f} us")
