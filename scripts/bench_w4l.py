#!/usr/bin/env python
"""Microbenchmark of the many-row int4 MFMA GEMM (pie_qgemm_w4m, M > 32) on the Llama-3-8B layer shapes: TFLOP/s per shape.
PIE_HIP_LIB selects another build of the library (ablation builds with -DW4L_ABL=...).  Developer tool."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from proxy_inference_engine_amd import hip_ops as ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, N, K in (("qkv", 6144, 4096), ("o_proj", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
    w = torch.randn(N, K, dtype=torch.bfloat16, device="cuda") * 0.03
    pw = ops.repack_w4s(*ops.quantize(w))
    w4m = ops.repack_w4m(pw)
    x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
    for _ in range(3 if M > 512 else 50):
        ops.quantized_matmul_rows(x, pw, w4m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10 if M > 512 else 300
    e0.record()
    for _ in range(reps):
        ops.quantized_matmul_rows(x, pw, w4m)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:8s} M={M} N={N} K={K}: {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s", flush=True)
