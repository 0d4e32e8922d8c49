"""Kernel time of pie_qgemv_w4g64 (k_w4s_gemv_rows from two rows on) by row count, on the 8B model's matrix shapes, with the weights
rotated over enough copies that nothing is served from the Infinity Cache.  Developer helper (PIE_HIP_LIB selects a library variant).

    python scripts/bench_rows.py [--rows 1,2,3,4,5] [--iters 200]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from proxy_inference_engine_amd import _ffi, hip_ops as ops  # noqa: E402

SHAPES = {"q|k|v": (6144, 4096), "o_proj": (4096, 4096), "gate|up": (28672, 4096), "down": (4096, 14336)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="1,2,3,4,5")
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    rows = [int(r) for r in args.rows.split(",")]
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, (N, K) in SHAPES.items():
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        codes, scales, biases = ops.quantize(w)
        nbytes = N * K * 9 // 16
        copies = max(2, int(600e6 // nbytes))
        wts = [ops.repack_w4s(codes, scales, biases) for _ in range(copies)]
        line = f"{name:8s} [{N} x {K}] {nbytes / 1e6:6.1f} MB x {copies:2d} copies:"
        for M in rows:
            x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            for i in range(copies):
                ops.quantized_matmul(x, wts[i])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.iters):
                ops.quantized_matmul(x, wts[i % copies])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            line += f"  M={M}: {us:6.1f} us ({nbytes / us / 1e6:4.2f} TB/s)"
        print(line, flush=True)
        lib = _ffi.load()
        if hasattr(lib, "pie_debug_rows_prof"):  # -DPIE_ROWS_PROF build: stamps of workgroup 10 during the last launch (100 MHz)
            import ctypes as C
            buf = (C.c_ulonglong * 128)()
            lib.pie_debug_rows_prof(buf)
            for w in range(8):
                t = [buf[w * 16 + i] for i in range(16)]
                print(f"    wave {w}:", " ".join(f"{(v - t[0]) / 100:5.2f}" if v else "    -" for v in t[:10] + t[14:]),
                      f"| units {t[10]}: cycles per unit: unpack {t[11] / max(t[10], 1):.0f}, rows {t[12] / max(t[10], 1):.0f}, issue {t[13] / max(t[10], 1):.0f}", flush=True)
        del wts


if __name__ == "__main__":
    main()
