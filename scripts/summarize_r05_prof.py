"""Turns gpurun_out/r5/prof_step (scripts/profile_r05.sh on the MI355X box) into the committed summaries under profiles/; every CSV it copies gets
a first line naming the library build (pie_version()) it was measured on."""
import glob
import json
import os
import re  # noqa: F401
import shutil
from pathlib import Path

import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
# PIE_PROF_BITS=2 | 6: the same passes over the 2- / 6-bit step (step_bench --bits, scripts/profile_r05.sh with BITS=), W2S / W6S units
BITS = int(os.environ.get("PIE_PROF_BITS", "4"))
SFX = "" if BITS == 4 else f"_bits{BITS}"
FMT = {4: 0, 2: 5, 6: 6}[BITS]  # the kernel name's FMT template argument (common.hpp)
P = ROOT / "gpurun_out" / "r5" / ("prof_step" + SFX)
OUT = ROOT / "profiles"


def counters(d):
    df = pd.read_csv(sorted(glob.glob(str(P / d / "*" / "*_counter_collection.csv")), key=lambda q: Path(q).stat().st_mtime)[-1])  # (gpurun merges runs: newest)
    return df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


version = (P / "pie_version.txt").read_text().strip()
for src, dst in (("step_stats", f"r05_step{SFX}_kernel_stats.csv"), ("bench_stats", f"r05_bench{SFX}_kernel_stats.csv")):
    hit = sorted(glob.glob(str(P / src / "*" / "*_kernel_stats.csv")), key=lambda q: Path(q).stat().st_mtime)[-1:]
    if hit:
        (OUT / dst).write_text(f"# library: {version}; rocprofv3 --kernel-trace --stats, scripts/profile_r05.sh ({src})\n" + Path(hit[0]).read_text())

H, I, QD, KVD, V, L = 4096, 14336, 4096, 1024, 128256, 32
T = 133


def lin(n, k):
    return n * k * BITS // 8 + 2 * (n * k // 64) * 2


QKV = lin(QD + 2 * KVD, H) + H * 2 + 2 * KVD * 2
ATT = 2 * KVD * 2 * T  # the K and V rows of the context
alg = {f"k_w4s_gemv<BF16, 1, 2, 1, 0, {FMT}, 1>": ("qkv: rmsnorm + GEMV + RoPE + cache append, then -- behind the XCD-local seam of the same launch -- the split-KV attention (partials merged by o_proj) "
                                               "and the Infinity-Cache warm-up of o_proj on the workgroups that do neither", QKV + ATT, L - 1 if BITS == 4 else L),
       "k_w4s_gemv<BF16, 3, 2, 1, 0, 0, 1>": ("layer 0's qkv + attention with the embedding row dequantised in its prologue", QKV + ATT, 1),
       f"k_w4s_gemv<BF16, 2, 1, 1, 0, {FMT}, 0>": ("o_proj: split merge + GEMV + residual", lin(H, QD), L),
       f"k_w4s_gemv<BF16, 1, 3, 1, 0, {FMT}, 0>": ("gate/up: rmsnorm + GEMV + SwiGLU", lin(2 * I, H) + H * 2, L),
       f"k_w4s_gemv<BF16, 0, 1, 4, 0, {FMT}, 0>": ("down: GEMV + residual", lin(H, I), L),
       f"k_w4s_gemv<BF16, 1, 4, 1, 0, {FMT}, 0>": ("lm_head: rmsnorm + GEMV + log-softmax partials", lin(V, H) + H * 2, 1),
       "k_logits_finish<BF16>": ("tail: log-softmax + argmax", V * 4, 1)}
f, w = counters("step_fetch"), counters("step_write")


def pick(df, key):
    names = df.Kernel_Name.map(short)
    hit = df[names == key]
    return hit if len(hit) else df[names.str.startswith(key.rsplit(",", 1)[0])]


rows, step_meas, step_alg = {}, 0.0, 0.0
for key, (what, a, per_step) in alg.items():
    pf, pw = pick(f, key), pick(w, key)
    if not len(pf) or not len(pw):
        print("not in the capture:", key)
        continue
    fk = pf["mean"].iloc[0] * 1024 * 2  # KB; gfx950: wide streaming reads are tallied at half their size (MI355X_MICROARCH.md, HBM)
    wk = pw["mean"].iloc[0] * 1024
    rows[key] = {"what": what, "launches_per_step": per_step, "algorithmic_bytes": a, "hbm_bytes_per_launch": int(fk + wk),
                 "FETCH_SIZE_KB": round(fk / 2048, 2), "WRITE_SIZE_KB": round(wk / 1024, 2), "ratio": round((fk + wk) / a, 4)}
    step_meas += (fk + wk) * per_step
    step_alg += a * per_step
gu = rows[f"k_w4s_gemv<BF16, 1, 3, 1, 0, {FMT}, 0>"]
out = {"library": version, "kernel": "k_w4s_gemv<BF16, rmsnorm, swiglu> (gate/up, N=28672 K=4096)", "hbm_bytes_per_launch": gu["hbm_bytes_per_launch"],
       "algorithmic_bytes_per_launch": gu["algorithmic_bytes"], "FETCH_SIZE_KB": gu["FETCH_SIZE_KB"], "WRITE_SIZE_KB": gu["WRITE_SIZE_KB"],
       "step": {"hbm_bytes": int(step_meas), "algorithmic_bytes": int(step_alg), "ratio": round(step_meas / step_alg, 4), "context": T},
       "per_kernel": rows,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the PRODUCT decode step driven through the C ABI by "
               "tools/step_bench (--graph 0: the launch sequence of pie_decoder_step, the kernels and arguments the hipGraph replays; "
               "8 steps after 2 warm-up, means over all dispatches).  FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced "
               "streams at 64 B: MI355X_MICROARCH.md, HBM section; calibrated in round 1 against a pure stream kernel).  `library` is the "
               "pie_version() of the build that was measured: bench.py reports traffic only for exactly that build."}
out["weight_bits"] = BITS
json.dump(out, open(OUT / f"r05_traffic{SFX}.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1)[:1500])
