"""Turns gpurun_out/r2/prof (scripts/profile_r02.sh on the MI355X box) into the committed summaries under profiles/."""
import glob
import json
import shutil
from pathlib import Path

import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "gpurun_out" / "r2" / "prof"
OUT = ROOT / "profiles"


def counters(d):
    df = pd.read_csv(glob.glob(str(P / d / "runc" / "*_counter_collection.csv"))[0])
    return df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()


def short(name):
    return name.split("(")[0].replace("void ", "")


shutil.copy(glob.glob(str(P / "step_stats" / "runc" / "*_kernel_stats.csv"))[0], OUT / "r02_step_kernel_stats.csv")
shutil.copy(glob.glob(str(P / "prefill_stats" / "runc" / "*_kernel_stats.csv"))[0], OUT / "r02_prefill4096_kernel_stats.csv")
bs = glob.glob(str(P / "bench_stats" / "*" / "*_kernel_stats.csv"))
if bs:
    shutil.copy(bs[0], OUT / "r02_bench_kernel_stats.csv")

# ---- HBM traffic of the product decode step (8B int4, context 128..138, capacity 512, eager launches of the same kernels)
H, I, QD, KVD, V, L = 4096, 14336, 4096, 1024, 128256, 32
T = 133


def lin(n, k):
    return n * k // 2 + 2 * (n * k // 64) * 2


alg = {"k_w4s_gemv<BF16, 1, 2, 1, 0, 0>": ("qkv: rmsnorm + GEMV + RoPE + cache append", lin(QD + 2 * KVD, H) + H * 2 + 2 * KVD * 2, L),
       "k_attn_decode<BF16, 128, 4, false, false>": ("split-KV attention (cache capacity 512: partials merged by o_proj)", 2 * KVD * 2 * T, L),
       "k_w4s_gemv<BF16, 2, 1, 1, 0, 0>": ("o_proj: split merge + GEMV + residual", lin(H, QD), L),
       "k_w4s_gemv<BF16, 1, 3, 1, 0, 0>": ("gate/up: rmsnorm + GEMV + SwiGLU", lin(2 * I, H) + H * 2, L),
       "k_w4s_gemv<BF16, 0, 1, 4, 0, 0>": ("down: GEMV + residual", lin(H, I), L),
       "k_w4s_gemv<BF16, 1, 4, 1, 0, 0>": ("lm_head: rmsnorm + GEMV + log-softmax partials", lin(V, H) + H * 2, 1),
       "k_logits_finish<BF16>": ("tail: log-softmax + argmax", V * 4, 1),
       "k_embedding_w4g64<BF16, 4>": ("embedding row + RoPE table", H // 2 + 2 * (H // 64) * 2, 1)}
f, w = counters("step_fetch"), counters("step_write")


def pick(df, key):
    """Rows of one kernel; a key matches its exact short name or, for templates that grew parameters, its prefix."""
    names = df.Kernel_Name.map(short)
    hit = df[names == key]
    return hit if len(hit) else df[names.str.startswith(key[:-1] + ",")]


rows, step_meas, step_alg = {}, 0.0, 0.0
for key, (what, a, per_step) in alg.items():
    fk = pick(f, key)["mean"].iloc[0] * 1024 * 2  # KB; gfx950: wide streaming reads are tallied at half their size
    wk = pick(w, key)["mean"].iloc[0] * 1024
    rows[key] = {"what": what, "launches_per_step": per_step, "algorithmic_bytes": a, "hbm_bytes_per_launch": int(fk + wk),
                 "FETCH_SIZE_KB": round(fk / 2048, 2), "WRITE_SIZE_KB": round(wk / 1024, 2), "ratio": round((fk + wk) / a, 4)}
    step_meas += (fk + wk) * per_step
    step_alg += a * per_step
gu = rows["k_w4s_gemv<BF16, 1, 3, 1, 0, 0>"]
json.dump({"kernel": "k_w4s_gemv<BF16, rmsnorm, swiglu> (gate/up, N=28672 K=4096)", "hbm_bytes_per_launch": gu["hbm_bytes_per_launch"],
           "algorithmic_bytes_per_launch": gu["algorithmic_bytes"], "FETCH_SIZE_KB": gu["FETCH_SIZE_KB"], "WRITE_SIZE_KB": gu["WRITE_SIZE_KB"],
           "step": {"hbm_bytes": int(step_meas), "algorithmic_bytes": int(step_alg), "ratio": round(step_meas / step_alg, 4), "context": T},
           "per_kernel": rows,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the PRODUCT decode step driven through the C ABI by "
                   "tools/step_bench (--mode launch --graph 0: the launch sequence of pie_decoder_step, the kernels and arguments the hipGraph replays; "
                   "8 steps after 2 warm-up, means over all dispatches).  FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced "
                   "streams at 64 B: MI355X_MICROARCH.md, HBM section; calibrated in round 1 against a pure stream kernel).  The doubling over-states "
                   "the small kernels whose reads are not wide streams (attention, tail, embedding: a few hundred KB per step in all)."},
          open(OUT / "r02_traffic.json", "w"), indent=1)

# ---- MFMA activity of the prompt path (4096-token prefill, 8B int4): the many-row W4 GEMM and the causal attention
m, b = counters("prefill_mfma"), counters("prefill_busy")
out = {}
for name in sorted(set(m.Kernel_Name)):
    mm = m[(m.Kernel_Name == name) & (m.Counter_Name == "SQ_INSTS_VALU_MFMA_MOPS_BF16")]
    if mm.empty or mm["mean"].iloc[0] == 0:
        continue
    cu = m[(m.Kernel_Name == name) & (m.Counter_Name == "SQ_BUSY_CU_CYCLES")]["mean"].iloc[0]
    bb = b[b.Kernel_Name == name].set_index("Counter_Name")["mean"]
    mops = float(mm["mean"].iloc[0])
    out[short(name)] = {"launches": int(mm["count"].iloc[0]), "SQ_INSTS_VALU_MFMA_MOPS_BF16": mops, "mfma_flops_per_launch": mops * 512,
                        "SQ_BUSY_CU_CYCLES": float(cu), "GRBM_GUI_ACTIVE": float(bb.get("GRBM_GUI_ACTIVE", 0)),
                        "SQ_BUSY_CYCLES": float(bb.get("SQ_BUSY_CYCLES", 0)), "SQ_WAVES": float(bb.get("SQ_WAVES", 0))}
# the few-row kernel (16-token prompt)
try:
    m16 = counters("prefill16_mfma")
    st16 = pd.read_csv(glob.glob(str(P / "prefill16_stats" / "runc" / "*_kernel_stats.csv"))[0])
    shutil.copy(glob.glob(str(P / "prefill16_stats" / "runc" / "*_kernel_stats.csv"))[0], OUT / "r02_prefill16_kernel_stats.csv")
    for name in sorted(set(m16.Kernel_Name)):
        mm = m16[(m16.Kernel_Name == name) & (m16.Counter_Name == "SQ_INSTS_VALU_MFMA_MOPS_BF16")]
        if mm.empty or mm["mean"].iloc[0] == 0 or "w4m" not in name:
            continue
        dur = st16[st16.Name == name]["AverageNs"]
        out[short(name) + " @16 rows"] = {"launches": int(mm["count"].iloc[0]), "SQ_INSTS_VALU_MFMA_MOPS_BF16": float(mm["mean"].iloc[0]),
                                          "mfma_flops_per_launch": float(mm["mean"].iloc[0]) * 512,
                                          "mean_duration_us": float(dur.iloc[0]) / 1e3 if len(dur) else None}
except (IndexError, FileNotFoundError) as e:
    print("no 16-row passes:", e)
# durations of the 4096-row kernels from the kernel-trace pass
st = pd.read_csv(glob.glob(str(P / "prefill_stats" / "runc" / "*_kernel_stats.csv"))[0])
for k in list(out):
    hit = st[st.Name.map(short) == k]
    if len(hit):
        out[k]["mean_duration_us"] = float(hit["AverageNs"].iloc[0]) / 1e3
        out[k]["mfma_TFLOPs"] = out[k]["mfma_flops_per_launch"] / (out[k]["mean_duration_us"] * 1e-6) / 1e12
        out[k]["frac_of_dense_bf16_peak_2500"] = out[k]["mfma_TFLOPs"] / 2500.0
json.dump({"workload": "tools/step_bench --model 8b --no-mega --prefill 4096 (one 4096-token prompt through pie_decoder_prefill, after the one-off tile repack)",
           "note": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES and --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes), means per "
                   "launch over all launches of a kernel (its shapes differ per Linear).  mfma_flops_per_launch = MOPS x 512 (rocprofv3's MfmaFlopsBF16).",
           "kernels": out}, open(OUT / "r02_prefill_mfma.json", "w"), indent=1)
print(json.dumps(json.load(open(OUT / "r02_traffic.json"))["step"]))
for k, v in rows.items():
    print(f"{k:48s} {v['hbm_bytes_per_launch']/1e6:9.3f} MB vs {v['algorithmic_bytes']/1e6:9.3f} MB  x{v['ratio']}")
for k, v in out.items():
    print(k[:70], {a: (round(x, 1) if isinstance(x, float) else x) for a, x in v.items()})
