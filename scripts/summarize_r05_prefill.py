"""gpurun_out/r5/prof_prefill (scripts/profile_r05_prefill.sh) -> profiles/r05_prefill{128,512,4096}_kernel_stats.csv, r05_prefill_mfma.json,
r05_dense_prefill512.json: per-kernel mean durations, MFMA operations per launch (SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = FLOPs) and the fraction
of the dense bf16 MFMA peak (2.5 PFLOP/s) each kernel and each prompt length reaches."""
import glob
import json
import sys
from pathlib import Path

import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "gpurun_out" / "r5" / "prof_prefill"
OUT = ROOT / "profiles"
PEAK = 2500.0  # TFLOP/s, dense bf16 MFMA (MI355X_MICROARCH.md)


def counters(d):
    f = sorted(glob.glob(str(P / d / "*" / "*_counter_collection.csv")), key=lambda q: Path(q).stat().st_mtime)  # (gpurun merges runs: newest)
    if not f:
        return None
    df = pd.read_csv(f[-1])
    return df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def one(tag, header):
    stats = sorted(glob.glob(str(P / f"stats_{tag}" / "*" / "*_kernel_stats.csv")), key=lambda q: Path(q).stat().st_mtime)
    if not stats:
        return None
    st = pd.read_csv(stats[-1])
    (OUT / f"r05_prefill{tag}_kernel_stats.csv").write_text(header + Path(stats[-1]).read_text())
    m, b = counters(f"mfma_{tag}"), counters(f"busy_{tag}")
    out = {}
    if m is not None:
        for name in sorted(set(m.Kernel_Name)):
            mm = m[(m.Kernel_Name == name) & (m.Counter_Name == "SQ_INSTS_VALU_MFMA_MOPS_BF16")]
            if mm.empty or mm["mean"].iloc[0] == 0:
                continue
            mops = float(mm["mean"].iloc[0])
            k = short(name)
            out[k] = {"launches_in_pmc_pass": int(mm["count"].iloc[0]), "mfma_flops_per_launch": mops * 512}
            if b is not None:
                bb = b[b.Kernel_Name == name].set_index("Counter_Name")["mean"]
                out[k]["SQ_WAVES"] = float(bb.get("SQ_WAVES", 0))
            hit = st[st.Name.map(short) == k]
            if len(hit):
                out[k]["mean_duration_us"] = float(hit["AverageNs"].iloc[0]) / 1e3
                out[k]["calls_in_trace"] = int(hit["Calls"].iloc[0])
                out[k]["mfma_TFLOPs"] = out[k]["mfma_flops_per_launch"] / (out[k]["mean_duration_us"] * 1e-6) / 1e12
                out[k]["frac_of_dense_bf16_peak"] = out[k]["mfma_TFLOPs"] / PEAK
    tot_flops = sum(v["mfma_flops_per_launch"] * v["launches_in_pmc_pass"] for v in out.values() if "mean_duration_us" in v)
    tot_us = sum(v["mean_duration_us"] * v["launches_in_pmc_pass"] for v in out.values() if "mean_duration_us" in v)
    plain = ""
    pl = P / f"plain_{tag}.log"
    if pl.exists() and pl.read_text().strip():
        plain = pl.read_text().strip().splitlines()[-1]
    res = {"unprofiled_run": plain, "kernels": out}
    if tot_us:
        res["all_mfma_kernels"] = {"TFLOPs": tot_flops / (tot_us * 1e-6) / 1e12, "frac_of_dense_bf16_peak": tot_flops / (tot_us * 1e-6) / 1e12 / PEAK}
    return res


version = (P / "pie_version.txt").read_text().strip() if (P / "pie_version.txt").exists() else "?"
note = ("rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES and --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes, one prompt each), means per launch "
        "over all launches of a kernel (its shapes differ per Linear); durations from the --kernel-trace --stats pass; mfma_flops_per_launch = MOPS x 512; peak 2.5 PFLOP/s")
int4 = {}
for tag in ("128", "512", "4096"):
    r = one(tag, f"# library: {version}; rocprofv3 --kernel-trace --stats -- tools/step_bench --model 8b --prefill {tag} --prefill-reps 8 (3 at 4096)\n")
    if r:
        int4[tag] = r
if int4:
    json.dump({"library": version, "workload": "tools/step_bench --model 8b --prefill N: one N-token prompt through pie_decoder_prefill (8B-shaped int4 g=64, bf16), after the one-off tile repack",
               "note": note, "prompts": int4}, open(OUT / "r05_prefill_mfma.json", "w"), indent=1)
d = one("dense512", f"# library: {version}; rocprofv3 --kernel-trace --stats -- python3.10 scripts/bench_prefill.py --dense --prompts 512 --iterated-max 0 (k_w16l_gemm = the hand-written 16-bit GEMM; k_fill / at::native = the synthetic checkpoint)\n")
if d:
    json.dump({"library": version, "workload": "BASELINE.json configs[2] prefill half: Llama-3-8B-shaped dense bf16, one 512-token prompt (scripts/bench_prefill.py --dense --prompts 512)",
               "gemm": "k_w16l_gemm (w16_gemm.hpp): hand-written; hipBLASLt was removed in round 5 (the round-4 library numbers: git history of this file, 11.67 ms)", "note": note, **d},
              open(OUT / "r05_dense_prefill512.json", "w"), indent=1)
for tag, r in list(int4.items()) + ([("dense512", d)] if d else []):
    print(tag, r.get("unprofiled_run"), r.get("all_mfma_kernels"))
    for k, v in r["kernels"].items():
        print(f"   {k[:60]:60s} {v.get('mean_duration_us', 0):9.1f} us {v.get('mfma_TFLOPs', 0):8.1f} TFLOP/s")
