"""gpurun_out/r4/prof_prefill (scripts/profile_r04_prefill.sh) -> profiles/r04_prefill_mfma.json + r04_prefill4096_kernel_stats.csv."""
import glob
import json
from pathlib import Path

import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "gpurun_out" / "r4" / "prof_prefill"
OUT = ROOT / "profiles"


def counters(d):
    df = pd.read_csv(glob.glob(str(P / d / "*" / "*_counter_collection.csv"))[0])
    return df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


version = (P / "pie_version.txt").read_text().strip()
stats = glob.glob(str(P / "prefill_stats" / "*" / "*_kernel_stats.csv"))[0]
(OUT / "r04_prefill4096_kernel_stats.csv").write_text(f"# library: {version}; rocprofv3 --kernel-trace --stats -- tools/step_bench --model 8b --prefill 4096 --prefill-reps 3\n" + Path(stats).read_text())
st = pd.read_csv(stats)
m, b = counters("prefill_mfma"), counters("prefill_busy")
out = {}
for name in sorted(set(m.Kernel_Name)):
    mm = m[(m.Kernel_Name == name) & (m.Counter_Name == "SQ_INSTS_VALU_MFMA_MOPS_BF16")]
    if mm.empty or mm["mean"].iloc[0] == 0:
        continue
    bb = b[b.Kernel_Name == name].set_index("Counter_Name")["mean"]
    mops = float(mm["mean"].iloc[0])
    k = short(name)
    out[k] = {"launches": int(mm["count"].iloc[0]), "SQ_INSTS_VALU_MFMA_MOPS_BF16": mops, "mfma_flops_per_launch": mops * 512,
              "SQ_BUSY_CYCLES": float(bb.get("SQ_BUSY_CYCLES", 0)), "SQ_WAVES": float(bb.get("SQ_WAVES", 0))}
    hit = st[st.Name.map(short) == k]
    if len(hit):
        out[k]["mean_duration_us"] = float(hit["AverageNs"].iloc[0]) / 1e3
        out[k]["mfma_TFLOPs"] = out[k]["mfma_flops_per_launch"] / (out[k]["mean_duration_us"] * 1e-6) / 1e12
        out[k]["frac_of_dense_bf16_peak_2500"] = out[k]["mfma_TFLOPs"] / 2500.0
tot_flops = sum(v["mfma_flops_per_launch"] * v["launches"] for v in out.values() if "mean_duration_us" in v)
tot_us = sum(v["mean_duration_us"] * v["launches"] for v in out.values() if "mean_duration_us" in v)
plain = (P / "plain.log").read_text().strip().splitlines()[-1] if (P / "plain.log").exists() else ""
json.dump({"library": version, "workload": "tools/step_bench --model 8b --prefill 4096 (one 4096-token prompt through pie_decoder_prefill, after the one-off tile repack)",
           "unprofiled_run": plain,
           "note": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES and --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes), means per launch over all "
                   "launches of a kernel (its shapes differ per Linear); durations from the kernel-trace pass.  mfma_flops_per_launch = MOPS x 512.",
           "all_mfma_kernels": {"TFLOPs": tot_flops / (tot_us * 1e-6) / 1e12, "frac_of_dense_bf16_peak_2500": tot_flops / (tot_us * 1e-6) / 1e12 / 2500.0},
           "kernels": out}, open(OUT / "r04_prefill_mfma.json", "w"), indent=1)
for k, v in out.items():
    print(f"{k:50s} x{v['launches']:4d} {v.get('mean_duration_us', 0):9.1f} us {v.get('mfma_TFLOPs', 0):8.1f} TFLOP/s")
print("all:", round(tot_flops / (tot_us * 1e-6) / 1e12, 1), "TFLOP/s;", plain)
