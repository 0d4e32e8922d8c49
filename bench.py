#!/usr/bin/env python
"""bench.py -- decode tokens/s of Llama-3-8B int4 g=64, batch 1, on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one decode step (one token) through InferenceEngine.generate_step: hipGraph replay of the fused
HIP launch sequence, synthetic random-weight checkpoint (no network), weights and KV cache resident in HBM.
Workload (config.workload): Llama-3-8B-shaped, int4 group 64, bf16 activations, 128-token prompt, greedy.
N > 1: batch-1 decode has no independent units inside one sequence and the 8B model fits one card, so ranks are
independent REPLICAS (one sequence each, no data-path collective): value = sum of tokens over ranks / max time.

Prints ONE JSON line on rank 0 with the driver's contract plus:
  roofline     -- the dominant kernel (gate/up W4S GEMV): algorithmic bytes per launch / mean launch duration,
                  measured here with HIP events on the launch stream, against the 8 TB/s HBM3E peak;
                  `step` = the same fraction for the whole decode step (the north-star figure).
                  `stream_peak` / `frac_of_stream`: the same against a bare streaming read measured in this run (pie_stream_read).
                  `step.floor`: what the launch structure itself costs -- per dependent launch 1.59 us of command-processor time, 0.56 us of
                  workgroup start-up, 0.63 us to publish its output, 1.15 us until the consumer has touched it (the barrier-bit chain of
                  tools/pipeline_probe, profiles/r04_pipeline_probe.txt) -- plus the step's bytes at the stream rate measured in this run;
                  `ms_per_step / floor` says how far the step is from ITS floor, `frac` how far from 8 TB/s.
  tp70b        -- only with --gpus N > 1 on the default workload: after the replicas' timing the same ranks decode ONE Llama-3-70B int4
                  sequence tensor-parallel (BASELINE.json configs[4]), once per communicator backend (IPC one-shot, RCCL); `value` stays
                  the replicas figure.
  repetitions  -- 5 repetitions of the K timed steps (the first is `value`): median and min ms per step.
  cpu_baseline -- the CPU oracle (oracle/pie_oracle.c, OpenMP) timed on this box's host cores on a bounded sample, at the GPU run's
                  context (the same 128-token prompt).
  parity       -- the same oracle steps teacher-forced through the HIP model (full 32 layers, V = 128256), outside the timed
                  region: greedy ids where the oracle's top-2 margin exceeds the bound, max / rms logit error in bf16 ulps of
                  the largest logit; the process exits non-zero when it fails.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--layers", type=int, default=0, help="debug: override num_hidden_layers (invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=48)
    ap.add_argument("--kv-splits", type=int, default=0)
    ap.add_argument("--paged", action="store_true", help="KV cache in 64-token pages of one slab (PagedKVCache) instead of contiguous per-layer buffers")
    ap.add_argument("--model", choices=["8b", "70b", "qv", "3b"], default="8b",
                    help="70b: BASELINE.json configs[4]'s model on ONE card (40 GB int4); qv: configs[3]'s Qwen2-VL-7B text tower; 3b: Llama-3.2-3B; not the metric's workload")
    ap.add_argument("--tp", type=int, default=0, help="tensor parallelism over ALL ranks (== --gpus): ONE sequence decoded by the sharded model "
                    "(BASELINE.json configs[4] with --model 70b); strong scaling, value = that sequence's tokens/s")
    ap.add_argument("--tp-backend", choices=["ipc", "rccl"], default="ipc", help="--tp: the communicator behind the fused step's collectives: the one-shot all-reduce over "
                    "IPC-mapped peer memory (default) or RCCL (ncclAllReduce on the launch stream): the comparator for the first run on a real node")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE", help="developer: one of the library's test / tuning switches (_ffi.KNOBS) for this run")
    ap.add_argument("--bits", type=int, choices=[2, 4, 6, 8], default=4, help="8 / 6 / 2: MLX int8 / int6 / int2 g=64 weights (W8S / W6S / W2S units; a different workload than the metric's)")
    ap.add_argument("--dense", action="store_true", help="BASELINE.json configs[2]: unquantised bf16 weights (a different workload than the metric's)")
    return ap.parse_args()


def _err_eps(got, ref):
    """max |got - ref| and rms(got - ref), in bf16 ulps (2^-8) of the largest / the rms reference logit."""
    eps = 2.0 ** -8
    return (float(np.abs(got - ref).max()) / (eps * float(np.abs(ref).max())),
            float(np.sqrt(np.mean((got - ref) ** 2))) / (eps * float(np.sqrt(np.mean(ref ** 2)))))


def cpu_baseline(cfg, weights_host, n_tokens, model=None, prompt=None, what="Llama-3-8B int4"):
    """Oracle decode tokens/s on the host cores AT THE GPU RUN'S CONTEXT: the benchmark's own prompt (last_only lm_head, untimed), then
    `n_tokens` timed steps, each timed by itself (median and minimum step time reported beside the aggregate).
    With `model` (the HIP model on the same weights) the same steps are then teacher-forced on the GPU and compared:
    the parity gate of the benchmarked configuration itself (full depth, full vocabulary).

    32 layers of random weights amplify every rounding difference (two HIP configurations that differ only in the summation order
    of attention are already 3 ulps apart, scripts/diag_parity_full.py), so the tolerance is not a constant: the same steps also run
    through the oracle in FLOAT32 (same int4 weights, no 16-bit rounding anywhere) and the gate is that the HIP logits are as
    close to that exact-arithmetic result as the bf16 oracle's are (<= 1.5 x its distance, floor 8 ulps), plus identical greedy ids
    wherever the bf16 oracle's top-2 margin exceeds twice its own distance from the float32 result."""
    from oracle import pie_oracle as po

    threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("PIE_CPU_THREADS", "16")))  # the 1-GPU box's CPU share
    po.set_threads(threads)
    orc = po.OracleLlama(cfg, weights_host, "bfloat16")
    cache = [po.OracleKVCache() for _ in orc.layers]
    if prompt is None:
        prompt = np.random.default_rng(1).integers(0, cfg["vocab_size"], 4)
    prompt = np.asarray(prompt, np.int64)
    P = len(prompt)
    t0 = time.perf_counter()
    tok, _ = po.logprobs_argmax(orc.forward(prompt, cache, last_only=True))
    t_prompt = time.perf_counter() - t0
    fed, want, step_s = [], [], []
    for _ in range(n_tokens):
        fed.append(int(tok))
        t0 = time.perf_counter()
        logits = orc.forward(np.array([tok]), cache, last_only=True)
        step_s.append(time.perf_counter() - t0)
        tok, _ = po.logprobs_argmax(logits)
        want.append((np.asarray(logits, np.float32).reshape(-1).copy(), int(tok)))
    dt = float(np.sum(step_s))
    base = {"value": n_tokens / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "median_ms_per_step": 1e3 * float(np.median(step_s)), "min_ms_per_step": 1e3 * float(np.min(step_s)),
            "sample": f"oracle/pie_oracle.c (OpenMP, {threads} threads), same synthetic {what} weights, {P}-token prompt ({t_prompt:.1f} s, untimed), "
                      f"then {n_tokens} greedy decode steps at context {P + 1}..{P + n_tokens} ({dt:.1f} s)"}
    if model is None:
        return base, None
    del orc, cache
    w32 = {k: (v if v.dtype == np.uint32 else po.from_bits(v, "bfloat16")) for k, v in weights_host.items()}
    exact = po.OracleLlama(cfg, w32, "float32")
    ecache = [po.OracleKVCache() for _ in exact.layers]
    exact.forward(prompt, ecache, last_only=True)
    gcache = model.make_cache()
    model.step(torch.from_numpy(prompt).to(torch.int32).cuda(), gcache)  # the batched prompt path (MLX's qmm regime from 6 rows, as the oracle's)
    hip_exact = orc_exact = hip_orc = (0.0, 0.0)
    checked = equal = 0
    for t, (ref, otok) in zip(fed, want):
        truth = np.asarray(exact.forward(np.array([t]), ecache, last_only=True), np.float32).reshape(-1)
        gtok, _, glogits = model.step(torch.tensor([t], dtype=torch.int32, device="cuda"), gcache)
        got = glogits.float().cpu().numpy().reshape(-1)
        hip_exact = tuple(max(a, b) for a, b in zip(hip_exact, _err_eps(got, truth)))
        orc_exact = tuple(max(a, b) for a, b in zip(orc_exact, _err_eps(ref, truth)))
        hip_orc = tuple(max(a, b) for a, b in zip(hip_orc, _err_eps(got, ref)))
        top2 = np.sort(ref)[-2:]
        if top2[1] - top2[0] > 2.0 * float(np.abs(ref - truth).max()):  # the bf16 oracle's own choice is not within its rounding noise
            checked += 1
            equal += int(int(gtok.item()) == otok)
    tol = max(1.5 * orc_exact[0], 8.0)
    parity = {"steps": n_tokens, "ids_checked": checked, "ids_equal_where_margin": equal,
              "max_logit_err_eps": hip_orc[0], "max_rms_err_eps": hip_orc[1],
              "vs_float32_oracle": {"hip_max_eps": hip_exact[0], "hip_rms_eps": hip_exact[1], "bf16_oracle_max_eps": orc_exact[0], "bf16_oracle_rms_eps": orc_exact[1]},
              "tolerance_eps": tol, "tolerance_rule": "HIP vs float32 oracle <= max(1.5 x (bf16 oracle vs float32 oracle), 8) ulps of the largest logit",
              "reference": "oracle/pie_oracle.c (parity unpinned: the reference holds no fixture for this path)",
              "min_ids_checked": min(16, n_tokens // 2),
              "ok": bool(equal == checked and checked >= min(16, n_tokens // 2) and hip_exact[0] <= tol)}
    return base, parity


def tp_shard(full_cfg, tp, rank, dist, backend):
    """This rank's shard of a synthetic tensor-parallel checkpoint, generated directly in the local shapes (the full 70B checkpoint never
    exists anywhere); the replicated tensors (embedding table, norm weights) are rank 0's, broadcast once.  Returns (local config, weights, HipComm)."""
    from proxy_inference_engine_amd.models.utils import synthetic_checkpoint
    from proxy_inference_engine_amd.tp import HipComm, shard_config
    cfg = {k: v for k, v in shard_config(full_cfg, tp).items() if k not in ("tp_world", "tp_vocab_shard")}
    cfg["tie_word_embeddings"] = False
    weights = synthetic_checkpoint(cfg, seed=100 + rank, dtype=torch.bfloat16)
    v_loc = full_cfg["vocab_size"] // tp
    for k in ("weight", "scales", "biases"):
        weights[f"lm_head.{k}"] = weights[f"lm_head.{k}"][:v_loc].contiguous()
    for k, t in weights.items():
        if k.startswith("model.embed_tokens.") or k.endswith("layernorm.weight") or k == "model.norm.weight":
            if dist.get_backend() == "nccl":
                dist.broadcast(t, src=0)
            else:
                h = t.cpu()
                dist.broadcast(h, src=0)
                t.copy_(h)
    return cfg, weights, HipComm(cfg["hidden_size"], backend=backend)


def tp70b_leg(args, rank, world, dist):
    """--gpus N > 1 on the default workload: the first contact with a multi-GPU node should also measure BASELINE.json configs[4] -- ONE
    Llama-3-70B int4 sequence decoded tensor-parallel over all N ranks, once per communicator backend.  Strong scaling: tokens/s of that one
    sequence.  Every failure is recorded, never raised: the replicas line above must survive it."""
    from proxy_inference_engine_amd import InferenceEngine
    from proxy_inference_engine_amd.models.llama import Model, ModelArgs
    from proxy_inference_engine_amd.models.utils import LLAMA3_70B
    out = {"model": "Llama-3-70B-shaped int4 g=64, batch 1", "tp": world, "steps": args.steps, "warmup": args.warmup}
    for backend in ("ipc", "rccl"):
        comm = model = eng = gen = None
        try:
            cfg, weights, comm = tp_shard(dict(LLAMA3_70B), world, rank, dist, backend)
            model = Model(ModelArgs(**cfg), weights, tp=comm)
            del weights
            torch.cuda.empty_cache()
            eng = InferenceEngine(model=model)
            prompt = torch.randint(0, LLAMA3_70B["vocab_size"], (args.prompt,), generator=torch.Generator().manual_seed(1))
            eng.prepare_engine(prompt, temp=0)
            gen = eng.generate_step(prompt)
            for _ in range(1 + args.warmup):
                next(gen)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                next(gen)
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
            elapsed = float(t.item())
            T_mid = args.prompt + 1 + args.warmup + args.steps // 2
            err = comm.status()
            out[backend] = {"tokens_per_s": args.steps / elapsed, "ms_per_step": 1e3 * elapsed / args.steps, "launches_per_step": model.graph_launches(True),
                            "hbm_gbps_per_gpu": model.step_bytes(T_mid, True) * (args.steps / elapsed) / 1e9, "comm_gave_up_at_epoch": err or None}
        except Exception as e:  # noqa: BLE001 -- recorded in the line
            out[backend] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            del eng, gen, model
            if comm is not None:
                try:
                    comm.close()
                except Exception:  # noqa: BLE001
                    pass
            torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(n_dev, 1))  # one rank per GPU; PIE_BENCH_BACKEND=gloo lets ranks share a card (rehearsal)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PIE_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from proxy_inference_engine_amd import InferenceEngine
    from proxy_inference_engine_amd.models.llama import Model, ModelArgs
    from proxy_inference_engine_amd.models.utils import LLAMA3_8B, LLAMA3_70B, LLAMA32_3B, QWEN2VL_7B_TEXT, synthetic_checkpoint

    for kv in args.knob:
        from proxy_inference_engine_amd import _ffi as _knob_ffi
        name, _, value = kv.partition("=")
        _knob_ffi.set_knob(name, int(value))
    cfg = dict({"70b": LLAMA3_70B, "qv": QWEN2VL_7B_TEXT, "3b": LLAMA32_3B}.get(args.model, LLAMA3_8B))
    if args.layers:
        cfg["num_hidden_layers"] = args.layers
    if args.dense:
        cfg["quantization"] = None
    elif args.bits != 4:
        cfg["quantization"] = {"group_size": 64, "bits": args.bits}
    tp = args.tp if args.tp > 1 else 0
    if tp and (tp != world or args.dense or args.bits != 4 or args.paged):
        raise SystemExit("--tp N needs N ranks (--gpus N under torch.distributed.run), int4 weights and contiguous caches")
    comm = None
    if tp:
        full_cfg = cfg
        cfg, weights, comm = tp_shard(full_cfg, tp, rank, dist, args.tp_backend)
    else:
        # heavy-tailed lm_head rows: the parity gate's id comparison needs steps whose greedy token is decided by more than rounding noise
        # (models/utils.py: synthetic_checkpoint); same shapes and bytes, so the timing is that of any Llama-3-8B int4 checkpoint
        weights = synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16, lm_head_tail=1.5)
    # the CPU oracle + the teacher-forced parity gate run for every one-card configuration except the 70B model (its oracle step takes 3 s and
    # 120 GB of host memory twice over); dense weights: 8 oracle steps (15 GB per step at 16 bits + 30 GB in float32)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.model in ("8b", "qv", "3b")
    cpu_tokens = min(args.cpu_tokens, 8) if args.dense else args.cpu_tokens
    weights_host = None
    if want_cpu:  # the oracle reads the same checkpoint, in the reference's on-disk layout, from host memory
        weights_host = {k: (v.cpu().numpy().view(np.uint32) if v.dtype == torch.int32 else v.view(torch.int16).cpu().numpy().view(np.uint16))
                        for k, v in weights.items()}
    model = Model(ModelArgs(**cfg), weights, kv_splits=args.kv_splits, tp=comm)
    del weights
    torch.cuda.empty_cache()

    if args.paged:
        model.enable_paged_kv(num_pages=(args.prompt + args.warmup + args.steps + 2) // 64 + 2)
    eng = InferenceEngine(model=model)
    prompt = torch.randint(0, cfg["vocab_size"], (args.prompt,), generator=torch.Generator().manual_seed(1 if tp else 1 + rank))
    eng.prepare_engine(prompt, temp=0)
    gen = eng.generate_step(prompt)
    next(gen)  # prefill + first token
    for _ in range(args.warmup):
        next(gen)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        next(gen)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    launches_per_step = model.graph_launches(True)  # kernel nodes of the captured step graph (hipGraphGetNodes), not a formula
    # 5 repetitions, median and min.  `value` stays the driver's contract (the K steps just timed); four more repetitions of
    # the same K steps from the same prompt (the prefix cache makes the re-prefill one token) are reported beside it.
    rep_ms = [1e3 * elapsed / args.steps]
    if world == 1 and not args.layers:
        for _ in range(4):
            gen = eng.generate_step(prompt)
            next(gen)
            for _ in range(args.warmup):
                next(gen)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                next(gen)
            torch.cuda.synchronize()
            rep_ms.append(1e3 * (time.perf_counter() - t1) / args.steps)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    tokens_per_s = (1 if tp else world) * args.steps / elapsed  # TP: all ranks decode ONE sequence together
    T_mid = args.prompt + 1 + args.warmup + args.steps // 2  # context length in the middle of the timed region
    step_bytes = model.step_bytes(T_mid, True)
    step_gbps = step_bytes * (args.steps / elapsed) / 1e9  # per GPU

    # dominant kernel (gate/up GEMV: 38 % of the step's bytes), timed alone with HIP events on the launch stream,
    # cycling through the layers so every launch streams weights that are not in the 256 MiB Infinity Cache.
    # Events bracket a whole sweep over the layers (back-to-back launches on the launch stream), so the figure is the
    # kernel's steady-state duration -- what rocprofv3 --kernel-trace reports per dispatch -- not launch gaps.
    reps = 8
    n_l = cfg["num_hidden_layers"]
    for li in range(n_l):
        model.launch_kernel("gate_up", li)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        for li in range(n_l):
            model.launch_kernel("gate_up", li)
        e1.record()
    torch.cuda.synchronize()
    k_ms = float(np.median([a.elapsed_time(b) for a, b in ev])) / n_l
    k_bytes = model.kernel_bytes("gate_up", T_mid)
    k_gbps = k_bytes / (k_ms * 1e-3) / 1e9

    # The measured-stream denominator: a BARE streaming read of 1 GiB shaped like the GEMV's stream (pie_stream_read: one
    # 8-wave workgroup per CU, non-temporal 16-byte loads, nothing computed), timed here with HIP events; best of 5.
    from proxy_inference_engine_amd import _ffi
    sbuf = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    sbuf.random_(0, 255)
    lib = _ffi.load()
    _ffi.check(lib.pie_stream_read(sbuf.data_ptr(), sbuf.numel(), _ffi.stream()))
    torch.cuda.synchronize()
    s_ms = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ffi.check(lib.pie_stream_read(sbuf.data_ptr(), sbuf.numel(), _ffi.stream()))
        e1.record()
        torch.cuda.synchronize()
        s_ms.append(e0.elapsed_time(e1))
    stream_gbps = sbuf.numel() / (min(s_ms) * 1e-3) / 1e9
    del sbuf

    # HBM bytes per launch of the dominant kernel from the PMC passes over the product step (tools/step_bench under rocprofv3 --pmc,
    # scripts/profile_r03.sh, profiles/README.md).  The file names the library build it was measured on; any other build gets null
    # rather than last round's kernels' traffic.
    traffic = None
    tf = ROOT / "profiles" / "r05_traffic.json"
    if tf.exists() and not args.layers and not args.dense and args.model == "8b" and args.bits == 4:
        rec = json.loads(tf.read_text())
        if rec.get("library") == _ffi.load().pie_version().decode():
            traffic = rec.get("hbm_bytes_per_launch")

    # The floor of the launch structure: a synthetic chain of the step's 161 dependent launches, written as raw AQL packets with
    # the barrier bit and agent fences HIP emits, costs per launch 1.59 us (empty dispatch) + 0.56 (8-wave workgroups, LDS, one barrier) + 0.63
    # (publish the output) + 1.15 (the consumer's first touch of it) = 3.93 us before any weight is streamed, and 1.150 ms with the 8B model's
    # 4.24 GB behind it (tools/pipeline_probe.cpp; profiles/r04_pipeline_probe.txt, "serial: barrier bit, agent fences").  The serial sum below
    # assumes no overlap between those per-launch costs and the stream; the probe's chain overlaps the stream's head with the staging.
    # (The probe chain is the FIVE-launches-per-layer structure, 162 launches.  Since round 5 the 32 / 8 / 128 head geometry runs four per layer -- the
    # attention behind an XCD-local seam of the q|k|v launch -- so `launches` is 130 there and the step may come out below the 162-launch chain.)
    per_launch_us = {"dispatch": 1.59, "workgroup_start": 0.56, "publish": 0.63, "first_touch": 1.15}
    n_launch = launches_per_step if launches_per_step and launches_per_step > 0 else 5 * n_l + 2
    stream_ms = step_bytes / (stream_gbps * 1e9) * 1e3
    launch_ms = n_launch * sum(per_launch_us.values()) * 1e-3
    floor = {"launches": n_launch, "per_launch_us": per_launch_us, "launch_ms": launch_ms, "stream_ms": stream_ms, "serial_sum_ms": launch_ms + stream_ms,
             "probe_chain_ms_8b": 1.150, "probe_chain_launches": 162, "source": "tools/pipeline_probe.cpp, profiles/r04_pipeline_probe.txt (raw-AQL chain of the step's shape); stream term = step bytes / stream_peak of this run",
             "ms_per_step_over_serial_sum": (1e3 * elapsed / args.steps) / (launch_ms + stream_ms)}
    if args.model == "8b" and args.bits == 4 and not args.dense and not args.layers:
        floor["ms_per_step_over_probe_chain"] = (1e3 * elapsed / args.steps) / 1.150

    shape_cfg = full_cfg if tp else cfg
    out = {
        "metric": (f"decode tokens/sec, Llama-3-{args.model.upper()} int4 g=64 batch=1, TP={tp}; achieved HBM GB/s per GPU" if tp else
                   "decode tokens/sec, Llama-3-8B int4 g=64 batch=1; achieved HBM GB/s" if (args.model == "8b" and args.bits == 4 and not args.dense) else
                   f"decode tokens/sec, {'Qwen2-VL-7B text tower' if args.model == 'qv' else 'Llama-3-' + args.model.upper()} "
                   f"{'dense bf16' if args.dense else f'int{args.bits} g=64'} batch=1; achieved HBM GB/s (NOT the BASELINE.json metric's workload)"),
        "value": tokens_per_s, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if tp else "weak", "vs_baseline": None,
        "dtype": "bf16", "dtype_detail": ("bf16 weights" if args.dense else f"uint{args.bits} g=64 weights") + " x bf16 activations, fp32 accumulate (v_dot2c_f32_bf16)", "data": "synthetic",
        "config": {"workload": f"{'Qwen2-VL-7B text tower' if args.model == 'qv' else 'Llama-3-' + args.model.upper()}-shaped (H{shape_cfg['hidden_size']} L{n_l} {shape_cfg['num_attention_heads']}/{shape_cfg['num_key_value_heads']} heads I{shape_cfg['intermediate_size']} V{shape_cfg['vocab_size']}) {'dense bf16' if args.dense else f'int{args.bits} g=64'} greedy decode, batch 1, "
                               f"{args.prompt}-token prompt, context {args.prompt + 1 + args.warmup}..{args.prompt + 1 + args.warmup + args.steps}",
                   "parallelism": f"TP={tp} ({args.tp_backend} collectives)" if tp else ("replicas" if world > 1 else "single GPU"),
                   "launches_per_step": launches_per_step,  # kernel nodes of the captured hipGraph
                   "hipgraph": True, "kv": "paged (64-token pages)" if args.paged else "contiguous"},
        "roofline": {"bound": "hbm", "kernel": "k_w4s_gemv<bf16, rmsnorm, swiglu> (gate/up)", "achieved": k_gbps, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": k_gbps / HBM_PEAK_GBPS, "traffic": traffic, "bytes_per_launch": k_bytes,
                     "ms_per_launch": k_ms,
                     "stream_peak": stream_gbps, "frac_of_stream": k_gbps / stream_gbps,  # measured in this run: pie_stream_read over 1 GiB
                     "step": {"achieved": step_gbps, "frac": step_gbps / HBM_PEAK_GBPS, "frac_of_stream": step_gbps / stream_gbps, "bytes_per_step": step_bytes,
                              "floor": floor}},
        "repetitions": {"ms_per_step": rep_ms, "median_ms_per_step": float(np.median(rep_ms)), "min_ms_per_step": float(np.min(rep_ms))},
    }
    if world > 1 and not tp and args.model == "8b" and args.bits == 4 and not args.dense and not args.layers and not args.paged \
            and os.environ.get("PIE_BENCH_TP70B", "1") != "0":
        del eng, gen, model
        eng = gen = model = None
        torch.cuda.empty_cache()
        out["tp70b"] = tp70b_leg(args, rank, world, dist)
    if rank == 0:
        if want_cpu:
            del eng, gen
            try:
                what = ("Qwen2-VL-7B text tower" if args.model == "qv" else "Llama-3-" + args.model.upper()) + (" dense bf16" if args.dense else f" int{args.bits}")
                out["cpu_baseline"], out["parity"] = cpu_baseline(cfg, weights_host, cpu_tokens, model, prompt=prompt.numpy(), what=what)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
                out["parity"] = {"ok": False, "error": str(e)}
        else:
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port",
                                   "sample": "skipped (timed on rank 0 at N=1 only)"}
        print(json.dumps(out), flush=True)
    if comm is not None:
        err = comm.status()
        del eng, gen, model
        comm.close()
        if err:
            raise SystemExit(f"tensor-parallel communicator gave up waiting for a peer (epoch {err})")
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and out.get("parity") is not None and not out["parity"].get("ok", False):
        raise SystemExit("parity gate of the benchmarked model FAILED: " + json.dumps(out["parity"]))


if __name__ == "__main__":
    main()
