"""Aggregate decode throughput of the multi-sequence step (pie_decoder_step_batch) on the 8B int4 model: B sequences share one
pass over the weights.  Prints ms per step and total tokens/s per batch size, next to the single-sequence graph-replayed step.

    python scripts/bench_batch.py [--batches 1,2,4,8,16,32,64] [--prompt 128] [--steps 32] [--mixed 64,512]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from proxy_inference_engine_amd.models.llama import Model, ModelArgs  # noqa: E402
from proxy_inference_engine_amd.models.utils import LLAMA3_8B, synthetic_checkpoint  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,4,8,16,32,64")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--mixed", default="", help="P1,P2,...: per batch size, one fresh prompt of P tokens riding the decode step (step_mixed) vs a prompt pass + a step")
    ap.add_argument("--kv-int8", action="store_true", help="a second pass with the sequences on int8 pages (per-head scales 1/16; prompts through prefill_batch)")
    args = ap.parse_args()
    cfg = dict(LLAMA3_8B)
    if args.layers:
        cfg["num_hidden_layers"] = args.layers
    model = Model(ModelArgs(**cfg), synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16))
    torch.cuda.empty_cache()
    batches = [int(b) for b in args.batches.split(",")]
    n_mixed = len([x for x in args.mixed.split(",") if x])
    pages_per_seq = (args.prompt + 8 + 2 * args.steps + n_mixed * 2 * (max(4, args.steps // 4) + 2) + 63) // 64 + 1
    mixed = [int(x) for x in args.mixed.split(",") if x]
    model.enable_paged_kv(num_pages=max(batches) * pages_per_seq + 4 + (max(mixed) // 64 + 2 if mixed else 0), max_blocks=pages_per_seq)
    # single-sequence baseline: the graph-replayed decode step on a paged cache
    c0 = model.make_cache()
    g = torch.Generator().manual_seed(1)
    model.step(torch.randint(0, cfg["vocab_size"], (args.prompt,), generator=g).cuda(), c0)
    for _ in range(8):
        model.step(None, c0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.step(None, c0)
    torch.cuda.synchronize()
    single = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"mode": "single sequence, hipGraph step", "ms_per_step": round(single * 1e3, 3), "tokens_per_s": round(1 / single, 1)}), flush=True)
    c0[0].page_manager.release()
    for B in batches:
        caches = []
        toks = []
        for i in range(B):
            c = model.make_cache()
            tok, _, _ = model.step(torch.randint(0, cfg["vocab_size"], (args.prompt + (i % 7),), generator=g).cuda(), c)
            caches.append(c)
            toks.append(tok)
        tokens = torch.cat(toks)
        for _ in range(4):
            tokens, _, _ = model.step_batch(tokens, caches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tokens, _, _ = model.step_batch(tokens, caches)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"mode": "step_batch", "sequences": B, "ms_per_step": round(dt * 1e3, 3), "tokens_per_s": round(B / dt, 1),
                          "vs_single": round(B / dt * single, 2)}), flush=True)
        for P in mixed:  # a prompt of P tokens arrives while B sequences decode: its rows in THEIR pass, or a pass of its own first
            prompt = torch.randint(0, cfg["vocab_size"], (P,), generator=g).tolist()
            reps = max(4, args.steps // 4)

            def timed(fn):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps

            def one_pass():
                pc = model.make_cache()
                model.step_mixed(tokens, caches, [prompt], [pc])
                pc[0].page_manager.release()

            def two_passes():
                pc = model.make_cache()
                model.prefill_batch([prompt], [pc])
                model.step_batch(tokens, caches)
                pc[0].page_manager.release()

            t_two, t_one = timed(two_passes), timed(one_pass)
            print(json.dumps({"mode": "mixed pass", "sequences": B, "prompt_rows": P, "ms_step_mixed": round(t_one * 1e3, 3),
                              "ms_prompt_pass_plus_step": round(t_two * 1e3, 3), "ratio": round(t_one / t_two, 3)}), flush=True)
        for c in caches:
            c[0].page_manager.release()
    if args.kv_int8:  # the same step on the reference page's own storage: int8 rows + per-head fp16 scales (half the cache bytes)
        L, Hkv = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
        sc = torch.full((L, Hkv), 1 / 16, dtype=torch.float16)
        model.enable_paged_kv(num_pages=max(batches) * pages_per_seq + 4, max_blocks=pages_per_seq, kv_dtype=torch.int8, kv_scales=(sc, sc))
        for B in batches:
            caches = [model.make_cache() for _ in range(B)]
            prompts = [torch.randint(0, cfg["vocab_size"], (args.prompt + (i % 7),), generator=g).tolist() for i in range(B)]
            tokens, _, _ = model.prefill_batch(prompts, caches)
            tokens = tokens.clone()
            for _ in range(4):
                tokens, _, _ = model.step_batch(tokens, caches)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tokens, _, _ = model.step_batch(tokens, caches)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            print(json.dumps({"mode": "step_batch on int8 pages", "sequences": B, "ms_per_step": round(dt * 1e3, 3), "tokens_per_s": round(B / dt, 1)}), flush=True)
            for c in caches:
                c[0].page_manager.release()


if __name__ == "__main__" and "--engine" not in sys.argv:
    main()


def engine_mode():
    """End-to-end continuous batching: R requests (prompt P, G new tokens each) through BatchedEngine with at most S in flight."""
    import argparse as _a
    ap = _a.ArgumentParser()
    ap.add_argument("--engine", action="store_true")
    ap.add_argument("--requests", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--slots", type=int, default=32)
    ap.add_argument("--chunk", type=int, default=0, help="BatchedEngine(prefill_chunk=N): chunked prefill")
    ap.add_argument("--shared", type=int, default=0, help="the first N tokens of every prompt are the same (a system prompt); with --share-prefix its pages are computed once")
    ap.add_argument("--share-prefix", action="store_true")
    ap.add_argument("--no-mixed", action="store_true")
    args = ap.parse_args()
    from proxy_inference_engine_amd.engine import BatchedEngine
    cfg = dict(LLAMA3_8B)
    model = Model(ModelArgs(**cfg), synthetic_checkpoint(cfg, seed=0, dtype=torch.bfloat16))
    torch.cuda.empty_cache()
    pages = args.slots * ((args.prompt + args.new + 63) // 64 + 1) + 4
    eng = BatchedEngine(model, num_pages=pages, max_batch=args.slots, prefill_chunk=args.chunk or None, share_prefix=args.share_prefix, mixed=not args.no_mixed)
    g = torch.Generator().manual_seed(3)
    system = torch.randint(0, cfg["vocab_size"], (args.shared,), generator=g).tolist()
    prompts = [system + torch.randint(0, cfg["vocab_size"], (args.prompt - args.shared + (i % 5),), generator=g).tolist() for i in range(args.requests)]
    eng.generate(prompts[:args.slots], 4)                       # warm-up: scratch, tile copies, hipBLASLt plans
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.generate(prompts, args.new)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_new = sum(len(o) for o in out)
    print(json.dumps({"mode": "BatchedEngine", "requests": args.requests, "slots": args.slots, "prompt": args.prompt, "new_tokens": n_new,
                      "seconds": round(dt, 3), "generated_tokens_per_s": round(n_new / dt, 1),
                      "prompt_plus_generated_tokens_per_s": round((n_new + sum(len(p) for p in prompts)) / dt, 1), "batched_steps": eng.steps,
                      "mixed_passes": eng.mixed_passes, "shared_pages": eng.shared_pages, "prefill_chunk": args.chunk, "mixed": not args.no_mixed}))


if __name__ == "__main__" and "--engine" in sys.argv:
    engine_mode()
