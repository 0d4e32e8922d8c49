"""CPU, world_size 2 over gloo: the tensor-parallel sharding plan (proxy_inference_engine_amd/tp.py) reproduces the
unsharded graph.  Each rank shards the same synthetic int4 checkpoint, runs one transformer block + the vocab-parallel
tail on ITS shard with the oracle's ops, exchanges the two row-parallel fp32 partials with all-reduce, and compares
with the unsharded oracle block.  (The reference has no distributed path: this is the build's own N > 1 coverage.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pie_oracle as po
from tests._util import assert_bits_close

CFG = {"model_type": "llama", "hidden_size": 256, "num_hidden_layers": 1, "intermediate_size": 512,
       "num_attention_heads": 4, "num_key_value_heads": 2, "rms_norm_eps": 1e-5, "vocab_size": 512,
       "rope_theta": 10000.0, "max_position_embeddings": 2048, "tie_word_embeddings": False,
       "quantization": {"group_size": 64, "bits": 4}}
DT = "bfloat16"


def _to_torch(w):
    return {k: torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v.view(np.int16)) for k, v in w.items()}


def _to_np(w):
    return {k: (v.numpy().view(np.uint32) if v.dtype == torch.int32 else v.numpy().view(np.uint16)) for k, v in w.items()}


def _f32(bits):  # T storage bits -> fp32 values (for un-rounded partial sums)
    return po.from_bits(bits, DT)


def _block(w, cfg, x, offset, kc, vc, reduce_fn):
    """One TransformerBlock (language.py:144-154) on a (possibly sharded) checkpoint; reduce_fn sums row-parallel partials."""
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or cfg["hidden_size"] // nh
    p = "model.layers.0"
    freqs = po.llama3_rope_freqs(D, cfg["rope_theta"], cfg["max_position_embeddings"])
    lin = lambda name, inp, dt=DT: po.quantized_matmul(inp, w[f"{p}.{name}.weight"], w[f"{p}.{name}.scales"] if dt == DT else _f32(w[f"{p}.{name}.scales"]),
                                                        w[f"{p}.{name}.biases"] if dt == DT else _f32(w[f"{p}.{name}.biases"]), dtype=dt)
    xn = po.rms_norm(x, w[f"{p}.input_layernorm.weight"], cfg["rms_norm_eps"], DT)
    q = po.rope(lin("self_attn.q_proj", xn).reshape(nh, 1, D), freqs, offset, DT)
    k = po.rope(lin("self_attn.k_proj", xn).reshape(nkv, 1, D), freqs, offset, DT)
    v = lin("self_attn.v_proj", xn).reshape(nkv, 1, D)
    kc[:, offset], vc[:, offset] = k[:, 0], v[:, 0]
    att = po.sdpa(q, kc, vc, D ** -0.5, None, DT, True, T=offset + 1).reshape(1, nh * D)
    r = po.round_T(reduce_fn(lin("self_attn.o_proj", att, "float32")), DT)          # fp32 partial -> sum -> ONE rounding
    h = po.add(x, r, DT)
    hn = po.rms_norm(h, w[f"{p}.post_attention_layernorm.weight"], cfg["rms_norm_eps"], DT)
    act = po.silu_mul(lin("mlp.gate_proj", hn), lin("mlp.up_proj", hn), DT)
    r = po.round_T(reduce_fn(lin("mlp.down_proj", act, "float32")), DT)
    return po.add(h, r, DT)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from proxy_inference_engine_amd import tp
        full = po.synth_checkpoint(CFG, seed=3, dtype=DT, lm_head_gain=8.0)
        shard_t, local = tp.shard_checkpoint(_to_torch(full), CFG, rank, world)
        shard = _to_np(shard_t)
        grp = tp.TPGroup()
        rng = np.random.default_rng(0)
        x = po.round_T(rng.standard_normal((1, CFG["hidden_size"])), DT)
        D = local["head_dim"]
        # a few cached positions so attention is non-trivial; every rank holds ITS kv heads
        kfull = po.round_T(rng.standard_normal((CFG["num_key_value_heads"], 8, D)), DT)
        vfull = po.round_T(rng.standard_normal((CFG["num_key_value_heads"], 8, D)), DT)
        nkv_l = local["num_key_value_heads"]
        kc, vc = kfull[rank * nkv_l:(rank + 1) * nkv_l].copy(), vfull[rank * nkv_l:(rank + 1) * nkv_l].copy()

        def reduce_fn(partial):
            t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.float32))
            return grp.all_reduce_partial(t).numpy()

        h_tp = _block(shard, local, x, 5, kc, vc, reduce_fn)
        h_ref = _block(full, CFG, x, 5, kfull.copy(), vfull.copy(), lambda p: p)
        assert_bits_close(po.to_bits(h_tp, DT), po.to_bits(h_ref, DT), max_ulp=1, max_frac=0.05, what=f"rank {rank} block output")
        # vocab-parallel tail
        hn = po.rms_norm(h_ref, full["model.norm.weight"], CFG["rms_norm_eps"], DT)
        logits = po.quantized_matmul(hn, full["lm_head.weight"], full["lm_head.scales"], full["lm_head.biases"], dtype=DT)[0]
        tok_ref, lp_ref = po.logprobs_argmax(logits)
        mine = po.quantized_matmul(hn, shard["lm_head.weight"], shard["lm_head.scales"], shard["lm_head.biases"], dtype=DT)[0]
        vs = local["tp_vocab_shard"]
        assert np.array_equal(mine, logits[rank * vs:(rank + 1) * vs])
        m = float(mine.max())
        lse, tok = grp.merge_logit_stats(m, float(np.exp(mine.astype(np.float64) - m).sum()), int(mine.argmax()), rank * vs)
        assert tok == tok_ref and abs((logits[tok] - lse) - lp_ref[tok]) < 1e-4
        ret[rank] = "ok"
    except Exception as e:  # surface the failure in the parent
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


def test_tp2_block_and_tail_match_unsharded():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_shard_plan_rejects_misaligned_splits():
    from proxy_inference_engine_amd import tp
    with pytest.raises(ValueError):
        tp.shard_config(dict(CFG, intermediate_size=576), 2)      # 288 per rank is not a multiple of the 64-wide group
    with pytest.raises(ValueError):
        tp.shard_config(CFG, 4)                                   # 2 kv heads cannot be split 4 ways
    c = tp.shard_config(CFG, 2)
    assert (c["num_attention_heads"], c["num_key_value_heads"], c["intermediate_size"], c["tp_vocab_shard"]) == (2, 1, 256, 256)


def _fused_tail_f32(stats):
    """Restatement of k_tp_tail_stats' merge (csrc/tp_comm.hip): fp32 arithmetic, ranks visited in order."""
    gm = np.array([s[0] for s in stats], dtype=np.float32)
    GM = gm.max()
    S = np.float32(0.0)
    tok = 2 ** 31 - 1
    for m, se, arg in stats:
        S = np.float32(S + np.float32(se) * np.exp(np.float32(m) - GM, dtype=np.float32))
        if np.float32(m) == GM:
            tok = min(tok, int(arg))
    return float(GM + np.log(S, dtype=np.float32)), tok


def _tail_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from proxy_inference_engine_amd import tp
        grp = tp.TPGroup()
        rng = np.random.default_rng(11)
        logits = po.round_T(rng.standard_normal(CFG["vocab_size"]) * 3.0, DT).astype(np.float32)
        logits[[17, 300]] = logits.max() + 1.0          # the maximum appears on BOTH shards: the lowest vocabulary index must win
        vs = CFG["vocab_size"] // world
        mine = logits[rank * vs:(rank + 1) * vs]
        m = float(mine.max())
        triple = (m, float(np.exp(mine - np.float32(m), dtype=np.float32).sum(dtype=np.float32)), int(mine.argmax()) + rank * vs)
        lse64, tok64 = grp.merge_logit_stats(triple[0], triple[1], triple[2] - rank * vs, rank * vs)
        gathered = [None] * world
        dist.all_gather_object(gathered, triple)       # what the device kernel's push / pull delivers: every rank's triple, rank order
        lse32, tok32 = _fused_tail_f32(gathered)
        assert tok32 == tok64 == 17, (tok32, tok64)
        assert abs(lse32 - lse64) <= 4e-6 * max(1.0, abs(lse64))
        ret[rank] = "ok"
    except Exception as e:
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


def test_tp2_fused_tail_merge_matches_functional_merge():
    """The fused step's vocabulary-parallel tail merges (max, sum exp, argmax) triples in fp32 on the device; same token (ties:
    lowest index across shards) and log-sum-exp as TPGroup.merge_logit_stats' float64 host merge."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_tail_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_70b_shards_fit_the_fused_decoder(world):
    """BASELINE.json configs[4]: every TP degree of Llama-3-70B yields a shard the fused decoder accepts (pie_decoder_create's shape
    rules, the o_proj attention-merge prologue's K limit, the per-wave run limit of the GEMV, the communicator's slot size)."""
    from proxy_inference_engine_amd import tp
    from proxy_inference_engine_amd.models.utils import LLAMA3_70B
    c = tp.shard_config(LLAMA3_70B, world)
    nh, nkv, D, I, H = c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"], c["intermediate_size"], c["hidden_size"]
    v_loc = c["tp_vocab_shard"]
    assert (nh, nkv, I, v_loc) == (64 // world, 8 // world, 28672 // world, 128256 // world)
    assert H % 64 == 0 and I % 64 == 0 and D in (64, 128) and nh % nkv == 0 and v_loc % 2 == 0
    assert nh * D <= 2 * 8 * 8 * 64          # PRO_ATTN stages the merged attention vector with <= 2 pieces per thread
    assert H <= 1 << 20                       # pie_comm_create's max_elems bound
    for n_rows in (nh * D + 2 * nkv * D, H, 2 * I, v_loc):
        assert (n_rows // 2 + 2047) // 2048 <= 64, n_rows   # GEMV_MAX_RUN row pairs per wave on 2048 waves
