"""Tensor-parallel sharding plan for Llama-shaped int4 checkpoints -- host side.

The reference has no parallelism of any kind; `north_star` asks for Megatron-style TP with RCCL
all-reduce over xGMI "only for models that do not fit one 288 GB card".  The measured 8B workload does not shard
(replicas only); this module is the sharder + communicator that the device TP step will sit on:

  column-parallel  q, k, v (by heads), gate, up (by rows)      -> no communication, local attention heads
  row-parallel     o_proj, down_proj (K split on multiples of the 64-wide quantisation group)
                   -> fp32 partial sums, ONE all-reduce(sum) of [1, H] each, then the single rounding to T and the
                      residual add: the same rounding points as the unsharded graph (language.py:108,127,151,153)
  vocab-parallel   lm_head (rows) -> per-rank (max, sum exp, argmax) triples, all-gather of 3 numbers per rank
  replicated       norms, embedding table (row gather)

Works on torch tensors on any device (the gloo test runs it on CPU).  `.weight` are MLX uint32 code words carried
in int32 tensors [N, K/8]; `.scales` / `.biases` [N, K/64] (models/utils.py:96-111 of the reference).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

GROUP = 64  # quantisation group size


def _rows(t: dict, name: str, r0: int, r1: int) -> dict:
    return {f"{name}.{k}": t[f"{name}.{k}"][r0:r1].contiguous() for k in ("weight", "scales", "biases")}


def _cols(t: dict, name: str, k0: int, k1: int) -> dict:
    if k0 % GROUP or k1 % GROUP:
        raise ValueError(f"{name}: row-parallel split [{k0}, {k1}) is not aligned to the {GROUP}-wide quantisation group")
    return {f"{name}.weight": t[f"{name}.weight"][:, k0 // 8:k1 // 8].contiguous(),
            f"{name}.scales": t[f"{name}.scales"][:, k0 // GROUP:k1 // GROUP].contiguous(),
            f"{name}.biases": t[f"{name}.biases"][:, k0 // GROUP:k1 // GROUP].contiguous()}


def shard_config(config: dict, world: int) -> dict:
    nh = config["num_attention_heads"]
    nkv = config.get("num_key_value_heads") or nh
    D = config.get("head_dim") or config["hidden_size"] // nh
    I, V = config["intermediate_size"], config["vocab_size"]
    if nh % world or nkv % world:
        raise ValueError(f"TP={world} must divide the head counts ({nh} q / {nkv} kv)")
    if (I // world) % GROUP or I % world or (nh // world * D) % GROUP:
        raise ValueError(f"TP={world}: intermediate_size/world and local q width must be multiples of {GROUP}")
    if V % (2 * world):
        raise ValueError(f"TP={world}: vocab_size must be divisible by 2*world")
    c = dict(config)
    c.update(num_attention_heads=nh // world, num_key_value_heads=nkv // world, head_dim=D,
             intermediate_size=I // world, tp_world=world, tp_vocab_shard=V // world)
    return c


def shard_checkpoint(weights: dict, config: dict, rank: int, world: int) -> tuple[dict, dict]:
    """Returns (this rank's checkpoint in the same key layout, the local config)."""
    local = shard_config(config, world)
    nh = config["num_attention_heads"]
    nkv = config.get("num_key_value_heads") or nh
    D = local["head_dim"]
    I, V = config["intermediate_size"], config["vocab_size"]
    qw, kvw, iw, vw = nh // world * D, nkv // world * D, I // world, V // world
    out: dict = {}
    for i in range(config["num_hidden_layers"]):
        p = f"model.layers.{i}"
        out[f"{p}.input_layernorm.weight"] = weights[f"{p}.input_layernorm.weight"]
        out[f"{p}.post_attention_layernorm.weight"] = weights[f"{p}.post_attention_layernorm.weight"]
        out.update(_rows(weights, f"{p}.self_attn.q_proj", rank * qw, (rank + 1) * qw))
        out.update(_rows(weights, f"{p}.self_attn.k_proj", rank * kvw, (rank + 1) * kvw))
        out.update(_rows(weights, f"{p}.self_attn.v_proj", rank * kvw, (rank + 1) * kvw))
        out.update(_cols(weights, f"{p}.self_attn.o_proj", rank * qw, (rank + 1) * qw))
        out.update(_rows(weights, f"{p}.mlp.gate_proj", rank * iw, (rank + 1) * iw))
        out.update(_rows(weights, f"{p}.mlp.up_proj", rank * iw, (rank + 1) * iw))
        out.update(_cols(weights, f"{p}.mlp.down_proj", rank * iw, (rank + 1) * iw))
    for k in ("weight", "scales", "biases"):
        out[f"model.embed_tokens.{k}"] = weights[f"model.embed_tokens.{k}"]
    out["model.norm.weight"] = weights["model.norm.weight"]
    head = "model.embed_tokens" if config.get("tie_word_embeddings", True) else "lm_head"
    out.update({k.replace(head, "lm_head"): v for k, v in _rows(weights, head, rank * vw, (rank + 1) * vw).items()})
    return out, local


class TPGroup:
    """The two collectives of the TP decode step over torch.distributed (backend "nccl" = RCCL over xGMI on the
    GPUs; "gloo" in the CPU tests).  Messages are tiny ([1, H] fp32 = 16-32 KiB): latency-bound, one-shot."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_reduce_partial(self, partial_f32: torch.Tensor) -> torch.Tensor:
        """Sum of the row-parallel fp32 partials; the caller rounds to T once afterwards."""
        dist.all_reduce(partial_f32, op=dist.ReduceOp.SUM, group=self.group)
        return partial_f32

    def merge_logit_stats(self, local_max: float, local_sumexp: float, local_argmax: int, vocab_offset: int):
        """Vocab-parallel tail: (max, sum exp(x - max), first argmax) of every shard -> global (lse, token)."""
        mine = torch.tensor([local_max, local_sumexp, float(local_argmax + vocab_offset)], dtype=torch.float64)
        allv = [torch.zeros(3, dtype=torch.float64) for _ in range(self.world)]
        dist.all_gather(allv, mine, group=self.group)
        stats = torch.stack(allv)
        M = stats[:, 0].max()
        lse = M + torch.log((stats[:, 1] * torch.exp(stats[:, 0] - M)).sum())
        cand = stats[stats[:, 0] == M]
        return float(lse), int(cand[:, 2].min().item())  # ties: the lowest vocabulary index wins (mx.argmax)
