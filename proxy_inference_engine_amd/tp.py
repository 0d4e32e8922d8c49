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
    q = config.get("quantization") or {}
    if q.get("group_size", GROUP) != GROUP or q.get("bits", 4) != 4:
        raise ValueError(f"tensor-parallel sharding is written for int4 group-{GROUP} checkpoints (the K split of o_proj / down_proj slices "
                         f"code words and groups); got quantization = {dict(q)}")
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


class HipComm:
    """The native communicator of the fused tensor-parallel step (include/pie_hip.h, pie_comm_*), enqueued on HIP streams and capturable
    in the decoder's hipGraph.  backend "ipc" (default): the one-shot all-reduce over IPC-mapped peer buffers, its push half in the
    row-parallel GEMV's epilogue; torch.distributed is only the side channel that carries the 64-byte IPC handles at start-up (any
    backend; gloo in the one-card test).  backend "rccl": the same collectives through RCCL (ncclAllReduce / ncclAllGather on the launch
    stream) -- the comparator and fall-back on a real multi-GPU node; the side channel carries RCCL's 128-byte unique id."""

    def __init__(self, max_elems: int, group=None, backend: str = "ipc"):
        import ctypes as C
        from . import _ffi
        _ffi.require_gpu()
        if backend not in ("ipc", "rccl"):
            raise ValueError("HipComm: backend must be 'ipc' or 'rccl'")
        self._lib = _ffi.load()
        self.group = group
        self.backend = backend
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.handle = C.c_void_p()
        if backend == "rccl":
            uid = C.create_string_buffer(128)
            if self.rank == 0:
                _ffi.check(self._lib.pie_comm_rccl_unique_id(uid))
            if self.world > 1:
                box = [uid.raw]
                dist.broadcast_object_list(box, src=0, group=group)
                uid = C.create_string_buffer(box[0], 128)
            _ffi.check(self._lib.pie_comm_create_rccl(self.rank, self.world, int(max_elems), uid, C.byref(self.handle)))
            return
        _ffi.check(self._lib.pie_comm_create(self.rank, self.world, int(max_elems), C.byref(self.handle)))
        if self.world > 1:
            mine = C.create_string_buffer(64)
            _ffi.check(self._lib.pie_comm_export(self.handle, mine))
            handles: list = [None] * self.world
            dist.all_gather_object(handles, mine.raw, group=group)
            _ffi.check(self._lib.pie_comm_connect(self.handle, b"".join(handles)))
            dist.barrier(group=group)  # every rank has mapped every peer before anyone pushes

    def all_reduce(self, partial_f32: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks (rank order: bit-identical on every rank), stream-ordered, no host synchronisation."""
        from . import _ffi
        if partial_f32.dtype != torch.float32 or not partial_f32.is_contiguous() or not partial_f32.is_cuda:
            raise ValueError("HipComm.all_reduce: contiguous float32 device tensor expected")
        _ffi.check(self._lib.pie_allreduce_f32(self.handle, _ffi.p(partial_f32), partial_f32.numel(), _ffi.stream()))
        return partial_f32

    def status(self) -> int:
        """Synchronises; non-zero = a bounded wait for a peer's data gave up (the number is the collective's epoch)."""
        import ctypes as C
        from . import _ffi
        err = C.c_uint(0)
        _ffi.check(self._lib.pie_comm_status(self.handle, C.byref(err)))
        return int(err.value)

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle.value:
            torch.cuda.synchronize()
            if self.world > 1 and dist.is_initialized():
                dist.barrier(group=self.group)  # no peer may still be pushing into this rank's area when it is freed
            self._lib.pie_comm_destroy(self.handle)
            self.handle = None


def fused_shard(config: dict, weights: dict, comm: HipComm, kv_splits: int = 0):
    """This rank's shard of an int4 checkpoint as a models.llama.Model whose decode step is the FUSED launch sequence with
    tensor parallelism inside (o_proj / down_proj as fp32 partials + one-shot all-reduce + residual, vocabulary-parallel tail):
    the product path of SURVEY.md 8 row e.  `weights` is the FULL checkpoint (on the host or the device); only the shard is kept."""
    from .models.llama import Model, ModelArgs
    w, c = shard_checkpoint(weights, config, comm.rank, comm.world)
    dev = torch.device("cuda", torch.cuda.current_device())
    w = {k: v.to(dev) for k, v in w.items()}
    c = {k: v for k, v in c.items() if k not in ("tp_world", "tp_vocab_shard")}
    c["tie_word_embeddings"] = False  # shard_checkpoint always emits this rank's `lm_head.*` rows
    return Model(ModelArgs(**c), w, kv_splits=kv_splits, tp=comm)


class TPLlama:
    """One rank of a tensor-parallel Llama decode step ON THE DEVICE: this rank's shard of an int4 g=64 checkpoint through the
    op-level HIP kernels (hip_ops), the two row-parallel Linears as un-rounded fp32 partials (pie_qgemv_w4g64_f32) summed with
    one all-reduce each, then the single rounding to T and the residual add -- the rounding points of the unsharded graph.
    Functional form of the plan above (one launch per op, no fused epilogues, no graph): it exists for models that do not
    fit one 288 GB card, which none of BASELINE.json's configurations is."""

    def __init__(self, config: dict, weights: dict, tp: TPGroup, device=None):
        from . import hip_ops
        from .cache.kv_cache import ReusableKVCache
        from .models.llama.utils import Llama3RoPE
        self.ops, self.tp = hip_ops, tp
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        w, c = shard_checkpoint(weights, config, tp.rank, tp.world)
        w = {k: v.to(self.device) for k, v in w.items()}
        self.cfg, self.full_cfg = c, config
        self.dtype = w["model.norm.weight"].dtype
        self.nh, self.nkv, self.D = c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"]
        trip = lambda n: (w[f"{n}.weight"], w[f"{n}.scales"], w[f"{n}.biases"])  # noqa: E731
        cat = lambda names: [torch.cat(t, dim=0) for t in zip(*(trip(n) for n in names))]  # noqa: E731
        self.layers = []
        for i in range(config["num_hidden_layers"]):
            p = f"model.layers.{i}"
            self.layers.append(dict(
                attn_norm=w[f"{p}.input_layernorm.weight"], mlp_norm=w[f"{p}.post_attention_layernorm.weight"],
                qkv=hip_ops.repack_w4s(*cat([f"{p}.self_attn.{n}_proj" for n in "qkv"])),      # natural row order: op-level RoPE
                o=hip_ops.repack_w4s(*trip(f"{p}.self_attn.o_proj")),                           # K = local q width
                gate=hip_ops.repack_w4s(*trip(f"{p}.mlp.gate_proj")), up=hip_ops.repack_w4s(*trip(f"{p}.mlp.up_proj")),
                down=hip_ops.repack_w4s(*trip(f"{p}.mlp.down_proj"))))                          # K = local intermediate width
        self.embed = trip("model.embed_tokens")
        self.norm = w["model.norm.weight"]
        self.lm_head = hip_ops.repack_w4s(*trip("lm_head"))                                     # this rank's vocabulary rows
        self.vocab_offset = tp.rank * c["tp_vocab_shard"]
        rs = config.get("rope_scaling") or {}
        max_len = config.get("max_position_embeddings") or 8192
        self.rope = Llama3RoPE(max_len, max_len, self.D, config.get("rope_theta", 10000.0), float(rs.get("factor", 1.0)),
                               float(rs.get("low_freq_factor", 1.0)), float(rs.get("high_freq_factor", 1.0)), device=self.device)
        self.cache = [ReusableKVCache() for _ in self.layers]
        self.eps = float(config["rms_norm_eps"])

    def _row_parallel(self, x: torch.Tensor, wmat) -> torch.Tensor:
        part = self.ops.quantized_matmul_partial(x, wmat)          # fp32 [1, H], this rank's K slice
        self.tp.all_reduce_partial(part)                           # RCCL / gloo sum over the ranks
        return part.to(self.dtype)                                 # the Linear's one rounding to T

    def step(self, token: int) -> tuple[int, float, torch.Tensor]:
        """One decode step of Model.__call__(inputs[1,1]) + greedy tail; returns (next token, logsumexp, hidden [1, H])."""
        ops, D = self.ops, self.D
        ids = torch.tensor([int(token)], dtype=torch.int32, device=self.device)
        x = ops.embedding(ids, *self.embed)                                                     # [1, H] replicated
        offset = self.cache[0].offset
        for lw, kv in zip(self.layers, self.cache):
            xn = ops.rms_norm(x, lw["attn_norm"], self.eps)
            qkv = ops.quantized_matmul(xn, lw["qkv"])[0]
            q, k, v = qkv[:self.nh * D], qkv[self.nh * D:(self.nh + self.nkv) * D], qkv[(self.nh + self.nkv) * D:]
            q = ops.rope(q.reshape(self.nh, 1, D).contiguous(), D, offset=offset, freqs=self.rope.freqs)
            k = ops.rope(k.reshape(self.nkv, 1, D).contiguous(), D, offset=offset, freqs=self.rope.freqs)
            keys, values = kv.update_and_fetch(k[None], v.reshape(1, self.nkv, 1, D))
            att = ops.scaled_dot_product_attention(q[None], kv.keys, kv.values, D ** -0.5, T=kv.offset)   # local heads only
            x = ops.add(x, self._row_parallel(att.reshape(1, self.nh * D), lw["o"]))
            hn = ops.rms_norm(x, lw["mlp_norm"], self.eps)
            act = ops.silu_mul(ops.quantized_matmul(hn, lw["gate"]), ops.quantized_matmul(hn, lw["up"]))
            x = ops.add(x, self._row_parallel(act, lw["down"]))
        xn = ops.rms_norm(x, self.norm, self.eps)
        logits = ops.quantized_matmul(xn, self.lm_head)[0].float()                              # this rank's vocabulary shard
        m = logits.max()
        stats = torch.zeros((self.tp.world, 3), dtype=torch.float64, device=self.device)
        stats[self.tp.rank] = torch.stack([m.double(), torch.exp(logits - m).sum().double(),
                                           (torch.nonzero(logits == m)[0, 0] + self.vocab_offset).double()])
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.tp.group)                       # all-gather of 3 numbers per rank
        M = stats[:, 0].max()
        lse = M + torch.log((stats[:, 1] * torch.exp(stats[:, 0] - M)).sum())
        tok = int(stats[stats[:, 0] == M][:, 2].min().item())                                   # ties: lowest vocabulary index
        return tok, float(lse), x
