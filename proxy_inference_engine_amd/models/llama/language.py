"""Llama-shaped decoder on the MI355X decode runtime.

Host-side mirror of models/llama/language.py of the reference (ModelArgs :13-29, Attention :32-108,
MLP :111-127, TransformerBlock :130-154, LlamaModel :157-187, Model :190-219).  The reference builds a lazy MLX
graph of ~17 primitives per layer; here `Model` owns a native decoder (csrc/decoder.hip) that runs the same
graph as 5 fused HIP launches per layer, and this file only (1) repacks the checkpoint's MLX-quantised triplets
into the streaming layouts at load, and (2) keeps the reference's calling convention:

    logits[1, L, V] = model(inputs[1, L], mask=None, cache=[ReusableKVCache, ...])

Checkpoints: config["quantization"] = {"group_size": 64 | 128, "bits": 2 | 3 | 4 | 6 | 8} (2 / 3-bit codes ride the 4-bit units, 6-bit
codes the 8-bit units), no entry = dense 16-bit weights, or a per-module mix of both; 32-wide groups are refused by name.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import base
from ... import _ffi, hip_ops
from ...cache.kv_cache import BaseCache, PageAllocator, PagedKVCache, PagedSequence, ReusableKVCache
from .utils import Llama3RoPE


class ModelArgs(base.BaseModelArgs):
    """config.json keys the Llama path reads (language.py:13-29; unknown keys are ignored)."""
    model_type: str = "llama"
    hidden_size: int = 0
    num_hidden_layers: int = 0
    intermediate_size: int = 0
    num_attention_heads: int = 0
    rms_norm_eps: float = 1e-5
    vocab_size: int = 0
    head_dim: int | None = None
    max_position_embeddings: int | None = None
    num_key_value_heads: int | None = None
    attention_bias: bool = False
    mlp_bias: bool = False
    rope_theta: float = 10000
    rope_traditional: bool = False
    rope_scaling: dict | None = None
    tie_word_embeddings: bool = True
    quantization: dict | None = None


class TransformerBlock:
    """One entry of `model.layers` (PromptCache.create_kv_cache counts them, prompt_cache.py:39-41).
    Holds the layer's device weights; the arithmetic of language.py:144-154 runs inside the decoder."""

    def __init__(self, attn_norm, mlp_norm, wqkv, wo, wgateup, wdown, biases=(None, None, None, None)):
        self.input_layernorm = attn_norm
        self.post_attention_layernorm = mlp_norm
        self.wqkv, self.wo, self.wgateup, self.wdown = wqkv, wo, wgateup, wdown
        self.bqkv, self.bo, self.bgateup, self.bdown = biases  # attention_bias / mlp_bias (language.py:42-53,117-126), packed row order

    def nbytes(self) -> int:
        return sum(w.nbytes for w in (self.wqkv, self.wo, self.wgateup, self.wdown))


def _is_quantized(weights: dict, prefix: str) -> bool:
    """The reference's per-module predicate (models/utils.py:99-109): a Linear / Embedding is quantised iff the checkpoint holds
    "{path}.scales" (and its input width is a multiple of 64, which mlx_lm guarantees when it wrote the scales)."""
    return f"{prefix}.scales" in weights


def _group_format(weights: dict, names: list[str]) -> bool:
    """Quantised (True) or dense (False) for a group of Linears that this build streams as ONE matrix (q|k|v, gate|up).  The reference
    decides per module; a group whose members disagree cannot be one matrix and is refused by name."""
    flags = [_is_quantized(weights, n) for n in names]
    if any(flags) != all(flags):
        qs = [n for n, f in zip(names, flags) if f]
        ds = [n for n, f in zip(names, flags) if not f]
        raise ValueError(
            f"{', '.join(qs)} quantised but {', '.join(ds)} dense (models/utils.py:99-109 decides per module): the MI355X path streams "
            f"{' | '.join(n.rsplit('.', 1)[-1] for n in names)} as one packed matrix, so these Linears must share a format")
    return flags[0]


def _triplet(weights: dict, prefix: str):
    """The MLX-quantised triplet of one module."""
    w, s, b = weights.get(f"{prefix}.weight"), weights.get(f"{prefix}.scales"), weights.get(f"{prefix}.biases")
    if w is None:
        raise ValueError(f"{prefix}.weight is missing from the checkpoint")
    if s is None or b is None:
        raise ValueError(f'{prefix}: the checkpoint has "{prefix}.{"scales" if s is not None else "biases"}" but not both halves of the affine pair')
    return w, s, b


def _recode_mlx_codes(w: torch.Tensor, bits: int, to_bits: int) -> torch.Tensor:
    """MLX-packed codes of width `bits` (2, 3 or 6; int32 words [N, K * bits / 32]) re-packed at width `to_bits` (4 or 8).  The VALUES are
    untouched (q < 2**bits <= 2**to_bits), so w = scale * q + bias is the same number: the streaming formats here hold 4- and 8-bit codes, and
    a narrower code is a code.  MLX packs a row as a little-endian bit stream (code k at bits [k * bits, (k + 1) * bits)): 2-bit codes sixteen
    to a word, 3- and 6-bit codes eight / four to three bytes."""
    N = w.shape[0]
    by = w.contiguous().view(torch.uint8).reshape(N, -1).to(torch.int32)  # little-endian bytes of the row
    if bits == 2:
        q = torch.stack([(by >> (2 * i)) & 3 for i in range(4)], dim=-1).reshape(N, -1)
    else:
        t = by.reshape(N, -1, 3)
        v = t[..., 0] | (t[..., 1] << 8) | (t[..., 2] << 16)                # 24 bits = 8 three-bit or 4 six-bit codes
        n, m = (8, 7) if bits == 3 else (4, 63)
        q = torch.stack([(v >> (bits * i)) & m for i in range(n)], dim=-1).reshape(N, -1)
    q = q.to(torch.uint8)
    if to_bits == 4:
        q = q[:, 0::2] | (q[:, 1::2] << 4)
    return q.contiguous().view(torch.int32)


def _dense(weights: dict, prefix: str, dtype: torch.dtype) -> torch.Tensor:
    w = weights.get(f"{prefix}.weight")
    if w is None or w.dtype != dtype or f"{prefix}.scales" in weights:
        raise ValueError(f"{prefix}: expected a dense {dtype} weight (no '{prefix}.scales' in the checkpoint)")
    return w


class Model:
    def __init__(self, args: ModelArgs, weights: dict[str, torch.Tensor], kv_splits: int = 0, tp=None):
        """weights: the checkpoint in the layout models/utils.py:51-125 of the reference consumes (HF names,
        `.weight` uint32 codes carried as int32, `.scales`/`.biases` in the activation dtype), on the GPU.
        tp: a connected proxy_inference_engine_amd.tp.HipComm -- this model is then ONE RANK of a tensor-parallel group: `args`
        and `weights` are the rank's shard (tp.shard_checkpoint: local heads / intermediate rows, its own `lm_head` rows, the
        full embedding table), every rank must make the same calls in the same order, and logits / logprobs cover the rank's
        vocabulary rows [rank * V / world, (rank + 1) * V / world) while the sampled token is global."""
        self.args = args
        self.tp = tp
        self.model_type = args.model_type
        device = _ffi.require_gpu()
        q = args.quantization or {}
        self.dense = not q  # no "quantization" entry: nn.Linear / nn.Embedding with 16-bit weights (models/utils.py:96-97)
        if q and (q.get("group_size") not in (32, 64, 128) or q.get("bits") not in (2, 3, 4, 6, 8)):
            # nn.quantize(model, **config["quantization"]) takes any group_size in {32, 64, 128} and bits in {2, 3, 4, 6, 8}
            # (models/utils.py:96-111): all fifteen are served.  64-wide groups are the W4S / W8S streaming units (one group per lane), 128-wide
            # groups write every scale / bias to both of their 64-wide halves (below), 32-wide groups are the W4S32 / W8S32 units (two scale /
            # bias pairs per lane); 2- / 3-bit codes ride the 4-bit units and 6-bit codes the 8-bit units (below: a narrower code is a code).
            raise ValueError(f"config['quantization'] = {dict(q)}: mx.quantize knows group_size 32, 64, 128 and bits 2, 3, 4, 6, 8 "
                             f"(got group_size={q.get('group_size')}, bits={q.get('bits')})")
        self.checkpoint_bits = int(q["bits"]) if q else 16
        # 2- and 6-bit codes in 64- / 128-wide groups stream as W2S / W6S units, 0.3125 / 0.8125 B per weight -- the checkpoint's own bytes (round 5).
        # Only the embedding TABLE (one row per step) is re-packed as 4- / 8-bit codes; a tied lm_head is packed from the original codes.
        self.native_narrow = self.checkpoint_bits if bool(q) and self.checkpoint_bits in (2, 6) and q.get("group_size") in (64, 128) else 0
        self.native_w2 = self.native_narrow == 2
        embed_codes_narrow = None
        if q and self.checkpoint_bits in (2, 3, 6):
            # Same weights, wider container: 3-bit codes (and 2- / 6-bit codes in 32-wide groups) are stored as 4-bit codes / bytes, scales and biases
            # unchanged -- every product is what the narrow code gives (same q, same affine pair, same fp32 sums); HBM holds 0.5625 / 1.0625 B per
            # weight instead of the checkpoint's 0.4375 (3-bit: a native unit would need two planes at ~2.1 VALU instructions per weight and lose).
            to_bits = 8 if self.checkpoint_bits == 6 else 4
            weights = dict(weights)
            for k in [k for k in weights if k.endswith(".scales")]:
                wk = k[:-len(".scales")] + ".weight"
                if weights[wk].shape[-1] * 32 % self.checkpoint_bits or (weights[wk].shape[-1] * 32 // self.checkpoint_bits) % 64:
                    raise ValueError(f"{wk}: {self.checkpoint_bits}-bit rows must hold a multiple of 64 codes")
                if self.native_narrow:
                    if wk == "model.embed_tokens.weight":
                        embed_codes_narrow = weights[wk]
                        weights[wk] = _recode_mlx_codes(weights[wk], self.checkpoint_bits, to_bits)
                    continue
                weights[wk] = _recode_mlx_codes(weights[wk], self.checkpoint_bits, to_bits)
            q = dict(q, bits=to_bits)
        self.group_size = int(q.get("group_size", 64)) if q else 0
        if self.group_size == 128:
            # mx.quantize(w, group_size=128): one (scale, bias) per 128 weights.  The streaming units keep one per 64-wide lane group, so each
            # is written twice: w = s q + b holds for both halves unchanged -- the same dequantised matrix (qmm regime: bit for bit), the same
            # affine sums up to fp32 association (qmv regime) -- for 0.5625 instead of the checkpoint's 0.53125 B per weight in HBM.
            weights = dict(weights)
            for k in [k for k in weights if k.endswith(".scales") or k.endswith(".biases")]:
                wk = k[:k.rindex(".")] + ".weight"
                code_bits = self.checkpoint_bits if self.native_narrow and wk != "model.embed_tokens.weight" else int(q["bits"])  # native narrow Linears keep their codes
                if (weights[wk].shape[-1] * 32 // code_bits) % 128:
                    raise ValueError(f"{k}: group_size 128 needs a multiple of 128 input features")
                weights[k] = weights[k].repeat_interleave(2, dim=-1).contiguous()
        self.bits = int(q["bits"]) if q else 16
        self.n_heads = args.num_attention_heads
        self.n_kv_heads = args.num_key_value_heads or self.n_heads
        self.head_dim = args.head_dim or args.hidden_size // self.n_heads
        self.dtype = weights["model.norm.weight"].dtype
        self.device = device
        H, I, V = args.hidden_size, args.intermediate_size, args.vocab_size
        if tp is not None:
            if (args.tie_word_embeddings and tp.world > 1) or V % (2 * tp.world):
                raise ValueError("a tensor-parallel shard carries its own lm_head rows (tie_word_embeddings=False) and needs vocab_size % (2 * world) == 0")
            V //= tp.world  # the rank's slice of the vocabulary: lm_head rows, logits, logprobs
        self.vocab_out = V
        weights = base.sanitize(weights, args.tie_word_embeddings)  # language.py:212-219

        rs = args.rope_scaling or {}
        max_len = args.max_position_embeddings or 8192  # language.py:56
        self.rope = Llama3RoPE(max_len, max_len, self.head_dim, args.rope_theta, float(rs.get("factor", 1.0)),
                               float(rs.get("low_freq_factor", 1.0)), float(rs.get("high_freq_factor", 1.0)), device=device)

        # rotate-half RoPE wants partners (i, i + D/2) on adjacent packed rows; the traditional form rotates (2i, 2i+1), which
        # already are adjacent in the plain q|k|v concatenation
        qkv_map = None if args.rope_traditional else hip_ops.qkv_row_map(self.n_heads, self.n_kv_heads, self.head_dim).to(device)
        gu_map = hip_ops.gateup_row_map(I).to(device)

        g32 = self.group_size == 32
        wfmt = (4 if self.bits == 8 else 3) if g32 else (2 if self.bits == 8 else 0)  # PIE_W_INT8_G32 / INT4_G32 / INT8_G64 / INT4_G64
        fmt_code = {False: 2, True: wfmt + 1}  # pie_layer_weights.fmt_*: PIE_W_* + 1 (0 = the decoder's default format)
        PIE_W_INT2_G64, PIE_W_INT6_G64 = 5, 6  # W2S / W6S units: every Linear of a native 2- / 6-bit checkpoint (the embedding table stays wfmt = 4- / 8-bit codes)
        self.mixed = False  # some module is dense although config["quantization"] is set (per-module predicate, models/utils.py:99-109)

        def quantized(names: list[str]) -> bool:
            if self.dense:
                return False
            qf = _group_format(weights, names)
            self.mixed |= not qf
            return qf

        def pack(names: list[str], row_map=None):
            """One streaming-layout matrix from the (concatenated) Linear weights `names`; returns (matrix, format code)."""
            if not quantized(names):
                ws = [_dense(weights, n, self.dtype) for n in names]
                return hip_ops.repack_dense(torch.cat(ws, dim=0) if len(ws) > 1 else ws[0], row_map=row_map), fmt_code[False]
            trip = [torch.cat(t, dim=0) for t in zip(*(_triplet(weights, n) for n in names))]
            if self.native_narrow:
                if names == ["model.embed_tokens"]:  # tied lm_head: the table's own narrow codes, not the gather's 4- / 8-bit copy
                    trip[0] = embed_codes_narrow
                if self.native_narrow == 6:
                    return hip_ops.repack_w6s(*trip, row_map=row_map), PIE_W_INT6_G64 + 1
                return hip_ops.repack_w2s(*trip, row_map=row_map), PIE_W_INT2_G64 + 1
            if g32:
                return hip_ops.repack_w4s32(*trip, row_map=row_map, bits=self.bits), fmt_code[True]
            return (hip_ops.repack_w8s if self.bits == 8 else hip_ops.repack_w4s)(*trip, row_map=row_map), fmt_code[True]

        def bias(names: list[str], row_map=None):
            """The (concatenated) Linear biases of `names` in the packed row order of the matching matrix; a Linear without a
            `.bias` entry contributes zeros (Qwen2-style checkpoints carry q/k/v biases but none for o_proj), None if none has one."""
            if not any(f"{n}.bias" in weights for n in names):
                return None
            rows = [int((weights[f"{n}.weight"]).shape[0]) for n in names]
            parts = [weights[f"{n}.bias"].reshape(-1).to(self.dtype) if f"{n}.bias" in weights
                     else torch.zeros(r, dtype=self.dtype, device=device) for n, r in zip(names, rows)]
            b = torch.cat(parts)
            return (b[row_map.long()] if row_map is not None else b).contiguous()

        self.layers: list[TransformerBlock] = []
        for i in range(args.num_hidden_layers):
            pfx = f"model.layers.{i}"
            (wqkv, f_qkv), (wo, f_o) = pack([f"{pfx}.self_attn.{n}_proj" for n in "qkv"], qkv_map), pack([f"{pfx}.self_attn.o_proj"])
            (wgu, f_gu), (wdown, f_down) = pack([f"{pfx}.mlp.gate_proj", f"{pfx}.mlp.up_proj"], gu_map), pack([f"{pfx}.mlp.down_proj"])
            blk = TransformerBlock(
                weights[f"{pfx}.input_layernorm.weight"].contiguous(),
                weights[f"{pfx}.post_attention_layernorm.weight"].contiguous(),
                wqkv, wo, wgu, wdown,
                biases=(bias([f"{pfx}.self_attn.{n}_proj" for n in "qkv"], qkv_map) if args.attention_bias else None,
                        bias([f"{pfx}.self_attn.o_proj"]) if args.attention_bias else None,
                        bias([f"{pfx}.mlp.gate_proj", f"{pfx}.mlp.up_proj"], gu_map) if args.mlp_bias else None,
                        bias([f"{pfx}.mlp.down_proj"]) if args.mlp_bias else None),
            )
            blk.formats = (f_qkv, f_o, f_gu, f_down)
            self.layers.append(blk)
        self.embed_quantized = quantized(["model.embed_tokens"])
        if not self.embed_quantized:
            self.embed_tokens = (_dense(weights, "model.embed_tokens", self.dtype).contiguous(), None, None)
        else:
            self.embed_tokens = tuple(t.contiguous() for t in _triplet(weights, "model.embed_tokens"))
        self.norm = weights["model.norm.weight"].contiguous()
        head = "model.embed_tokens" if args.tie_word_embeddings else "lm_head"  # language.py:206-209
        self.lm_head, f_head = pack([head])
        f_embed = fmt_code[self.embed_quantized]

        # decoder-owned outputs live in torch tensors so callers can read them without copies
        self.logits = torch.zeros(V, dtype=self.dtype, device=device)
        self.logprobs = torch.zeros(V, dtype=torch.float32, device=device)
        self.token = torch.zeros(1, dtype=torch.int32, device=device)
        self.hidden = torch.zeros(H, dtype=self.dtype, device=device)

        lib = _ffi.load()
        cfg = _ffi.pie_decoder_config(_ffi.dtype_code(self.dtype), H, args.num_hidden_layers, self.n_heads, self.n_kv_heads,
                                      self.head_dim, I, V, float(args.rms_norm_eps), int(args.tie_word_embeddings), int(kv_splits),
                                      1 if self.dense else wfmt, int(bool(args.rope_traditional)),
                                      tp.rank if tp is not None else 0, tp.world if tp is not None else 0)
        self._dec = C.c_void_p()
        _ffi.check(lib.pie_decoder_create(C.byref(cfg), C.byref(self._dec)))
        if tp is not None:  # world == 1: the tensor-parallel code path on one rank (push to self / one-rank RCCL communicator)
            _ffi.check(lib.pie_decoder_set_comm(self._dec, tp.handle))
        for i, blk in enumerate(self.layers):
            lw = _ffi.pie_layer_weights(blk.input_layernorm.data_ptr(), blk.post_attention_layernorm.data_ptr(),
                                        blk.wqkv.packed.data_ptr(), blk.wo.packed.data_ptr(), blk.wgateup.packed.data_ptr(),
                                        blk.wdown.packed.data_ptr(),
                                        *(b.data_ptr() if b is not None else None for b in (blk.bqkv, blk.bo, blk.bgateup, blk.bdown)),
                                        *blk.formats)
            _ffi.check(lib.pie_decoder_set_layer(self._dec, i, C.byref(lw)))
        gw = _ffi.pie_global_weights(self.embed_tokens[0].data_ptr(), *(t.data_ptr() if t is not None else None for t in self.embed_tokens[1:]),
                                     self.norm.data_ptr(), self.lm_head.packed.data_ptr(), self.rope.freqs.data_ptr(), f_embed, f_head)
        _ffi.check(lib.pie_decoder_set_globals(self._dec, C.byref(gw)))
        # device-side token history: history[p] = greedy token chosen for position p (written by the tail kernel)
        self.history = torch.zeros(1 << 20, dtype=torch.int32, device=device)
        _ffi.check(lib.pie_decoder_bind_outputs(self._dec, _ffi.p(self.logits), _ffi.p(self.logprobs), _ffi.p(self.token), _ffi.p(self.hidden),
                                                _ffi.p(self.history), self.history.numel()))
        self._kv_key = None      # (pointers, capacity) currently in the decoder's device table
        self._kv_hold = None
        self._page_pool, self._page_blocks = None, 16
        self._batch_bufs: dict = {}
        self._dev_offset = None  # device-side cache offset the decoder believes in
        torch.cuda.synchronize(device)

    def __del__(self):
        dec = getattr(self, "_dec", None)
        if dec is not None and dec.value:
            try:
                _ffi.load().pie_decoder_destroy(dec)
            except Exception:
                pass
            self._dec = None

    # ------------------------------------------------------------------ cache plumbing
    def make_cache(self) -> list[BaseCache]:
        if self._page_pool is not None:
            return self.make_paged_cache(self._page_pool, max_blocks=self._page_blocks)
        return [ReusableKVCache() for _ in self.layers]

    def enable_paged_kv(self, num_pages: int = 512, max_blocks: int = 16, kv_dtype: torch.dtype | None = None,
                        kv_scales: tuple[torch.Tensor, torch.Tensor] | None = None) -> PageAllocator:
        """From now on make_cache() (PromptCache.create_kv_cache, prompt_cache.py:34-41) hands out paged caches drawing
        on one pool of `num_pages` 64-token pages (all layers).  Returns the pool.
        kv_dtype=torch.int8: the pages are the reference KVPage's own storage (page.hpp:25-32) -- int8 K / V rows with float16 per-head
        scales; kv_scales = (k, v), each float16 [n_layers, n_kv_heads], is written into every page (None: ones, the reference
        constructor's value -- far too coarse for real activations; pass amax / 127 of a calibration prompt).  Such a pool serves
        prefill_batch / step_batch (the continuous-batching path) and the single-sequence step() / InferenceEngine: a fresh token prompt goes through
        the several-prompts pass as a batch of one (quantised into its pages on the way); a suffix behind a cached prefix, and a prompt of
        embeddings, run as decode steps (the batched single-sequence pass reads T pages)."""
        kv_dtype = self.dtype if kv_dtype is None else kv_dtype
        if kv_dtype not in (self.dtype, torch.int8):
            raise ValueError(f"kv_dtype must be the model's dtype or torch.int8, got {kv_dtype}")
        self._page_pool = PageAllocator(num_pages, self.n_kv_heads, self.head_dim, dtype=kv_dtype, device=self.device,
                                        num_layers=len(self.layers))
        if kv_dtype == torch.int8 and kv_scales is not None:
            ks, vs = (t.to(device=self.device, dtype=torch.float16).contiguous() for t in kv_scales)
            if ks.shape != (len(self.layers), self.n_kv_heads) or vs.shape != ks.shape:
                raise ValueError("kv_scales: two float16 tensors [n_layers, n_kv_heads]")
            for li in range(len(self.layers)):
                hip_ops.page_i8_set_scales(self._page_pool.slab[li], num_pages, self.n_kv_heads, self.head_dim, ks[li], vs[li])
        _ffi.check(_ffi.load().pie_decoder_configure(self._dec, _ffi.PIE_OPT_KV_I8, int(kv_dtype == torch.int8)))
        self._kv_i8 = kv_dtype == torch.int8
        self._kv_key = None
        self._page_blocks = max_blocks
        return self._page_pool

    def make_paged_cache(self, allocator: PageAllocator | None = None, num_pages: int = 512, max_blocks: int = 16) -> list[BaseCache]:
        """Per-layer PagedKVCache objects over one PagedSequence: KV rows live in 64-token pages of
        the allocator's slab (one plane per layer) instead of per-layer contiguous buffers; growth takes pages, never
        copies.  Pass a shared `allocator` to keep several sequences in one pool."""
        if allocator is None:
            allocator = PageAllocator(num_pages, self.n_kv_heads, self.head_dim, dtype=self.dtype, device=self.device,
                                      num_layers=len(self.layers))
        if (allocator.num_layers, allocator.num_heads, allocator.head_dim) != (len(self.layers), self.n_kv_heads, self.head_dim) or \
                allocator.dtype not in (self.dtype, torch.int8) or allocator.slab is None:
            raise ValueError("the allocator's geometry does not match this model")
        seq = PagedSequence(allocator, max_blocks)
        return [PagedKVCache(seq, i) for i in range(len(self.layers))]

    def _match_page_format(self, allocator: PageAllocator) -> None:
        """The decoder indexes the slabs with the page stride of ITS page format (PIE_OPT_KV_I8); a pool handed to make_cache(allocator=...)
        may be of the other one.  The batch entry points therefore follow the pool they are given (and the C ABI checks the slab size)."""
        want = allocator.dtype == torch.int8
        if getattr(self, "_kv_i8", None) != want:
            _ffi.check(_ffi.load().pie_decoder_configure(self._dec, _ffi.PIE_OPT_KV_I8, int(want)))
            self._kv_i8 = want

    def _sync_paged(self, cache: list[PagedKVCache], n_new: int) -> None:
        seq = cache[0].page_manager
        if any(not isinstance(c, PagedKVCache) or c.page_manager is not seq for c in cache):
            raise TypeError("the layers of a paged cache must share one PagedSequence")
        seq.reserve(n_new)
        a = seq.allocator
        self._match_page_format(a)  # int8 pages (round 4): the step's new K / V row is quantised into the sequence's page, attention reads the codes back
        key = ("paged", a.slab.data_ptr(), a.size(), seq.table.data_ptr(), seq.max_blocks, a.dtype == torch.int8)
        lib = _ffi.load()
        if key != self._kv_key:
            n = len(cache)
            slabs = (C.c_void_p * n)(*[a.slab[i].data_ptr() for i in range(n)])
            _ffi.check(lib.pie_decoder_set_paged_kv(self._dec, slabs, a.size(), _ffi.p(seq.table), seq.max_blocks, _ffi.stream()))
            self._kv_key = key
            self._kv_hold = (a, seq.table)  # keeps the slab and the table alive while the decoder points at them
        if self._dev_offset != seq.offset:
            _ffi.check(lib.pie_decoder_set_state(self._dec, seq.offset, -1, _ffi.stream()))
            self._dev_offset = seq.offset

    def _sync_cache(self, cache: list[ReusableKVCache], n_new: int) -> None:
        """The host half of cache.update_and_fetch for every layer (reusable.py:113-131), then make the decoder's
        device-side view (buffer addresses, capacity, offset) match the Python objects."""
        if len(cache) != len(self.layers):
            raise ValueError(f"expected {len(self.layers)} layer caches, got {len(cache)}")
        if isinstance(cache[0], PagedKVCache):
            return self._sync_paged(cache, n_new)
        off = cache[0].offset
        for c in cache:
            if not isinstance(c, ReusableKVCache):
                raise TypeError("the decode path runs on ReusableKVCache (prompt_cache.py:73)")
            if c.offset != off:
                raise ValueError("layer caches disagree on offset")
            c.reserve(n_new, self.n_kv_heads, self.head_dim, self.dtype, self.device)
        cap = min(c.capacity for c in cache)
        key = (tuple(c.keys.data_ptr() for c in cache), tuple(c.values.data_ptr() for c in cache), cap)
        lib = _ffi.load()
        if key != self._kv_key:
            n = len(cache)
            kp = (C.c_void_p * n)(*key[0])
            vp = (C.c_void_p * n)(*key[1])
            _ffi.check(lib.pie_decoder_set_kv(self._dec, kp, vp, cap, _ffi.stream()))
            self._kv_key = key
        if self._dev_offset != off:
            _ffi.check(lib.pie_decoder_set_state(self._dec, off, -1, _ffi.stream()))
            self._dev_offset = off

    def _advance(self, cache, n: int) -> None:
        if isinstance(cache[0], PagedKVCache):
            cache[0].page_manager.advance(n)
        else:
            for c in cache:
                c.advance(n)  # reusable.py:139
        self._dev_offset += n

    # ------------------------------------------------------------------ reference calling convention
    def __call__(self, inputs: torch.Tensor | None = None, mask=None, cache: list[BaseCache] | None = None,
                 inputs_embeds: torch.Tensor | None = None) -> torch.Tensor:
        """Model.__call__ (language.py:199-210): inputs [1, L] -> logits [1, L, V] in the activation dtype,
        lm_head on every position like the reference (the engine's fast path is `step`).
        inputs_embeds [1, L, hidden] replaces embed_tokens(inputs): the VLM text tower's entry
        (models/intern/language.py:148-158, LanguageModel(None, cache=cache, inputs_embeds=...), intern/ensemble.py:108)."""
        if inputs is None and inputs_embeds is None:
            raise ValueError("Either inputs or inputs_embeds must be provided")  # intern/language.py:188-189
        if cache is None:
            cache = self.make_cache()  # reference: cache=None means no caching; a throw-away cache is equivalent
        if mask is not None:
            self._check_mask(mask, (inputs_embeds.shape[-2] if inputs_embeds is not None else inputs.shape[-1]), cache)
        lib = _ffi.load()
        if inputs_embeds is not None:
            emb = self._check_embeds(inputs_embeds)
            L = emb.shape[0]
            self._sync_cache(cache, L)
            out = torch.empty((L, self.vocab_out), dtype=self.dtype, device=self.device)
            _ffi.check(lib.pie_decoder_prefill_embeds(self._dec, _ffi.p(emb), L, _ffi.p(out), _ffi.stream()))
        else:
            if inputs.dim() != 2 or inputs.shape[0] != 1:
                raise ValueError("batch-1 path: inputs must be [1, L]")
            ids = inputs.reshape(-1).to(device=self.device, dtype=torch.int32).contiguous()
            L = ids.numel()
            self._sync_cache(cache, L)
            out = torch.empty((L, self.vocab_out), dtype=self.dtype, device=self.device)
            _ffi.check(lib.pie_decoder_prefill(self._dec, _ffi.p(ids), L, _ffi.p(out), _ffi.stream()))
        self._advance(cache, L)
        return out.unsqueeze(0)

    def _check_mask(self, mask, L: int, cache) -> None:
        """Model.__call__(mask=...) (language.py:199-204): the reference builds the causal mask itself when none is given
        (models/base.py:37-53) and otherwise hands the caller's to sdpa.  The kernels here apply the causal mask implicitly, so a
        caller's mask is accepted when it IS that mask -- "causal", or an array blocking exactly the future positions (additive: < 0
        where blocked; boolean: False where blocked) -- and refused otherwise instead of being silently ignored."""
        if isinstance(mask, str):
            if mask != "causal":
                raise NotImplementedError(f"mask={mask!r}: only the causal mask is supported")
            return
        c0 = cache[0]
        offset = int(c0.page_manager.offset if isinstance(c0, PagedKVCache) else c0.offset)
        m = torch.as_tensor(mask)
        blocked = (~m) if m.dtype == torch.bool else (m < 0)
        blocked = blocked.reshape(-1, blocked.shape[-1]) if blocked.dim() > 2 else blocked
        want = base.create_causal_mask(L, offset, device=blocked.device) < 0
        if blocked.shape != want.shape or not torch.equal(blocked, want):
            raise NotImplementedError(f"an explicit mask of shape {tuple(m.shape)} that is not the causal mask for {L} new positions at offset {offset} "
                                      "is not supported: the attention kernels apply the causal mask of models/base.py:37-53 implicitly")

    def _check_embeds(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        emb = inputs_embeds
        if emb.dim() == 3:
            if emb.shape[0] != 1:
                raise ValueError("batch-1 path: inputs_embeds must be [1, L, hidden]")
            emb = emb[0]
        if emb.dim() != 2 or emb.shape[1] != self.args.hidden_size or emb.shape[0] == 0:
            raise ValueError(f"inputs_embeds must be [L, {self.args.hidden_size}]")
        return emb.to(device=self.device, dtype=self.dtype).contiguous()

    def embed(self, ids: torch.Tensor) -> torch.Tensor:
        """embed_tokens(ids) -> [L, hidden] (language.py:176; the VLM ensemble starts from it, intern/ensemble.py:46)."""
        from ... import hip_ops
        ids = ids.reshape(-1).to(device=self.device, dtype=torch.int32).contiguous()
        if self.embed_tokens[1] is None:
            return hip_ops.embedding_dense(ids, self.embed_tokens[0])
        return hip_ops.embedding(ids, *self.embed_tokens, bits=self.bits, group_size=32 if self.group_size == 32 else 64)

    def step_embeds(self, inputs_embeds: torch.Tensor, cache: list[BaseCache]):
        """`step` for a prompt given as embeddings: forwards the rows, lm_head + tail on the last one only.
        On an int8 page pool such a prompt runs as L decode steps (~1.2 ms per row on the 8B model: the batched single-sequence pass reads
        T pages and the several-prompts pass takes token ids only); warned about once."""
        emb = self._check_embeds(inputs_embeds)
        if emb.shape[0] >= 6 and isinstance(cache[0], PagedKVCache) and cache[0].page_manager.allocator.dtype == torch.int8 and not getattr(self, "_warned_i8_embeds", False):
            import warnings
            warnings.warn("a prompt of embeddings on int8 KV pages is processed one row per decode step; use T pages for VLM prompts, or token prompts", stacklevel=2)
            self._warned_i8_embeds = True
        L = emb.shape[0]
        self._sync_cache(cache, L)
        _ffi.check(_ffi.load().pie_decoder_prefill_embeds(self._dec, _ffi.p(emb), L, None, _ffi.stream()))
        self._advance(cache, L)
        pos = self._dev_offset
        token = self.history[pos:pos + 1] if pos < self.history.numel() else self.token.clone()
        return token, self.logprobs, self.logits

    def step(self, ids: torch.Tensor | None, cache: list[BaseCache], graph: bool = True):
        """Fast path of _inference (engine/inference_engine.py:252-271) for the greedy sampler without logits
        processors: forwards `ids` [L] (device int32) and returns (token[1], logprobs[V], logits[V]).
        `ids=None` feeds back the previous step's greedy token, which already sits in the decoder's device-side
        state (no copy, no host sync).  L == 1 replays the captured hipGraph.
        token is a view of the device-side history at the new position (stable); logprobs / logits are the
        decoder's output buffers, valid until the next call."""
        lib = _ffi.load()
        if ids is None:
            L = 1
            self._sync_cache(cache, L)
        else:
            ids = ids.reshape(-1).to(device=self.device, dtype=torch.int32).contiguous()
            L = ids.numel()
            if L >= 6 and isinstance(cache[0], PagedKVCache) and cache[0].page_manager.allocator.dtype == torch.int8 and cache[0].offset == 0:
                # A fresh prompt on int8 pages: the single-sequence prompt pass reads T pages (it would run the prompt as L decode steps,
                # ~1.2 ms per token), the several-prompts pass quantises into int8 pages -- one prompt is a batch of one.
                nxt, logprobs, logits = self.prefill_batch([ids.cpu().numpy()], [cache])  # (a prompt arrives once: the host copy is the pass's own row bookkeeping)
                pos = cache[0].offset  # position the chosen token will occupy
                self._kv_key = None     # the pass bound its own table: the next step re-binds the sequence and sets the device-side offset
                self._dev_offset = None
                _ffi.check(lib.pie_decoder_set_token_from(self._dec, _ffi.p(nxt), _ffi.stream()))  # device to device: step(None) feeds it back
                if pos < self.history.numel():
                    self.history[pos:pos + 1].copy_(nxt[:1])
                    return self.history[pos:pos + 1], logprobs[0], logits[0]
                return nxt[:1].clone(), logprobs[0], logits[0]
            self._sync_cache(cache, L)
            if L == 1:
                _ffi.check(lib.pie_decoder_set_token_from(self._dec, _ffi.p(ids), _ffi.stream()))
        if L == 1:
            flags = _ffi.PIE_STEP_LOGITS | (_ffi.PIE_STEP_GRAPH if graph else 0)
            _ffi.check(lib.pie_decoder_step(self._dec, flags, _ffi.stream()))
        else:
            _ffi.check(lib.pie_decoder_prefill(self._dec, _ffi.p(ids), L, None, _ffi.stream()))
        self._advance(cache, L)
        pos = self._dev_offset  # position the chosen token will occupy
        token = self.history[pos:pos + 1] if pos < self.history.numel() else self.token.clone()
        return token, self.logprobs, self.logits

    def step_batch(self, tokens: torch.Tensor, caches: list[list[BaseCache]], graph: bool = True):
        """One decode step for several sequences at once (continuous batching over the page pool; pie_decoder_step_batch):
        tokens [B] = each sequence's input token, caches = their per-layer PagedKVCache lists (model.make_cache() after
        enable_paged_kv(), all drawing on one PageAllocator, each already holding its prompt -- e.g. through step()).
        Returns (next_tokens [B] int32 greedy, logprobs [B, V] fp32, logits [B, V]) -- buffers owned by the model per batch size,
        valid until the next step_batch of that size; every cache advances by one position.  graph: replay a captured hipGraph
        of the step while the batch size and table width stay the same (captured on the second such step).
        The weights stream once for the whole batch: int4 models run the few-row MFMA GEMM up to 32 sequences."""
        seqs = []
        for c in caches:
            if len(c) != len(self.layers) or not isinstance(c[0], PagedKVCache):
                raise TypeError("step_batch runs on paged caches (enable_paged_kv(), then make_cache())")
            seqs.append(c[0].page_manager)
        a = seqs[0].allocator
        if any(s.allocator is not a for s in seqs) or len({id(s) for s in seqs}) != len(seqs):
            raise ValueError("step_batch: distinct sequences of one page pool")
        B = len(seqs)
        tokens = tokens.reshape(-1).to(device=self.device, dtype=torch.int32).contiguous()
        if tokens.numel() != B:
            raise ValueError("step_batch: one token per sequence")
        for s in seqs:
            s.reserve(1)
        # Persistent device buffers per batch size (the captured graph of the step keeps pointing at them): inputs are copied in,
        # the block table (batch_details.hpp:52-66) is rewritten in place when a sequence takes a page or the batch changes.
        mb = max(len(s.pages) for s in seqs)
        buf = self._batch_bufs.get(B)
        if buf is not None:
            self._batch_bufs[B] = self._batch_bufs.pop(B)  # most recently used last
        if buf is None or buf["table"].shape[1] < mb:
            V = self.args.vocab_size
            width = max(mb, 2 * buf["table"].shape[1]) if buf is not None else max(mb, 4)
            buf = {"table": torch.zeros((B, width), dtype=torch.int32, device=self.device),
                   "tokens": torch.empty(B, dtype=torch.int32, device=self.device), "ctx": torch.empty(B, dtype=torch.int32, device=self.device),
                   "logits": torch.empty((B, V), dtype=self.dtype, device=self.device),
                   "logprobs": torch.empty((B, V), dtype=torch.float32, device=self.device),
                   "next": torch.empty(B, dtype=torch.int32, device=self.device), "key": None}
            self._batch_bufs.pop(B, None)
            while len(self._batch_bufs) >= 4:  # a serving loop revisits few batch sizes: keep the 4 most recent sets (each [B, V] fp32 + T)
                self._batch_bufs.pop(next(iter(self._batch_bufs)))
            self._batch_bufs[B] = buf
        key = tuple(tuple(s.pages) for s in seqs)  # the page ids themselves: truncate + regrow reorders them at equal length, and id() of a retired sequence can be reused
        if buf["key"] != key:
            table = torch.zeros(buf["table"].shape, dtype=torch.int32)
            for i, s in enumerate(seqs):
                table[i, :len(s.pages)] = torch.tensor(s.pages, dtype=torch.int32)
            buf["table"].copy_(table)
            buf["key"] = key
        buf["tokens"].copy_(tokens)
        buf["ctx"].copy_(torch.tensor([s.offset + 1 for s in seqs], dtype=torch.int32))
        n = len(self.layers)
        slabs = (C.c_void_p * n)(*[a.slab[i].data_ptr() for i in range(n)])
        self._match_page_format(a)
        _ffi.check(_ffi.load().pie_decoder_step_batch(self._dec, _ffi.p(buf["tokens"]), _ffi.p(buf["ctx"]), slabs, a.size(), a.slab[0].numel() * a.slab.element_size(), _ffi.p(buf["table"]),
                                                      buf["table"].shape[1], B, _ffi.p(buf["logits"]), _ffi.p(buf["logprobs"]), _ffi.p(buf["next"]),
                                                      _ffi.PIE_STEP_GRAPH if graph else 0, _ffi.stream()))
        nxt, logprobs, logits = buf["next"], buf["logprobs"], buf["logits"]
        for s in seqs:
            s.advance(1)
        return nxt, logprobs, logits

    def prefill_batch(self, prompts: list, caches: list[list[BaseCache]]):
        """Several fresh prompts in ONE pass (pie_decoder_prefill_batch): their rows are concatenated for the GEMMs, every row keeps
        its own position, every prompt its own pages and a causal attention over its own rows only.  caches: empty paged caches
        (make_cache() after enable_paged_kv(), one per prompt).  Returns (next_tokens [S], logprobs [S, V], logits [S, V]) for the
        prompts' last positions; every cache then holds its prompt."""
        return self._varlen_pass("prefill_batch", None, [], prompts, caches)

    def step_mixed(self, tokens: torch.Tensor | None, decode_caches: list[list[BaseCache]], prompts: list, prompt_caches: list[list[BaseCache]]):
        """Decode-state sequences AND fresh prompts in one pass over the weights (pie_decoder_step_mixed; the reference's BatchDetails holds
        both kinds, batch_details.hpp:10-88): tokens [B] = the input token of each decoding sequence (caches as in step_batch), prompts /
        prompt_caches as in prefill_batch -- or caches that already hold a PREFIX (offset > 0: the next chunk of a long prompt, or a suffix
        behind shared prefix pages of a forked sequence): such rows attend to the sequence's pages from that offset.  Returns (next_tokens [B + S], logprobs [B + S, V], logits [B + S, V]): the decoding sequences
        first, then every prompt's last position; the decode caches advance by one position, the prompt caches hold their prompts.
        Every row rides the many-row regime of the Linears (as the rows of a prompt do), also when B alone would take the few-row one."""
        return self._varlen_pass("step_mixed", tokens, decode_caches, prompts, prompt_caches)

    def _varlen_pass(self, who: str, tokens, decode_caches, prompts, caches):
        import numpy as np
        seqs, dseqs = [], []
        for c in list(decode_caches) + list(caches):
            if len(c) != len(self.layers) or not isinstance(c[0], PagedKVCache):
                raise TypeError(f"{who} runs on paged caches (enable_paged_kv(), then make_cache())")
        for c in caches:
            if c[0].offset != 0 and who == "prefill_batch":
                raise ValueError(f"{who} takes fresh caches for the prompts (nothing cached before the prompt)")
            seqs.append(c[0].page_manager)
        dseqs = [c[0].page_manager for c in decode_caches]
        every = dseqs + seqs
        if not every:
            raise ValueError(f"{who}: an empty batch")
        a = every[0].allocator
        if len(prompts) != len(seqs) or any(s.allocator is not a for s in every) or len({id(s) for s in every}) != len(every):
            raise ValueError(f"{who}: one distinct sequence of one page pool per prompt and per decoding row")
        B = len(dseqs)
        if B:
            tokens = torch.as_tensor(tokens).reshape(-1).to(device=self.device, dtype=torch.int32)   # stays on the device: no host sync per pass
            if tokens.numel() != B:
                raise ValueError(f"{who}: one token per decoding sequence")
            if any(s.offset < 1 for s in dseqs):
                raise ValueError(f"{who}: a decoding sequence holds its prompt already")
        lens = [len(p) for p in prompts]
        if (lens and min(lens) < 1) or B + sum(lens) > 65535:
            raise ValueError(f"{who}: prompts must be non-empty and the pass holds at most 65535 rows")
        cached = [int(s.offset) for s in seqs]      # > 0: the prompt continues a cached prefix (a chunk of a long prompt, a suffix behind shared pages)
        if any(cached) and a.dtype == torch.int8:
            raise ValueError(f"{who}: a prompt that continues a cached prefix reads T pages (int8 pools: fresh prompts and decoding rows)")
        for s in dseqs:
            s.reserve(1)
        for s, n in zip(seqs, lens):
            s.reserve(n)
        S, N = len(seqs), sum(lens)
        starts = (B + np.concatenate([[0], np.cumsum(lens)[:-1]])).astype(np.int32) if S else np.zeros(0, np.int32)
        ids = np.concatenate([np.zeros(B, np.int32)] + [np.asarray(p, dtype=np.int32).reshape(-1) for p in prompts])   # the decode rows' ids are copied in on the device
        rows_d = np.arange(B, dtype=np.int32)
        rows_p = np.arange(B, B + N, dtype=np.int32)
        cont = np.repeat(np.asarray(cached, dtype=np.int32) > 0, lens) if S else np.zeros(0, bool)   # rows of continuing prompts: trivial segments
        row_seq = np.concatenate([rows_d, B + np.repeat(np.arange(S, dtype=np.int32), lens)]).astype(np.int32)
        row_ctx = np.concatenate([np.asarray([s.offset + 1 for s in dseqs], dtype=np.int32),
                                  rows_p - np.repeat(starts, lens) + np.repeat(np.asarray(cached, dtype=np.int32), lens) + 1]).astype(np.int32)
        seg_lo = np.concatenate([rows_d, np.where(cont, rows_p, np.repeat(starts, lens))]).astype(np.int32)
        seg_hi = np.arange(1, B + N + 1, dtype=np.int32)
        last = np.concatenate([rows_d, starts + np.asarray(lens, dtype=np.int32) - 1]).astype(np.int32)
        chunks = np.asarray([[starts[i], lens[i], cached[i], B + i] for i in range(S) if cached[i] > 0], dtype=np.int32).reshape(-1, 4)
        mb = max(len(s.pages) for s in every)
        table = np.zeros((B + S, mb), np.int32)
        for i, s in enumerate(every):
            table[i, :len(s.pages)] = s.pages
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
        t_ids, t_seq, t_ctx, t_lo, t_hi, t_last, t_table = (dev(x) for x in (ids, row_seq, row_ctx, seg_lo, seg_hi, last, table))
        if B:
            t_ids[:B] = tokens
        V = self.args.vocab_size
        logits = torch.empty((B + S, V), dtype=self.dtype, device=self.device)
        logprobs = torch.empty((B + S, V), dtype=torch.float32, device=self.device)
        nxt = torch.empty(B + S, dtype=torch.int32, device=self.device)
        n = len(self.layers)
        slabs = (C.c_void_p * n)(*[a.slab[i].data_ptr() for i in range(n)])
        self._match_page_format(a)
        lib = _ffi.load()
        head = (self._dec, _ffi.p(t_ids), _ffi.p(t_ctx), _ffi.p(t_seq), _ffi.p(t_lo), _ffi.p(t_hi), _ffi.p(t_last), B + N, B + S)
        tail = (slabs, a.size(), a.slab[0].numel() * a.slab.element_size(), _ffi.p(t_table), mb, _ffi.p(logits), _ffi.p(logprobs), _ffi.p(nxt))
        if B or len(chunks):
            _ffi.check(lib.pie_decoder_step_mixed(*head, B, *tail, len(chunks), chunks.ctypes.data if len(chunks) else None, _ffi.stream()))
        else:
            _ffi.check(lib.pie_decoder_prefill_batch(*head, *tail, _ffi.stream()))
        for s in dseqs:
            s.advance(1)
        for s, k in zip(seqs, lens):
            s.advance(k)
        return nxt, logprobs, logits

    def step_bytes(self, T: int, with_logits: bool = True) -> int:
        """Algorithmic HBM bytes of one decode step at context length T."""
        return int(_ffi.load().pie_decoder_step_bytes(self._dec, int(T), int(with_logits)))

    def launch_kernel(self, name: str, layer: int = 0) -> None:
        """Enqueues ONE launch of the step's sequence (for per-kernel timing); needs a prior step() for valid state."""
        _ffi.check(_ffi.load().pie_decoder_launch_kernel(self._dec, _ffi.KERNELS[name], int(layer), _ffi.stream()))

    def graph_launches(self, with_logits: bool = True) -> int:
        """Kernel nodes of the captured step graph (hipGraphGetNodes); -1 before the first graph-replayed step."""
        return int(_ffi.load().pie_decoder_graph_launches(self._dec, 1 if with_logits else 0))

    def kernel_bytes(self, name: str, T: int) -> int:
        return int(_ffi.load().pie_decoder_kernel_bytes(self._dec, _ffi.KERNELS[name], int(T)))

    def weight_bytes(self) -> int:
        return sum(b.nbytes() for b in self.layers) + self.lm_head.nbytes
