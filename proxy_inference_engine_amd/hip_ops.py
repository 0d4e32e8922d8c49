"""Op-level host mirror of the MLX ops the reference's decode path calls, on torch (ROCm) tensors.

Same names and argument meaning as the `mx.*` call sites (paths relative to
/root/reference/src/proxy_inference_engine/):
  quantize / dequantize            <- mx.quantize / mx.dequantize   (cache/kv_cache/cache.py:144-147)
  quantized_matmul                 <- mx.quantized_matmul           (via nn.QuantizedLinear, models/utils.py:111)
  rms_norm, rope, scaled_dot_product_attention  <- mx.fast.*        (language.py:137-141, llama/utils.py:42-50, base.py:111-113)
Every function launches hand-written HIP kernels through the C ABI on torch's current stream;
there is no torch-op or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _ffi

U32 = torch.int32  # uint32 code words are carried in int32 tensors (same bits)


def _dev(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise ValueError("pie_hip ops take device tensors (ROCm); got a CPU tensor")
    if not t.is_contiguous():
        raise ValueError("pie_hip ops take contiguous tensors")


def quantize(w: torch.Tensor, group_size: int = 64, bits: int = 4):
    """mx.quantize(w, group_size=64, bits=2|4|6|8): w [N,K] -> (codes [N,K*bits/32] uint32-in-int32, scales, biases [N,K/64])."""
    if group_size != 64 or bits not in (2, 4, 6, 8):
        raise ValueError("only group_size=64 with bits=2, 4, 6 or 8 is implemented")
    _dev(w)
    N, K = w.shape
    if K % 64:
        raise ValueError(f"last dimension must be a multiple of 64, got {K}")
    codes = torch.empty((N, K * bits // 32), dtype=U32, device=w.device)
    scales = torch.empty((N, K // 64), dtype=w.dtype, device=w.device)
    biases = torch.empty_like(scales)
    _ffi.check(_ffi.load().pie_quantize_g64(_ffi.p(w), N, K, bits, _ffi.dtype_code(w.dtype), _ffi.p(codes), _ffi.p(scales),
                                            _ffi.p(biases), _ffi.stream()))
    return codes, scales, biases


def dequantize(codes: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor, group_size: int = 64, bits: int = 4):
    if group_size != 64 or bits not in (4, 8):
        raise ValueError("only group_size=64 with bits=4 or 8 is implemented")
    for t in (codes, scales, biases):
        _dev(t)
    N, K = codes.shape[0], codes.shape[1] * 32 // bits
    out = torch.empty((N, K), dtype=scales.dtype, device=codes.device)
    _ffi.check(_ffi.load().pie_dequantize_g64(_ffi.p(codes), _ffi.p(scales), _ffi.p(biases), N, K, bits,
                                              _ffi.dtype_code(scales.dtype), _ffi.p(out), _ffi.stream()))
    return out


@dataclass
class W4SWeight:
    """One quantised Linear in the W4S streaming layout (include/pie_hip.h)."""
    packed: torch.Tensor          # uint8 [pie_w4s_bytes(N, K)]
    N: int
    K: int
    dtype: torch.dtype
    lin_bias: torch.Tensor | None = None

    @property
    def nbytes(self) -> int:
        return self.packed.numel()


@dataclass
class W8SWeight:
    """One MLX int8 g=64 Linear in the W8S streaming layout (include/pie_hip.h)."""
    packed: torch.Tensor          # uint8 [pie_w8s_bytes(N, K)]
    N: int
    K: int
    dtype: torch.dtype
    lin_bias: torch.Tensor | None = None

    @property
    def nbytes(self) -> int:
        return self.packed.numel()


@dataclass
class W2SWeight:
    """One MLX int2 g=64 Linear in the W2S streaming layout (include/pie_hip.h)."""
    packed: torch.Tensor          # uint8 [pie_w2s_bytes(N, K)]
    N: int
    K: int
    dtype: torch.dtype
    lin_bias: torch.Tensor | None = None

    @property
    def nbytes(self) -> int:
        return self.packed.numel()


@dataclass
class W6SWeight(W2SWeight):
    """One MLX int6 g=64 Linear in the W6S streaming layout (low-nibble plane + high-two-bit plane; include/pie_hip.h)."""


@dataclass
class W4S32Weight:
    """One MLX int4 group-32 Linear in the W4S32 streaming layout (include/pie_hip.h)."""
    packed: torch.Tensor          # uint8 [pie_w4s32_bytes(N, K)]
    N: int
    K: int
    dtype: torch.dtype
    lin_bias: torch.Tensor | None = None

    @property
    def nbytes(self) -> int:
        return self.packed.numel()


@dataclass
class W8S32Weight(W4S32Weight):
    """One MLX int8 group-32 Linear in the W8S32 streaming layout (include/pie_hip.h)."""


def repack_w4s32(codes, scales, biases, row_map: torch.Tensor | None = None, lin_bias=None, bits: int = 4) -> W4S32Weight:
    """Load-time repack of an MLX 4-bit (bits=8: 8-bit) group-32 triplet (weight [N_src, K*bits/32], scales / biases [N_src, K/32]) into W4S32
    (W8S32); row_map as for repack_w4s."""
    for t in (codes, scales, biases):
        _dev(t)
    N_src, K = codes.shape[0], codes.shape[1] * 32 // bits
    if tuple(scales.shape) != (N_src, K // 32) or tuple(biases.shape) != (N_src, K // 32):
        raise ValueError(f"group-32 scales / biases must be [{N_src}, {K // 32}], got {tuple(scales.shape)} / {tuple(biases.shape)}")
    N_out = N_src if row_map is None else int(row_map.numel())
    lib = _ffi.load()
    nbytes = (lib.pie_w8s32_bytes if bits == 8 else lib.pie_w4s32_bytes)(N_out, K)
    if nbytes == 0:
        raise ValueError(f"unsupported shape for W4S32 / W8S32: N={N_out} (must be even), K={K} (multiple of 64)")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=codes.device)
    if row_map is not None:
        row_map = row_map.to(device=codes.device, dtype=torch.int32).contiguous()
    _ffi.check((lib.pie_repack_w8g32 if bits == 8 else lib.pie_repack_w4g32)(_ffi.p(codes.contiguous()), _ffi.p(scales.contiguous()), _ffi.p(biases.contiguous()), N_src, K,
                                                                            _ffi.p(row_map), N_out, _ffi.p(packed), _ffi.stream()))
    return (W8S32Weight if bits == 8 else W4S32Weight)(packed, N_out, K, scales.dtype, lin_bias)


def repack_w8s32(codes, scales, biases, row_map: torch.Tensor | None = None, lin_bias=None) -> W8S32Weight:
    return repack_w4s32(codes, scales, biases, row_map=row_map, lin_bias=lin_bias, bits=8)


def repack_w2s(codes, scales, biases, row_map: torch.Tensor | None = None, lin_bias=None) -> W2SWeight:
    """Load-time repack of an MLX 2-bit group-64 triplet (weight [N_src, K/16], scales / biases [N_src, K/64]) into W2S; row_map as for repack_w4s."""
    for t in (codes, scales, biases):
        _dev(t)
    N_src, K = codes.shape[0], codes.shape[1] * 16
    if tuple(scales.shape) != (N_src, K // 64) or tuple(biases.shape) != (N_src, K // 64):
        raise ValueError(f"group-64 scales / biases must be [{N_src}, {K // 64}], got {tuple(scales.shape)} / {tuple(biases.shape)}")
    N_out = N_src if row_map is None else int(row_map.numel())
    nbytes = _ffi.load().pie_w2s_bytes(N_out, K)
    if nbytes == 0:
        raise ValueError(f"unsupported shape for W2S: N={N_out} (must be even), K={K} (multiple of 64)")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=codes.device)
    if row_map is not None:
        row_map = row_map.to(device=codes.device, dtype=torch.int32).contiguous()
    _ffi.check(_ffi.load().pie_repack_w2g64(_ffi.p(codes.contiguous()), _ffi.p(scales.contiguous()), _ffi.p(biases.contiguous()), N_src, K, _ffi.p(row_map), N_out,
                                            _ffi.p(packed), _ffi.stream()))
    return W2SWeight(packed, N_out, K, scales.dtype, lin_bias)


def repack_w6s(codes, scales, biases, row_map: torch.Tensor | None = None, lin_bias=None) -> W6SWeight:
    """Load-time repack of an MLX 6-bit group-64 triplet (weight [N_src, 3K/16]: MLX's bit stream; scales / biases [N_src, K/64]) into W6S."""
    for t in (codes, scales, biases):
        _dev(t)
    N_src, K = codes.shape[0], codes.shape[1] * 16 // 3
    if codes.shape[1] % 12 or tuple(scales.shape) != (N_src, K // 64) or tuple(biases.shape) != (N_src, K // 64):
        raise ValueError(f"6-bit rows hold 12 words per group of 64; scales / biases must be [{N_src}, {K // 64}], got {tuple(codes.shape)} / {tuple(scales.shape)} / {tuple(biases.shape)}")
    N_out = N_src if row_map is None else int(row_map.numel())
    nbytes = _ffi.load().pie_w6s_bytes(N_out, K)
    if nbytes == 0:
        raise ValueError(f"unsupported shape for W6S: N={N_out} (must be even), K={K} (multiple of 64)")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=codes.device)
    if row_map is not None:
        row_map = row_map.to(device=codes.device, dtype=torch.int32).contiguous()
    _ffi.check(_ffi.load().pie_repack_w6g64(_ffi.p(codes.contiguous()), _ffi.p(scales.contiguous()), _ffi.p(biases.contiguous()), N_src, K, _ffi.p(row_map), N_out,
                                            _ffi.p(packed), _ffi.stream()))
    return W6SWeight(packed, N_out, K, scales.dtype, lin_bias)


def repack_w8s(codes, scales, biases, row_map: torch.Tensor | None = None, lin_bias=None) -> W8SWeight:
    """Load-time repack of an MLX 8-bit triplet (weight [N_src, K/4], scales, biases) into W8S; row_map as for repack_w4s."""
    for t in (codes, scales, biases):
        _dev(t)
    N_src, K = codes.shape[0], codes.shape[1] * 4
    N_out = N_src if row_map is None else int(row_map.numel())
    nbytes = _ffi.load().pie_w8s_bytes(N_out, K)
    if nbytes == 0:
        raise ValueError(f"unsupported shape for W8S: N={N_out} (must be even), K={K} (multiple of 64)")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=codes.device)
    if row_map is not None:
        row_map = row_map.to(device=codes.device, dtype=torch.int32).contiguous()
    _ffi.check(_ffi.load().pie_repack_w8g64(_ffi.p(codes), _ffi.p(scales), _ffi.p(biases), N_src, K, _ffi.p(row_map), N_out,
                                            _ffi.p(packed), _ffi.stream()))
    return W8SWeight(packed, N_out, K, scales.dtype, lin_bias)


def repack_w4s(codes, scales, biases, row_map: torch.Tensor | None = None, lin_bias=None) -> W4SWeight:
    """Load-time repack of an MLX triplet (weight, scales, biases) [N_src,K] into W4S.
    row_map (int32 [N_out]) selects / reorders source rows (q|k|v concatenation, gate/up interleave)."""
    for t in (codes, scales, biases):
        _dev(t)
    N_src, K = codes.shape[0], codes.shape[1] * 8
    N_out = N_src if row_map is None else int(row_map.numel())
    if N_out % 2:
        raise ValueError("W4S needs an even number of rows")
    nbytes = _ffi.load().pie_w4s_bytes(N_out, K)
    if nbytes == 0:
        raise ValueError(f"unsupported shape for W4S: N={N_out}, K={K}")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=codes.device)
    if row_map is not None:
        row_map = row_map.to(device=codes.device, dtype=torch.int32).contiguous()
    _ffi.check(_ffi.load().pie_repack_w4g64(_ffi.p(codes), _ffi.p(scales), _ffi.p(biases), N_src, K, _ffi.p(row_map), N_out,
                                            _ffi.p(packed), _ffi.stream()))
    return W4SWeight(packed, N_out, K, scales.dtype, lin_bias)


@dataclass
class W16SWeight:
    """One dense 16-bit Linear in the W16S streaming layout (include/pie_hip.h)."""
    packed: torch.Tensor          # uint8 [pie_w16s_bytes(N, K)]
    N: int
    K: int
    dtype: torch.dtype
    lin_bias: torch.Tensor | None = None

    @property
    def nbytes(self) -> int:
        return self.packed.numel()


def repack_dense(w: torch.Tensor, row_map: torch.Tensor | None = None, lin_bias=None) -> W16SWeight:
    """Load-time repack of an nn.Linear weight [N_src, K] (bf16 / f16) into W16S; row_map as for repack_w4s."""
    _dev(w)
    if w.dtype not in (torch.bfloat16, torch.float16) or w.dim() != 2:
        raise ValueError("dense weights must be 2-D bfloat16 / float16")
    w = w.contiguous()
    N_src, K = w.shape
    N_out = N_src if row_map is None else int(row_map.numel())
    nbytes = _ffi.load().pie_w16s_bytes(N_out, K)
    if nbytes == 0:
        raise ValueError(f"unsupported shape for W16S: N={N_out} (must be even), K={K} (multiple of 64)")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    if row_map is not None:
        row_map = row_map.to(device=w.device, dtype=torch.int32).contiguous()
    _ffi.check(_ffi.load().pie_repack_dense(_ffi.p(w), N_src, K, _ffi.p(row_map), N_out, _ffi.p(packed), _ffi.stream()))
    return W16SWeight(packed, N_out, K, w.dtype, lin_bias)


def linear(x: torch.Tensor, w: W16SWeight) -> torch.Tensor:
    """nn.Linear.__call__ on a W16S weight: x [..., K] -> [..., N]; fp32 accumulate, one rounding (+ bias in T)."""
    _dev(x)
    if x.shape[-1] != w.K or x.dtype != w.dtype:
        raise ValueError(f"x [..., {x.shape[-1]}] {x.dtype} does not match weight K={w.K} {w.dtype}")
    M = x.numel() // w.K
    y = torch.empty((*x.shape[:-1], w.N), dtype=x.dtype, device=x.device)
    _ffi.check(_ffi.load().pie_gemv_dense(_ffi.p(x), M, _ffi.p(w.packed), w.N, w.K, _ffi.p(w.lin_bias), _ffi.p(y),
                                          _ffi.dtype_code(x.dtype), _ffi.stream()))
    return y


def embedding_dense(ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """nn.Embedding.__call__: ids int32 [L] -> [L, H] rows of the T [V, H] table."""
    _dev(ids)
    _dev(table)
    ids = ids.to(torch.int32).contiguous().view(-1)
    V, H = table.shape
    out = torch.empty((ids.numel(), H), dtype=table.dtype, device=table.device)
    _ffi.check(_ffi.load().pie_embedding_dense(_ffi.p(ids), ids.numel(), _ffi.p(table), V, H, _ffi.dtype_code(table.dtype), _ffi.p(out),
                                               _ffi.stream()))
    return out


def quantized_matmul(x: torch.Tensor, w: "W4SWeight | W8SWeight | W4S32Weight", transpose: bool = True, group_size: int | None = None, bits: int | None = None):
    """mx.quantized_matmul(x, w, scales, biases, transpose=True, group_size=64|32, bits=4|8) on a W4S / W8S / W4S32 weight:
    x [..., K] -> [..., N]; fp32 accumulate, result in x.dtype (+ nn.QuantizedLinear's bias when present)."""
    w_bits = 8 if isinstance(w, (W8SWeight, W8S32Weight)) else (6 if isinstance(w, W6SWeight) else (2 if isinstance(w, W2SWeight) else 4))
    w_group = 32 if isinstance(w, W4S32Weight) else 64
    if not transpose or (group_size is not None and group_size != w_group) or (bits is not None and bits != w_bits):
        raise ValueError("only transpose=True with the weight's own group size and bit width is implemented (the nn.QuantizedLinear form)")
    _dev(x)
    if x.shape[-1] != w.K or x.dtype != w.dtype:
        raise ValueError(f"x [..., {x.shape[-1]}] {x.dtype} does not match weight K={w.K} {w.dtype}")
    M = x.numel() // w.K
    y = torch.empty((*x.shape[:-1], w.N), dtype=x.dtype, device=x.device)
    lib = _ffi.load()
    fn = lib.pie_qgemv_w2g64 if w_bits == 2 else lib.pie_qgemv_w6g64 if w_bits == 6 else (lib.pie_qgemv_w8g32 if w_bits == 8 else lib.pie_qgemv_w4g32) if w_group == 32 else (lib.pie_qgemv_w8g64 if w_bits == 8 else lib.pie_qgemv_w4g64)
    _ffi.check(fn(_ffi.p(x), M, _ffi.p(w.packed), w.N, w.K, _ffi.p(w.lin_bias), _ffi.p(y), _ffi.dtype_code(x.dtype), _ffi.stream()))
    return y


def quantized_matmul_partial(x: torch.Tensor, w: W4SWeight) -> torch.Tensor:
    """The fp32 row sums of quantized_matmul before their rounding to T: x [..., K] -> fp32 [..., N].  Row-parallel shards of a
    tensor-parallel Linear are summed over the ranks in this form (tp.py)."""
    _dev(x)
    if not isinstance(w, W4SWeight) or x.shape[-1] != w.K or x.dtype != w.dtype:
        raise ValueError("quantized_matmul_partial takes a W4S weight matching x's last dimension and dtype")
    M = x.numel() // w.K
    y = torch.empty((*x.shape[:-1], w.N), dtype=torch.float32, device=x.device)
    _ffi.check(_ffi.load().pie_qgemv_w4g64_f32(_ffi.p(x), M, _ffi.p(w.packed), w.N, w.K, _ffi.p(y), _ffi.dtype_code(x.dtype), _ffi.stream()))
    return y


def embedding(ids: torch.Tensor, codes, scales, biases, bits: int = 4, group_size: int = 64) -> torch.Tensor:
    """nn.QuantizedEmbedding.__call__: ids int32 [L] -> [L, H]."""
    _dev(ids)
    ids = ids.to(torch.int32).contiguous().view(-1)
    V, H = codes.shape[0], codes.shape[1] * 32 // bits
    out = torch.empty((ids.numel(), H), dtype=scales.dtype, device=codes.device)
    if group_size == 32:
        _ffi.check(_ffi.load().pie_embedding_g32(_ffi.p(ids), ids.numel(), _ffi.p(codes), _ffi.p(scales), _ffi.p(biases), V, H, bits,
                                                 _ffi.dtype_code(scales.dtype), _ffi.p(out), _ffi.stream()))
        return out
    _ffi.check(_ffi.load().pie_embedding_g64(_ffi.p(ids), ids.numel(), _ffi.p(codes), _ffi.p(scales), _ffi.p(biases), V, H, bits,
                                             _ffi.dtype_code(scales.dtype), _ffi.p(out), _ffi.stream()))
    return out


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """mx.fast.rms_norm(x, weight, eps) over the last axis."""
    _dev(x), _dev(weight)
    H = x.shape[-1]
    y = torch.empty_like(x)
    _ffi.check(_ffi.load().pie_rms_norm(_ffi.p(x), _ffi.p(weight), float(eps), x.numel() // H, H, _ffi.dtype_code(x.dtype),
                                        _ffi.p(y), _ffi.stream()))
    return y


def rope(x: torch.Tensor, dims: int, traditional: bool = False, base=None, scale: float = 1.0, offset: int = 0,
         freqs: torch.Tensor | None = None) -> torch.Tensor:
    """mx.fast.rope(x[..., heads, L, D], dims, traditional, base=None, scale=1.0, offset, freqs)."""
    if base is not None or scale != 1.0 or freqs is None or dims != x.shape[-1]:
        raise ValueError("only the Llama3RoPE form is implemented: base=None, scale=1.0, freqs given, dims=D")
    _dev(x), _dev(freqs)
    L, D = x.shape[-2:]
    y = torch.empty_like(x)
    _ffi.check(_ffi.load().pie_rope_ex(_ffi.p(x), x.numel() // (L * D), L, D, _ffi.p(freqs), int(offset), int(bool(traditional)),
                                       _ffi.dtype_code(x.dtype), _ffi.p(y), _ffi.stream()))
    return y


_sdpa_ws: dict = {}


def scaled_dot_product_attention(q, k, v, scale: float, mask=None, T: int | None = None) -> torch.Tensor:
    """mx.fast.scaled_dot_product_attention for the decode step: q [1,Hq,1,D]; k,v [1,Hkv,cap,D] buffers whose
    first T positions are attended (default: all); mask must be None (L == 1: models/base.py:39-53)."""
    if q.shape[0] != 1:
        raise NotImplementedError("batch 1 only")
    if q.shape[-2] != 1:  # prompt form: mask must be the causal one ("causal"; models/base.py:37-53 builds exactly that)
        if not (isinstance(mask, str) and mask == "causal"):
            raise NotImplementedError("L > 1 needs mask='causal' (the additive causal mask of models/base.py:18-53)")
        for t in (q, k, v):
            _dev(t)
        _, Hq, L, D = q.shape
        Hkv, cap = k.shape[1], k.shape[2]
        total = cap if T is None else int(T)
        qt = q[0].transpose(0, 1).contiguous()                      # [L, Hq, D]
        out = torch.empty_like(qt)
        _ffi.check(_ffi.load().pie_sdpa_prefill(_ffi.p(qt), _ffi.p(k), _ffi.p(v), Hq, Hkv, L, total - L, cap, D, float(scale),
                                                _ffi.dtype_code(q.dtype), _ffi.p(out), _ffi.stream()))
        return out.transpose(0, 1).unsqueeze(0).contiguous()         # [1, Hq, L, D]
    if mask is not None and not (isinstance(mask, str) and mask == "causal"):  # one query row: the causal mask hides nothing
        raise NotImplementedError("decode form: one query position, mask=None")
    for t in (q, k, v):
        _dev(t)
    Hq, D = q.shape[1], q.shape[3]
    Hkv, cap = k.shape[1], k.shape[2]
    T = cap if T is None else int(T)
    key = (q.device, Hq, D)
    ws = _sdpa_ws.get(key)
    if ws is None:
        ws = torch.empty(_ffi.load().pie_sdpa_decode_workspace_bytes(Hq, D), dtype=torch.uint8, device=q.device)
        _sdpa_ws[key] = ws
    out = torch.empty_like(q)
    _ffi.check(_ffi.load().pie_sdpa_decode(_ffi.p(q), _ffi.p(k), _ffi.p(v), Hq, Hkv, T, cap, D, float(scale),
                                           _ffi.dtype_code(q.dtype), _ffi.p(out), _ffi.p(ws), _ffi.stream()))
    return out


def paged_kv_append(k: torch.Tensor, v: torch.Tensor, slab: torch.Tensor, n_pages: int, block_table: torch.Tensor,
                    positions: torch.Tensor) -> None:
    """Stores the new K / V rows [B, Hkv, D] of B sequences in their pages (sequence s at positions[s]; < 0 = idle).
    slab: one layer's page slab (cache/kv_cache/paged.py); block_table int32 [B, max_blocks]; positions int32 [B]."""
    for t in (k, v, slab, block_table, positions):
        _dev(t)
    if block_table.dtype != torch.int32 or positions.dtype != torch.int32:
        raise TypeError("block_table and positions must be int32")
    B, Hkv, D = k.shape
    if v.shape != k.shape or block_table.shape[0] != B or positions.shape[0] != B:
        raise ValueError("paged_kv_append: k, v [B, Hkv, D]; block_table [B, max_blocks]; positions [B]")
    if slab.numel() * slab.element_size() < n_pages * 2 * 64 * Hkv * D * 2:
        raise ValueError("paged_kv_append: slab smaller than n_pages pages")
    _ffi.check(_ffi.load().pie_paged_kv_append(_ffi.p(k), _ffi.p(v), _ffi.p(slab), n_pages, _ffi.p(block_table), block_table.shape[1],
                                               _ffi.p(positions), B, Hkv, D, _ffi.dtype_code(k.dtype), _ffi.stream()))


def paged_attention_decode(q: torch.Tensor, slab: torch.Tensor, n_pages: int, block_table: torch.Tensor, context_lens: torch.Tensor,
                           n_kv_heads: int, scale: float) -> torch.Tensor:
    """One decode query per sequence against its paged KV: q [B, Hq, D]; block_table int32 [B, max_blocks];
    context_lens int32 [B] = positions attended including the current one (0 = idle slot -> zeros).  What the
    reference's Attention::invoke_paged_attention_kernel placeholder stands for (src/layers/attention.cpp:71-83)."""
    for t in (q, slab, block_table, context_lens):
        _dev(t)
    if block_table.dtype != torch.int32 or context_lens.dtype != torch.int32:
        raise TypeError("block_table and context_lens must be int32")
    B, Hq, D = q.shape
    if block_table.shape[0] != B or context_lens.shape[0] != B:
        raise ValueError("paged_attention_decode: block_table [B, max_blocks]; context_lens [B]")
    if slab.numel() * slab.element_size() < n_pages * 2 * 64 * n_kv_heads * D * 2:
        raise ValueError("paged_attention_decode: slab smaller than n_pages pages")
    key = ("paged", q.device, B, Hq, D)
    ws = _sdpa_ws.get(key)
    if ws is None:
        ws = torch.empty(_ffi.load().pie_paged_attn_workspace_bytes(B, Hq, D), dtype=torch.uint8, device=q.device)
        _sdpa_ws[key] = ws
    out = torch.empty_like(q)
    _ffi.check(_ffi.load().pie_paged_attn_decode(_ffi.p(q), _ffi.p(slab), n_pages, _ffi.p(block_table), block_table.shape[1],
                                                 _ffi.p(context_lens), B, Hq, n_kv_heads, D, float(scale), _ffi.dtype_code(q.dtype),
                                                 _ffi.p(out), _ffi.p(ws), _ffi.stream()))
    return out


def _i8_slab_check(slab: torch.Tensor, n_pages: int, Hkv: int, D: int, who: str) -> None:
    if slab.numel() * slab.element_size() < n_pages * _ffi.load().pie_page_i8_bytes(Hkv, D):
        raise ValueError(f"{who}: slab smaller than n_pages int8 pages")


def page_i8_set_scales(slab: torch.Tensor, n_pages: int, n_kv_heads: int, head_dim: int, k_scales: torch.Tensor | None,
                       v_scales: torch.Tensor | None, page_ids: torch.Tensor | None = None) -> None:
    """Writes the per-head fp16 scales of int8 pages (page.hpp:31-32: key_cache_scale_ / value_cache_scale_, [heads, 1]; None = ones, the
    reference constructor's value) into the pages `page_ids` (int32, device; None = all n_pages)."""
    _dev(slab)
    for t in (k_scales, v_scales):
        if t is not None:
            _dev(t)
            if t.dtype != torch.float16 or t.numel() != n_kv_heads:
                raise ValueError("page_i8_set_scales: scales are float16 [n_kv_heads]")
    if page_ids is not None and page_ids.dtype != torch.int32:
        raise TypeError("page_ids must be int32")
    _i8_slab_check(slab, n_pages, n_kv_heads, head_dim, "page_i8_set_scales")
    n = n_pages if page_ids is None else page_ids.numel()
    _ffi.check(_ffi.load().pie_page_i8_set_scales(_ffi.p(slab), n_pages, n_kv_heads, head_dim, _ffi.p(page_ids), n, _ffi.p(k_scales), _ffi.p(v_scales),
                                                  _ffi.stream()))


def paged_kv_append_i8(k: torch.Tensor, v: torch.Tensor, slab: torch.Tensor, n_pages: int, block_table: torch.Tensor, positions: torch.Tensor) -> None:
    """paged_kv_append onto int8 pages: each row is quantised with its page's per-head scale, q = clamp(rint(x / s), -127, 127)."""
    for t in (k, v, slab, block_table, positions):
        _dev(t)
    if block_table.dtype != torch.int32 or positions.dtype != torch.int32:
        raise TypeError("block_table and positions must be int32")
    B, Hkv, D = k.shape
    if v.shape != k.shape or block_table.shape[0] != B or positions.shape[0] != B:
        raise ValueError("paged_kv_append_i8: k, v [B, Hkv, D]; block_table [B, max_blocks]; positions [B]")
    _i8_slab_check(slab, n_pages, Hkv, D, "paged_kv_append_i8")
    _ffi.check(_ffi.load().pie_paged_kv_append_i8(_ffi.p(k), _ffi.p(v), _ffi.p(slab), n_pages, _ffi.p(block_table), block_table.shape[1],
                                                  _ffi.p(positions), B, Hkv, D, _ffi.dtype_code(k.dtype), _ffi.stream()))


def paged_attention_decode_i8(q: torch.Tensor, slab: torch.Tensor, n_pages: int, block_table: torch.Tensor, context_lens: torch.Tensor,
                              n_kv_heads: int, scale: float) -> torch.Tensor:
    """paged_attention_decode over int8 pages: K / V rows are read as fp32(q) * fp32(scale of the page's head), the rest in fp32."""
    for t in (q, slab, block_table, context_lens):
        _dev(t)
    if block_table.dtype != torch.int32 or context_lens.dtype != torch.int32:
        raise TypeError("block_table and context_lens must be int32")
    B, Hq, D = q.shape
    if block_table.shape[0] != B or context_lens.shape[0] != B:
        raise ValueError("paged_attention_decode_i8: block_table [B, max_blocks]; context_lens [B]")
    _i8_slab_check(slab, n_pages, n_kv_heads, D, "paged_attention_decode_i8")
    key = ("paged", q.device, B, Hq, D)
    ws = _sdpa_ws.get(key)
    if ws is None:
        ws = torch.empty(_ffi.load().pie_paged_attn_workspace_bytes(B, Hq, D), dtype=torch.uint8, device=q.device)
        _sdpa_ws[key] = ws
    out = torch.empty_like(q)
    _ffi.check(_ffi.load().pie_paged_attn_decode_i8(_ffi.p(q), _ffi.p(slab), n_pages, _ffi.p(block_table), block_table.shape[1],
                                                    _ffi.p(context_lens), B, Hq, n_kv_heads, D, float(scale), _ffi.dtype_code(q.dtype),
                                                    _ffi.p(out), _ffi.p(ws), _ffi.stream()))
    return out


def silu_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.silu(a) * b (models/llama/language.py:127)."""
    _dev(a), _dev(b)
    y = torch.empty_like(a)
    _ffi.check(_ffi.load().pie_silu_mul(_ffi.p(a), _ffi.p(b), a.numel(), _ffi.dtype_code(a.dtype), _ffi.p(y), _ffi.stream()))
    return y


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _dev(a), _dev(b)
    y = torch.empty_like(a)
    _ffi.check(_ffi.load().pie_add(_ffi.p(a), _ffi.p(b), a.numel(), _ffi.dtype_code(a.dtype), _ffi.p(y), _ffi.stream()))
    return y


def logprobs_argmax(logits: torch.Tensor):
    """engine/inference_engine.py:268-271 + greedy sampler: returns (token int32 [1], logprobs fp32 [V])."""
    _dev(logits)
    logits = logits.reshape(-1)
    V = logits.numel()
    lp = torch.empty(V, dtype=torch.float32, device=logits.device)
    tok = torch.empty(1, dtype=torch.int32, device=logits.device)
    _ffi.check(_ffi.load().pie_logprobs_argmax(_ffi.p(logits), V, _ffi.dtype_code(logits.dtype), _ffi.p(lp), _ffi.p(tok),
                                               _ffi.stream()))
    return tok, lp


def qkv_row_map(n_heads: int, n_kv_heads: int, head_dim: int) -> torch.Tensor:
    n = (n_heads + 2 * n_kv_heads) * head_dim
    arr = (C.c_int32 * n)()
    _ffi.check(_ffi.load().pie_qkv_row_map(n_heads, n_kv_heads, head_dim, arr))
    return torch.tensor(list(arr), dtype=torch.int32)


def gateup_row_map(inter: int) -> torch.Tensor:
    arr = (C.c_int32 * (2 * inter))()
    _ffi.check(_ffi.load().pie_gateup_row_map(inter, arr))
    return torch.tensor(list(arr), dtype=torch.int32)


# ---------------------------------------------------------------------------- vision tower ops
class PackedLinear:
    """An nn.Linear weight [N, K] (16-bit) as W16M tiles: 32 output rows x 64 columns in MFMA operand order, zero-padded
    (csrc/w16_gemm.hpp) -- what the hand-written 16-bit GEMM streams straight into its fragment registers.  Built once per matrix."""

    def __init__(self, weight: torch.Tensor):
        _dev(weight)
        if weight.dim() != 2 or weight.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("PackedLinear: weight [N, K] in bfloat16 or float16")
        weight = weight.contiguous()
        self.N, self.K = (int(v) for v in weight.shape)
        self.shape = (self.N, self.K)  # of the Linear's weight
        self.dtype = weight.dtype
        lib = _ffi.load()
        self.tiles = torch.empty(int(lib.pie_w16m_bytes(self.N, self.K)), dtype=torch.uint8, device=weight.device)
        _ffi.check(lib.pie_repack_w16m(_ffi.p(weight), self.N, self.K, _ffi.dtype_code(weight.dtype), _ffi.p(self.tiles), _ffi.stream()))


def pack_linear(weight: torch.Tensor) -> PackedLinear:
    return PackedLinear(weight)


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """Rows (or entries) of gate_proj and up_proj interleaved (gate_0, up_0, gate_1, ...): the order linear_rows(..., swiglu=True) wants."""
    return torch.stack((gate, up), dim=1).reshape(2 * gate.shape[0], *gate.shape[1:]).contiguous()


def linear_rows(x: torch.Tensor, weight, bias: torch.Tensor | None = None, *, swiglu: bool = False, pad_to: int = 0) -> torch.Tensor:
    """nn.Linear on a block of rows: x [M, K] @ weight [N, K].T (+ bias [N]) -> [M, N]; dense 16-bit weights on the hand-written MFMA GEMM
    (fp32 accumulation; models/intern/vision.py:150-151,192-194,129-133; PatchEmbed's Conv3d with stride = kernel is the same product
    over flattened patches, vision.py:97-121).  weight: a PackedLinear (pack_linear: once per matrix), or the plain [N, K] tensor (packed
    on the spot).  x may carry 64 * ceil(K / 64) columns, zeros past K; with just K columns and K % 64 != 0 it is padded here.
    swiglu: weight rows (and bias) interleave gate and up (interleave_gate_up): returns silu(gate) * up [M, N / 2] (MLP, vision.py:196-197),
    bias and activation in the GEMM's epilogue.  pad_to: the output rows are zero-padded to a multiple of it (the next GEMM's operand)."""
    _dev(x)
    pk = weight if isinstance(weight, PackedLinear) else PackedLinear(weight)
    N, K = pk.N, pk.K
    Kx = -(-K // 64) * 64
    if x.dim() != 2 or x.shape[1] not in (K, Kx) or x.dtype != pk.dtype:
        raise ValueError("linear_rows: x [M, K] (or zero-padded to a multiple of 64 columns), weight [N, K] of one dtype")
    if bias is not None and (bias.shape != (N,) or bias.dtype != x.dtype):
        raise ValueError("linear_rows: bias must be [N] in the activation dtype")
    if swiglu and N % 8:
        raise ValueError("linear_rows: swiglu needs N % 8 == 0 (interleaved gate | up rows)")
    if x.shape[1] != Kx:
        x = torch.nn.functional.pad(x, (0, Kx - K))
    if x.stride(1) != 1 or x.stride(0) % 8 or x.stride(0) < Kx:
        x = x.contiguous()
    M = x.shape[0]
    cols = N // 2 if swiglu else N
    ldy = -(-cols // pad_to) * pad_to if pad_to else cols
    y = (torch.zeros if ldy != cols else torch.empty)((M, ldy), dtype=x.dtype, device=x.device)
    lib = _ffi.load()
    wsb = 0 if swiglu else int(lib.pie_linear_w16m_workspace(M, N, K))
    if wsb and ldy != cols:
        raise ValueError("linear_rows: pad_to is not available for shapes that split K")
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    _ffi.check(lib.pie_linear_w16m(_ffi.p(x), x.stride(0), _ffi.p(pk.tiles), _ffi.p(bias.contiguous() if bias is not None else None), M, N, K,
                                   _ffi.dtype_code(x.dtype), _ffi.p(y), ldy, int(swiglu), _ffi.p(ws), wsb, _ffi.stream()))
    return y


def gelu(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() (exact erf form; PatchMerger, vision.py:130)."""
    _dev(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    _ffi.check(_ffi.load().pie_gelu(_ffi.p(x), x.numel(), _ffi.dtype_code(x.dtype), _ffi.p(y), _ffi.stream()))
    return y


def vision_qkv_rope(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, num_heads: int, padded_head_dim: int | None = None,
                    bias: torch.Tensor | None = None):
    """qkv [N, 3 * H * D] (the qkv Linear's output, vision.py:152-156) -> (q [N, H, DP], k [H, N, DP], v [H, N, DP]) with
    apply_rotary_pos_emb_vision (vision.py:55-70) on q and k; cos / sin fp32 [N, D/2]; head dims D..DP-1 are zero.
    bias [3 * H * D]: the qkv Linear's bias when the GEMM was run without it (added here, same values)."""
    _dev(qkv), _dev(cos), _dev(sin)
    N = qkv.shape[0]
    D = qkv.shape[1] // (3 * num_heads)
    if qkv.dim() != 2 or qkv.shape[1] != 3 * num_heads * D or D % 2:
        raise ValueError("vision_qkv_rope: qkv must be [N, 3 * H * D] with even D")
    DP = padded_head_dim or (64 if D <= 64 else 128)
    if cos.shape != (N, D // 2) or sin.shape != (N, D // 2) or cos.dtype != torch.float32 or sin.dtype != torch.float32:
        raise ValueError("vision_qkv_rope: cos / sin must be float32 [N, D/2]")
    q = torch.empty((N, num_heads, DP), dtype=qkv.dtype, device=qkv.device)
    k = torch.empty((num_heads, N, DP), dtype=qkv.dtype, device=qkv.device)
    v = torch.empty((num_heads, N, DP), dtype=qkv.dtype, device=qkv.device)
    if bias is not None and (bias.shape != (qkv.shape[1],) or bias.dtype != qkv.dtype):
        raise ValueError("vision_qkv_rope: bias must be [3 * H * D] in the activation dtype")
    _ffi.check(_ffi.load().pie_vision_qkv_rope(_ffi.p(qkv.contiguous()), _ffi.p(bias.contiguous() if bias is not None else None), _ffi.p(cos.contiguous()), _ffi.p(sin.contiguous()), N, num_heads, D, DP,
                                               _ffi.dtype_code(qkv.dtype), _ffi.p(q), _ffi.p(k), _ffi.p(v), _ffi.stream()))
    return q, k, v


def segment_bounds(cu_seqlens, device) -> tuple[torch.Tensor, torch.Tensor]:
    """cu_seqlens (host ints, [0, ..., N]) -> per-row key ranges (seg_lo, seg_hi) int32 [N] on the device: the 0 blocks of the
    additive mask vision.py:160-167 builds."""
    import numpy as np
    cu = np.asarray([int(c) for c in cu_seqlens], dtype=np.int64)
    if cu.size < 2 or cu[0] != 0 or np.any(np.diff(cu) < 0):
        raise ValueError("cu_seqlens must start at 0 and be non-decreasing")
    lens = np.diff(cu)
    lo = np.repeat(cu[:-1], lens).astype(np.int32)
    hi = np.repeat(cu[1:], lens).astype(np.int32)
    return torch.from_numpy(lo).to(device), torch.from_numpy(hi).to(device)


def sdpa_segments(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, seg_lo: torch.Tensor, seg_hi: torch.Tensor, scale: float) -> torch.Tensor:
    """mx.fast.scaled_dot_product_attention with the block-diagonal mask of vision.py:160-176: q [N, H, D], k / v [H, N, D]
    (D = 64 or 128), row r attends keys [seg_lo[r], seg_hi[r]) -> [N, H, D]."""
    for t in (q, k, v, seg_lo, seg_hi):
        _dev(t)
    N, H, D = q.shape
    if k.shape != (H, N, D) or v.shape != (H, N, D) or seg_lo.shape != (N,) or seg_hi.shape != (N,):
        raise ValueError("sdpa_segments: q [N, H, D]; k, v [H, N, D]; seg_lo, seg_hi [N]")
    if seg_lo.dtype != torch.int32 or seg_hi.dtype != torch.int32:
        raise TypeError("segment bounds must be int32")
    out = torch.empty_like(q)
    _ffi.check(_ffi.load().pie_sdpa_segments(_ffi.p(q.contiguous()), _ffi.p(k.contiguous()), _ffi.p(v.contiguous()), _ffi.p(seg_lo), _ffi.p(seg_hi),
                                             N, H, D, float(scale), _ffi.dtype_code(q.dtype), _ffi.p(out), _ffi.stream()))
    return out


def bias_silu_mul(gate: torch.Tensor, up: torch.Tensor, bias_gate: torch.Tensor, bias_up: torch.Tensor) -> torch.Tensor:
    """silu(gate + bias_gate) * (up + bias_up) on [M, N] GEMM outputs (MLP of vision.py:196-197, biases folded in).  gate and up
    may be the two column halves of one [M, 2N] GEMM output (equal row strides, unit column stride)."""
    _dev(bias_gate), _dev(bias_up)
    if not (gate.is_cuda and up.is_cuda):
        raise ValueError("pie_hip ops take device tensors (ROCm); got a CPU tensor")
    M, N = gate.shape
    if up.shape != (M, N) or bias_gate.shape != (N,) or bias_up.shape != (N,):
        raise ValueError("bias_silu_mul: gate, up [M, N]; biases [N]")
    if gate.stride(1) != 1 or up.stride(1) != 1 or gate.stride(0) != up.stride(0):
        gate, up = gate.contiguous(), up.contiguous()
    y = torch.empty((M, N), dtype=gate.dtype, device=gate.device)
    _ffi.check(_ffi.load().pie_bias_silu_mul(_ffi.p(gate), _ffi.p(up), _ffi.p(bias_gate.contiguous()), _ffi.p(bias_up.contiguous()),
                                             M, N, gate.stride(0) if M > 1 else N, _ffi.dtype_code(gate.dtype), _ffi.p(y), _ffi.stream()))
    return y


def add_bias(x: torch.Tensor, r: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """x + (r + bias) on [M, N] (residual add with the preceding Linear's bias folded in, vision.py:212-218)."""
    for t in (x, r, bias):
        _dev(t)
    M, N = x.shape
    if r.shape != (M, N) or bias.shape != (N,):
        raise ValueError("add_bias: x, r [M, N]; bias [N]")
    y = torch.empty_like(x)
    _ffi.check(_ffi.load().pie_add_bias(_ffi.p(x.contiguous()), _ffi.p(r.contiguous()), _ffi.p(bias.contiguous()), M, N, _ffi.dtype_code(x.dtype),
                                        _ffi.p(y), _ffi.stream()))
    return y


def add_bias_rms_norm(x: torch.Tensor, r: torch.Tensor, bias: torch.Tensor, norm_weight: torch.Tensor, eps: float):
    """(y, rms_norm(y)) with y = x + (r + bias): the residual add and the norm after it in one pass over the rows."""
    for t in (x, r, bias, norm_weight):
        _dev(t)
    M, N = x.shape
    if r.shape != (M, N) or bias.shape != (N,) or norm_weight.shape != (N,):
        raise ValueError("add_bias_rms_norm: x, r [M, N]; bias, norm_weight [N]")
    y, xn = torch.empty_like(x), torch.empty_like(x)
    _ffi.check(_ffi.load().pie_add_bias_rms_norm(_ffi.p(x.contiguous()), _ffi.p(r.contiguous()), _ffi.p(bias.contiguous()), _ffi.p(norm_weight.contiguous()),
                                                 float(eps), M, N, _ffi.dtype_code(x.dtype), _ffi.p(y), _ffi.p(xn), _ffi.stream()))
    return y, xn


def quantized_matmul_rows(x: torch.Tensor, w: "W4SWeight", w4m: torch.Tensor | None = None) -> torch.Tensor:
    """mx.quantized_matmul in its many-row regime -- weights dequantised to T, T x T products on the MFMA units, fp32
    accumulation -- reading the int4 weights in 4-bit form (pie_qgemm_w4m: the few-row kernels up to 32 rows, the 256 x 256-tile
    prompt GEMM beyond).  x [M, K]; w: the W4S matrix (N % 32 == 0); w4m: its W4M tile copy from `repack_w4m` (built here when absent)."""
    _dev(x)
    if not isinstance(w, W4SWeight):
        raise TypeError("quantized_matmul_rows takes a W4SWeight (int4 g=64)")
    M, K = x.shape
    if K != w.K or M < 1 or w.N % 32:
        raise ValueError("quantized_matmul_rows: x [M, K], N % 32 == 0")
    if w4m is None:
        w4m = repack_w4m(w)
    y = torch.empty((M, w.N), dtype=x.dtype, device=x.device)
    _ffi.check(_ffi.load().pie_qgemm_w4m(_ffi.p(x.contiguous()), _ffi.p(w4m), M, w.N, K, _ffi.dtype_code(x.dtype), _ffi.p(y), _ffi.stream()))
    if w.lin_bias is not None:
        y = add(y, w.lin_bias.expand_as(y).contiguous())
    return y


def repack_w4m(w: "W4SWeight") -> torch.Tensor:
    """W4S stream -> W4M tiles (32 rows x 64 columns, MFMA operand order; include/pie_hip.h), same bytes per weight."""
    lib = _ffi.load()
    n = lib.pie_w4m_bytes(w.N, w.K)
    if n == 0:
        raise ValueError("repack_w4m: N must be a multiple of 32 and K of 64")
    out = torch.empty(n, dtype=torch.uint8, device=w.packed.device)
    _ffi.check(lib.pie_repack_w4s_to_w4m(_ffi.p(w.packed), w.N, w.K, _ffi.p(out), _ffi.stream()))
    return out


SAMPLE_MODES = {"categorical": 0, "top_k": 1, "top_p": 2, "min_p": 3}


def sample(logprobs: torch.Tensor, mode: str, temp: float, p: float = 0.0, k: int = 0, want_mask: bool = False):
    """The stochastic branches of make_sampler (samplers/__init__.py:39-46) as ONE HIP kernel (pie_sample): logprobs fp32 [rows, V] or [V]
    on the GPU -> token ids int32 [rows] on the same device, no host sync.  want_mask: also return (kept_count int32 [rows], kept uint8
    [rows, V]) -- the filter's kept set, for tests."""
    from .samplers import _rng
    x = logprobs
    if not x.is_cuda:
        raise ValueError("the samplers take device tensors (ROCm): there is no host path; got a CPU tensor")
    if x.dim() == 1:
        x = x[None]
    if x.dtype != torch.float32:
        x = x.float()
    x = x.contiguous()
    rows, V = x.shape
    seed, counter = _rng.hip_state(x.device)
    lib = _ffi.load()
    key = (str(x.device), rows, V)
    ws = _sample_ws.get(key)
    if ws is None:  # zeroed once; the kernels leave it ready for the next call
        if len(_sample_ws) > 8:
            _sample_ws.clear()
        ws = _sample_ws[key] = torch.zeros(int(lib.pie_sample_workspace_bytes(rows, V)) // 8, dtype=torch.int64, device=x.device)
    tokens = torch.empty(rows, dtype=torch.int32, device=x.device)
    kept = torch.empty(rows, dtype=torch.int32, device=x.device) if want_mask else None
    mask = torch.empty((rows, V), dtype=torch.uint8, device=x.device) if want_mask else None
    _ffi.check(lib.pie_sample(_ffi.p(x), rows, V, SAMPLE_MODES[mode], float(temp), float(p), int(k), seed, _ffi.p(counter), _ffi.p(ws), _ffi.p(tokens),
                              _ffi.p(kept) if want_mask else None, _ffi.p(mask) if want_mask else None, _ffi.stream()))
    return (tokens, kept, mask) if want_mask else tokens


_sample_ws: dict = {}
