"""CPU restatement of the reference's Qwen2.5-VL vision tower (models/intern/vision.py).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED with respect to MLX: MLX is not available here and the reference holds no fixtures for this tower, so nothing
pins the rounding points below to MLX's actual kernels.  The ALGORITHM is pinned: in fp32 this restatement agrees to 2e-7 with HF
transformers' independent Qwen2_5_VisionTransformerPretrainedModel on identical weights (tests/test_oracle.py), the model the
reference's vision.py is a port of.  The rounding points follow the same contracts as oracle/pie_oracle.c (every op's result
is rounded to the activation dtype T once; matmuls, softmax, RMSNorm statistics and elementwise formulas run in fp32 inside
an op; a Linear's bias is added to the T-rounded product).  Only tests/ may import this module.

Each function cites the lines of /root/reference/src/proxy_inference_engine/models/intern/vision.py it follows.  The index
logic (position ids, window order, cumulative lengths) is written with explicit loops, independently of the product's
vectorised version, so the two check each other."""
from __future__ import annotations

import math

import numpy as np

from . import pie_oracle as po


def finfo_min(dtype: str) -> float:
    return {"bfloat16": -3.3895313892515355e38, "float16": -65504.0, "float32": -3.4028234663852886e38}[dtype]


def rot_pos_emb(grid_thw, head_dim: int, merge: int) -> np.ndarray:
    """vision.py:245-279 with VisionRotaryEmbedding(head_dim // 2) (:73-85, :242): [N, head_dim / 2] fp32 angles."""
    dim = head_dim // 2
    inv_freq = np.array([1.0 / (10000.0 ** (np.float32(i) / np.float32(dim))) for i in range(0, dim, 2)], dtype=np.float32)
    rows = []
    for t, h, w in grid_thw:
        ids = []
        for bh in range(h // merge):           # reshape(h/m, m, w/m, m).transpose(0, 2, 1, 3).flatten(): merge blocks row-major,
            for bw in range(w // merge):       # then the m x m patches of the block
                for ih in range(merge):
                    for iw in range(merge):
                        ids.append((bh * merge + ih, bw * merge + iw))
        rows.extend(ids * t)                   # mx.tile(stacked_pos_ids, (t, 1))
    out = np.empty((len(rows), 2 * len(inv_freq)), np.float32)
    for n, (hp, wp) in enumerate(rows):
        out[n, :len(inv_freq)] = np.float32(hp) * inv_freq     # rotary_pos_emb_full[pos_ids] -> [N, 2, dim/2] -> reshape(N, -1)
        out[n, len(inv_freq):] = np.float32(wp) * inv_freq
    return out


def get_window_index(grid_thw, window_size: int, merge: int, patch: int):
    """vision.py:281-362: the order that groups merged tokens by window, and the cumulative window lengths in patches."""
    ws = window_size // merge // patch
    unit = merge * merge
    window_index, cu, base = [], [0], 0
    for t, h, w in grid_thw:
        lh, lw = h // merge, w // merge
        pad_h, pad_w = ws - lh % ws, ws - lw % ws
        nh, nw = (lh + pad_h) // ws, (lw + pad_w) // ws
        for ti in range(t):
            for wh in range(nh):
                for ww in range(nw):
                    count = 0
                    for ih in range(ws):
                        for iw in range(ws):
                            r, c = wh * ws + ih, ww * ws + iw
                            if r < lh and c < lw:
                                window_index.append(base + ti * lh * lw + r * lw + c)
                                count += 1
                    cu.append(cu[-1] + count * unit)
        base += t * lh * lw
    return np.array(window_index, np.int64), cu


def rope_vision(x: np.ndarray, angles: np.ndarray, dtype: str) -> np.ndarray:
    """apply_rotary_pos_emb_vision, vision.py:55-70: x [N, H, D] (values in T), angles [N, D/2]."""
    cos = np.tile(np.cos(angles.astype(np.float32)).astype(np.float32)[:, None, :], (1, 1, 2))
    sin = np.tile(np.sin(angles.astype(np.float32)).astype(np.float32)[:, None, :], (1, 1, 2))
    half = x.shape[-1] // 2
    rot = np.concatenate([-x[..., half:], x[..., :half]], axis=-1)       # rotate_half, vision.py:48-52
    return po.round_T((x.astype(np.float32) * cos) + (rot.astype(np.float32) * sin), dtype)


def gelu(x: np.ndarray, dtype: str) -> np.ndarray:
    """nn.GELU() exact form (vision.py:130)."""
    x32 = x.astype(np.float32)
    erf = np.array([math.erf(float(v) * 0.7071067811865476) for v in x32.reshape(-1)], np.float32).reshape(x32.shape)
    return po.round_T(x32 * (np.float32(1.0) + erf) * np.float32(0.5), dtype)


def attention(x, w, p, cu_seqlens, angles, H, dtype):
    """Attention.__call__, vision.py:152-186."""
    N = x.shape[0]
    qkv = po.linear(x, w[p + "attn.qkv.weight"], dtype, w[p + "attn.qkv.bias"]).reshape(N, 3, H, -1)
    D = qkv.shape[-1]
    q, k, v = rope_vision(qkv[:, 0], angles, dtype), rope_vision(qkv[:, 1], angles, dtype), qkv[:, 2]
    mask = np.full((N, N), finfo_min(dtype), np.float32)
    for i in range(1, len(cu_seqlens)):
        mask[cu_seqlens[i - 1]:cu_seqlens[i], cu_seqlens[i - 1]:cu_seqlens[i]] = 0.0
    t = lambda a: np.ascontiguousarray(a.transpose(1, 0, 2))                       # [H, N, D]
    out = po.sdpa(t(q), t(k), t(v), D ** -0.5, mask, dtype, True)
    out = np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(N, -1)
    return po.linear(out, w[p + "attn.proj.weight"], dtype, w[p + "attn.proj.bias"])


def mlp(x, w, p, dtype):
    """MLP.__call__, vision.py:196-197."""
    g = po.linear(x, w[p + "mlp.gate_proj.weight"], dtype, w[p + "mlp.gate_proj.bias"])
    u = po.linear(x, w[p + "mlp.up_proj.weight"], dtype, w[p + "mlp.up_proj.bias"])
    return po.linear(po.silu_mul(g, u, dtype), w[p + "mlp.down_proj.weight"], dtype, w[p + "mlp.down_proj.bias"])


def vision_forward(cfg: dict, w: dict, pixel_values: np.ndarray, grid_thw, dtype: str = "bfloat16", prefix: str = "vision_tower.",
                   want_states: bool = False):
    """VisionModel.__call__, vision.py:364-442.  w: storage-bit arrays keyed like the checkpoint; patch_embed.proj.weight in the
    PyTorch order [out, in, kT, kH, kW].  pixel_values [N, in * kT * kH * kW] values representable in T."""
    merge, unit = cfg["spatial_merge_size"], cfg["spatial_merge_size"] ** 2
    H = cfg["num_heads"]
    head_dim = cfg["hidden_size"] // H
    pw = w[prefix + "patch_embed.proj.weight"]
    x = po.linear(po.round_T(pixel_values, dtype), pw.reshape(pw.shape[0], -1), dtype)      # PatchEmbed (vision.py:110-121): stride = kernel
    angles = rot_pos_emb(grid_thw, head_dim, merge)
    window_index, cu_window = get_window_index(grid_thw, cfg["window_size"], merge, cfg["patch_size"])
    seen, cu_w = set(), []
    for c in cu_window:                                                                      # vision.py:381-390
        if c not in seen:
            seen.add(c)
            cu_w.append(c)
    N = x.shape[0]
    x = x.reshape(N // unit, unit, -1)[window_index].reshape(N, -1)
    angles = angles.reshape(N // unit, unit, -1)[window_index].reshape(N, -1)
    cu_full = [0]
    for t, h, ww in grid_thw:
        for _ in range(t):
            cu_full.append(cu_full[-1] + h * ww)
    states = [x]
    for i in range(cfg["depth"]):
        p = f"{prefix}blocks.{i}."
        cu = cu_full if i in cfg["fullatt_block_indexes"] else cu_w
        x = po.add(x, attention(po.rms_norm(x, w[p + "norm1.weight"], 1e-6, dtype), w, p, cu, angles, H, dtype), dtype)
        x = po.add(x, mlp(po.rms_norm(x, w[p + "norm2.weight"], 1e-6, dtype), w, p, dtype), dtype)
        states.append(x)
    y = po.rms_norm(x, w[prefix + "merger.ln_q.weight"], 1e-6, dtype).reshape(N // unit, -1)   # PatchMerger, vision.py:136-140
    y = po.linear(y, w[prefix + "merger.mlp.0.weight"], dtype, w[prefix + "merger.mlp.0.bias"])
    y = po.linear(gelu(y, dtype), w[prefix + "merger.mlp.2.weight"], dtype, w[prefix + "merger.mlp.2.bias"])
    y = y[np.argsort(window_index, kind="stable")]
    return (y, states) if want_states else y


def synth_vision_checkpoint(cfg: dict, seed: int = 0, dtype: str = "bfloat16", prefix: str = "vision_tower.") -> dict:
    """Random tower weights as storage bits (Linear W ~ N(0, 0.02^2)-scaled to keep activations O(1), norms 1 + N(0, 0.02^2))."""
    rng = np.random.default_rng(seed)
    Hd, I, O = cfg["hidden_size"], cfg["intermediate_size"], cfg["out_hidden_size"]
    unit = cfg["spatial_merge_size"] ** 2
    kin = cfg["in_channels"] * cfg["temporal_patch_size"] * cfg["patch_size"] ** 2
    out = {}

    def lin(name, N, K, bias=True):
        out[prefix + name + ".weight"] = po.to_bits(po.round_T(rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K), dtype), dtype)
        if bias:
            out[prefix + name + ".bias"] = po.to_bits(po.round_T(rng.standard_normal(N, dtype=np.float32) * 0.1, dtype), dtype)

    def norm(name, n):
        out[prefix + name + ".weight"] = po.to_bits(po.round_T(1.0 + 0.02 * rng.standard_normal(n, dtype=np.float32), dtype), dtype)

    pw = po.round_T(rng.standard_normal((Hd, kin), dtype=np.float32) / np.sqrt(kin), dtype)
    out[prefix + "patch_embed.proj.weight"] = po.to_bits(pw, dtype).reshape(Hd, cfg["in_channels"], cfg["temporal_patch_size"],
                                                                             cfg["patch_size"], cfg["patch_size"])
    for i in range(cfg["depth"]):
        p = f"blocks.{i}."
        norm(p + "norm1", Hd), norm(p + "norm2", Hd)
        lin(p + "attn.qkv", 3 * Hd, Hd), lin(p + "attn.proj", Hd, Hd)
        lin(p + "mlp.gate_proj", I, Hd), lin(p + "mlp.up_proj", I, Hd), lin(p + "mlp.down_proj", Hd, I)
    norm("merger.ln_q", Hd)
    lin("merger.mlp.0", Hd * unit, Hd * unit), lin("merger.mlp.2", O, Hd * unit)
    return out
