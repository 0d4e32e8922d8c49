"""Qwen2.5-VL vision tower on the HIP ops (host-side mirror of models/intern/vision.py:8-460).

Same structure and index logic as the reference -- PatchEmbed, `rot_pos_emb`, `get_window_index`, the window permutation,
blocks of RMSNorm / attention / RMSNorm / SwiGLU MLP, PatchMerger, the inverse permutation -- with every tensor op on the
device: dense GEMMs on the hand-written 16-bit MFMA kernel (pie_linear_w16m), rotary + q/k/v layout in one kernel, block-diagonal attention on the
MFMA units (full-image layers and 64-patch-window layers are the same kernel with different segment tables), pie_rms_norm,
pie_silu_mul, pie_add, pie_gelu.  Index bookkeeping (positions, window order, cu_seqlens) is host numpy, as it is host
Python in the reference; the row permutations are device gathers.

Layout notes: head_dim (80 for the 7B tower) is zero-padded to the attention kernel's 128 inside q / k / v; `proj`'s weight
is widened to match once at load (zero columns), so no un-padding pass exists.  PatchEmbed's Conv3d (stride = kernel) is a
GEMM over the flattened patches; its weight is flattened once at load in the pixel rows' (C, T, P, P) order."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch

from ... import hip_ops as ops


@dataclass
class VisionConfig:
    """vision.py:8-25 (defaults as there; Qwen2.5-VL-7B: hidden 1280, intermediate 3420, out_hidden 3584, 16 heads)."""
    depth: int = 32
    hidden_size: int = 1280
    intermediate_size: int = 3420
    out_hidden_size: int = 1536
    num_heads: int = 16
    patch_size: int = 14
    in_channels: int = 3
    spatial_merge_size: int = 2
    temporal_patch_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: list[int] = field(default_factory=lambda: [7, 15, 23, 31])


def check_array_shape(shape) -> bool:
    """vision.py:28-45: is a 5-D conv weight already in MLX's [out, kT, kH, kW, in] order?"""
    if len(shape) not in (4, 5):
        return False
    _, out_channels, kH, KW, t = shape
    if t == 3:
        return True
    return out_channels >= kH and out_channels >= KW and kH == KW


class VisionModel:
    def __init__(self, config: VisionConfig, weights: dict[str, torch.Tensor], prefix: str = "vision_tower.",
                 dtype: torch.dtype = torch.bfloat16, device=None):
        self.config = c = config
        self.dtype = dtype
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.spatial_merge_unit = c.spatial_merge_size ** 2
        self.head_dim = c.hidden_size // c.num_heads
        if self.head_dim % 4 or self.head_dim > 128:
            raise ValueError("vision head_dim must be a multiple of 4 and at most 128")
        self.padded_head_dim = 64 if self.head_dim <= 64 else 128

        def get(name):
            t = weights[prefix + name]
            return t.to(device=self.device, dtype=dtype).contiguous()

        w = weights[prefix + "patch_embed.proj.weight"]
        if w.dim() != 5:
            raise ValueError("patch_embed.proj.weight must be a 5-D Conv3d weight")
        if check_array_shape(tuple(w.shape)):   # MLX order [out, kT, kH, kW, in] -> [out, in, kT, kH, kW] (sanitize, vision.py:444-459)
            w = w.permute(0, 4, 1, 2, 3)
        # every Linear of the tower is a dense GEMM on the hand-written 16-bit MFMA kernel (ops.linear_rows -> pie_linear_w16m): the weights are
        # tiled in MFMA operand order once, here (ops.pack_linear); rounds 1-4 called hipBLASLt
        self.patch_w = ops.pack_linear(w.reshape(w.shape[0], -1).to(device=self.device, dtype=dtype).contiguous())   # [hidden, C*T*P*P]
        self.blocks = []
        H, D, DP = c.num_heads, self.head_dim, self.padded_head_dim
        for i in range(c.depth):
            p = f"blocks.{i}."
            proj = get(p + "attn.proj.weight")                                  # [hidden, H*D] -> [hidden, H*DP], zero columns
            if DP != D:
                wide = torch.zeros((proj.shape[0], H, DP), dtype=dtype, device=self.device)
                wide[:, :, :D] = proj.view(proj.shape[0], H, D)
                proj = wide.view(proj.shape[0], H * DP).contiguous()
            self.blocks.append({
                "norm1": get(p + "norm1.weight"), "norm2": get(p + "norm2.weight"),
                "qkv_w": ops.pack_linear(get(p + "attn.qkv.weight")), "qkv_b": get(p + "attn.qkv.bias"),
                "proj_w": ops.pack_linear(proj), "proj_b": get(p + "attn.proj.bias"),
                # gate_proj and up_proj as ONE GEMM [2 * intermediate, hidden] with rows (gate_i, up_i) interleaved: the biases and
                # silu(gate) * up ride in its epilogue (vision.py:196-197), the [N, 2 * intermediate] block is never written
                "gateup_w": ops.pack_linear(ops.interleave_gate_up(get(p + "mlp.gate_proj.weight"), get(p + "mlp.up_proj.weight"))),
                "gateup_b": ops.interleave_gate_up(get(p + "mlp.gate_proj.bias"), get(p + "mlp.up_proj.bias")),
                "down_w": ops.pack_linear(get(p + "mlp.down_proj.weight")), "down_b": get(p + "mlp.down_proj.bias"),
            })
        self._table_cache: dict = {}
        self._graphs: dict = {}
        self.merger = {"ln_q": get("merger.ln_q.weight"), "w0": ops.pack_linear(get("merger.mlp.0.weight")), "b0": get("merger.mlp.0.bias"),
                       "w2": ops.pack_linear(get("merger.mlp.2.weight")), "b2": get("merger.mlp.2.bias")}

    # ------------------------------------------------------------------ vision.py:245-279
    def rot_pos_emb(self, grid_thw) -> np.ndarray:
        """[N, head_dim/2] fp32 angles: each patch's row angles then column angles."""
        m = self.config.spatial_merge_size
        pos = []
        for t, h, w in grid_thw:
            hp = np.repeat(np.arange(h)[:, None], w, axis=1).reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)
            wp = np.repeat(np.arange(w)[None, :], h, axis=0).reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)
            pos.append(np.tile(np.stack([hp, wp], axis=-1), (t, 1)))
        pos = np.concatenate(pos, axis=0)
        dim = self.head_dim // 2                                             # VisionRotaryEmbedding(head_dim // 2), vision.py:242
        inv_freq = (1.0 / (np.float32(10000.0) ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))).astype(np.float32)
        seq = np.arange(int(max(max(h, w) for _, h, w in grid_thw)), dtype=np.float32)
        full = np.outer(seq, inv_freq).astype(np.float32)                    # vision.py:78-85
        return full[pos].reshape(pos.shape[0], -1)

    # ------------------------------------------------------------------ vision.py:281-362
    def get_window_index(self, grid_thw) -> tuple[np.ndarray, list[int]]:
        c = self.config
        window_index, cu = [], [0]
        base = 0
        ws = c.window_size // c.spatial_merge_size // c.patch_size
        for t, h, w in grid_thw:
            lh, lw = h // c.spatial_merge_size, w // c.spatial_merge_size
            index = np.arange(t * lh * lw).reshape(t, lh, lw)
            pad_h, pad_w = ws - lh % ws, ws - lw % ws
            nh, nw = (lh + pad_h) // ws, (lw + pad_w) // ws
            padded = np.pad(index, ((0, 0), (0, pad_h), (0, pad_w)), constant_values=-100)
            padded = padded.reshape(t, nh, ws, nw, ws).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, ws, ws)
            seqlens = (padded != -100).sum(axis=(2, 3)).reshape(-1)
            flat = padded.reshape(-1)
            window_index.append(flat[flat != -100] + base)
            for s in np.cumsum(seqlens) * self.spatial_merge_unit + cu[-1]:
                cu.append(int(s))
            base += t * lh * lw
        return np.concatenate(window_index, axis=0), cu

    # ------------------------------------------------------------------ vision.py:364-442
    def _tables(self, grid: tuple) -> dict:
        """Everything that depends on the grid only (positions, window order, segment tables), computed once per distinct grid
        and kept on the device."""
        t = self._table_cache.get(grid)
        if t is not None:
            return t
        unit = self.spatial_merge_unit
        seq_len = sum(a * h * w for a, h, w in grid)
        angles = self.rot_pos_emb(grid)
        window_index, cu_window = self.get_window_index(grid)
        cu_window = sorted(set(cu_window))                                                     # first occurrences (vision.py:381-390)
        angles = angles.reshape(seq_len // unit, unit, -1)[window_index].reshape(seq_len, -1)
        cu_full = [0]
        for a, h, w in grid:
            for _ in range(a):
                cu_full.append(cu_full[-1] + h * w)
        t = {"seq_len": seq_len,
             "widx": torch.from_numpy(window_index.astype(np.int64)).to(self.device),
             "reverse": torch.from_numpy(np.argsort(window_index, kind="stable").astype(np.int64)).to(self.device),
             "cos": torch.from_numpy(np.cos(angles).astype(np.float32)).to(self.device),
             "sin": torch.from_numpy(np.sin(angles).astype(np.float32)).to(self.device),
             "seg_full": ops.segment_bounds(cu_full, self.device), "seg_win": ops.segment_bounds(cu_window, self.device)}
        if len(self._table_cache) >= 64:
            self._table_cache.pop(next(iter(self._table_cache)))
        self._table_cache[grid] = t
        return t

    def _forward_device(self, x: torch.Tensor, t: dict, output_hidden_states: bool = False):
        """The device half: only launches on the current stream (capturable in a hipGraph)."""
        c = self.config
        seq_len, unit = t["seq_len"], self.spatial_merge_unit
        x = ops.linear_rows(x, self.patch_w)                                                  # PatchEmbed
        x = x.view(seq_len // unit, unit, -1)[t["widx"]].reshape(seq_len, -1).contiguous()
        cos, sin = t["cos"], t["sin"]
        states = (x,) if output_hidden_states else ()
        scale = self.head_dim ** -0.5
        fuse_norm = c.hidden_size % 8 == 0 and c.hidden_size <= 8192
        xn = ops.rms_norm(x, self.blocks[0]["norm1"], 1e-6) if self.blocks else None
        for i, b in enumerate(self.blocks):
            lo, hi = t["seg_full"] if i in c.fullatt_block_indexes else t["seg_win"]
            # hidden_states + attn(norm1(hidden_states))  (vision.py:212-217); xn = norm1(x) comes from the previous fused pass
            qkv = ops.linear_rows(xn, b["qkv_w"])                                           # bias folded into the rotary kernel
            q, k, v = ops.vision_qkv_rope(qkv, cos, sin, c.num_heads, self.padded_head_dim, bias=b["qkv_b"])
            att = ops.sdpa_segments(q, k, v, lo, hi, scale)
            r = ops.linear_rows(att.view(seq_len, -1), b["proj_w"])
            if fuse_norm:
                x, xn = ops.add_bias_rms_norm(x, r, b["proj_b"], b["norm2"], 1e-6)
            else:
                x = ops.add_bias(x, r, b["proj_b"])
                xn = ops.rms_norm(x, b["norm2"], 1e-6)
            # hidden_states + mlp(norm2(hidden_states))  (vision.py:218, :196-197)
            act = ops.linear_rows(xn, b["gateup_w"], b["gateup_b"], swiglu=True, pad_to=64)   # [N, I rounded up to 64 columns, zeros past I]
            r = ops.linear_rows(act, b["down_w"])
            nxt = self.blocks[i + 1]["norm1"] if i + 1 < len(self.blocks) else self.merger["ln_q"]   # the next consumer's norm
            if fuse_norm:
                x, xn = ops.add_bias_rms_norm(x, r, b["down_b"], nxt, 1e-6)
            else:
                x = ops.add_bias(x, r, b["down_b"])
                xn = ops.rms_norm(x, nxt, 1e-6)
            if output_hidden_states:
                states = (*states, x)
        # PatchMerger (vision.py:136-140), then undo the window order (:438-440)
        m = self.merger
        y = (xn if self.blocks else ops.rms_norm(x, m["ln_q"], 1e-6)).view(seq_len // unit, -1)
        y = ops.linear_rows(ops.gelu(ops.linear_rows(y, m["w0"], m["b0"])), m["w2"], m["b2"])
        y = y[t["reverse"]].contiguous()
        return (y, states) if output_hidden_states else y

    # ------------------------------------------------------------------ vision.py:364-442
    def __call__(self, hidden_states: torch.Tensor, grid_thw=None, output_hidden_states: bool | None = None, graph: bool = False):
        """pixel rows [N, C*T*P*P] + grid_thw [[t, h, w], ...] -> image features [N / merge^2, out_hidden_size].
        graph=True replays a hipGraph of the ~15 launches per block, captured on first use of a grid (the second call with
        that grid captures; the features are then a buffer owned by the graph, valid until its next replay)."""
        if grid_thw is None:
            raise ValueError("grid_thw must be provided for the VisionModel forward pass.")
        grid = tuple(tuple(int(v) for v in row) for row in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw))
        x = hidden_states.to(device=self.device, dtype=self.dtype).reshape(-1, self.patch_w.shape[1]).contiguous()
        if x.shape[0] != sum(a * h * w for a, h, w in grid):
            raise ValueError("pixel rows do not match grid_thw")
        t = self._tables(grid)
        if not graph or output_hidden_states:
            return self._forward_device(x, t, bool(output_hidden_states))
        g = self._graphs.get(grid)
        if g is None:
            out = self._forward_device(x, t)                 # first call: eager
            self._graphs[grid] = "warm"
            return out
        if g == "warm":
            static_in = x.clone()
            torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                static_out = self._forward_device(static_in, t)
            g = self._graphs[grid] = (cg, static_in, static_out)
        cg, static_in, static_out = g
        static_in.copy_(x)
        cg.replay()
        return static_out
