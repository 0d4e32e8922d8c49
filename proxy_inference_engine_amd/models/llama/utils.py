"""Llama3RoPE frequency table (mirror of models/llama/utils.py:5-50 of the reference).

Only __init__ runs on the host (load time, fp32 like the reference's mx ops); the rotation itself
(mx.fast.rope, utils.py:42-50) is fused into the QKV GEMV epilogue / hip_ops.rope.
NB the reference passes max_position_embeddings for BOTH wavelength thresholds and never reads
original_max_position_embeddings (language.py:64-66); reproduced as is.
"""
from __future__ import annotations

import math

import torch

from ... import hip_ops


class Llama3RoPE:
    def __init__(self, max_embedding_length: int, global_embedding_length: int, dimensions: int, base: float,
                 factor: float, low_freq_factor: float, high_freq_factor: float, traditional: bool = False, device=None):
        self.dimensions = dimensions
        self.max_position_embeddings = max_embedding_length
        self.traditional = traditional
        f32 = torch.float32
        low_freq_wavelen = global_embedding_length / low_freq_factor
        high_freq_wavelen = global_embedding_length / high_freq_factor
        freqs = torch.tensor(base, dtype=f32) ** (torch.arange(0, dimensions, 2, dtype=f32) / dimensions)
        wavelens = 2 * math.pi * freqs
        freqs = torch.where(wavelens > low_freq_wavelen, freqs * factor, freqs)
        is_medium = (wavelens > high_freq_wavelen) & (wavelens < low_freq_wavelen)
        denom = high_freq_factor - low_freq_factor
        if denom != 0:
            smooth = (max_embedding_length / wavelens - low_freq_factor) / denom
            smooth_freqs = freqs / ((1 - smooth) / factor + smooth)
            freqs = torch.where(is_medium, smooth_freqs, freqs)
        # denom == 0 (no rope_scaling): the reference divides 0/0 and masks it out, is_medium is all False
        self.freqs = freqs.to(device=device, dtype=f32).contiguous()

    def __call__(self, x: torch.Tensor, offset: int = 0) -> torch.Tensor:
        return hip_ops.rope(x, self.dimensions, traditional=self.traditional, base=None, scale=1.0, offset=offset, freqs=self.freqs)
