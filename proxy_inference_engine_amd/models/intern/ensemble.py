"""Qwen2-VL-style ensemble around the HIP text tower (host-side mirror of models/intern/ensemble.py:25-121).

What runs here: `get_input_embeddings` (embed_tokens on the device, then the image-token scatter of
`_merge_input_ids_with_image_features`, ensemble.py:62-91) and the text tower on the merged embeddings through
`pie_decoder_prefill_embeds` (LanguageModel(None, cache=cache, inputs_embeds=...), ensemble.py:106-108).  The text tower
is the Llama-shaped decoder with q/k/v bias (models/llama/language.py of this package, QWEN2VL_7B_TEXT shapes).

The vision tower itself (models/intern/vision.py:87-442: patch-embed conv, windowed ViT with 2-D rotary, patch merger) is
SURVEY.md 8 row f3 and is not built: `vision_tower` is a caller-supplied callable (pixel_values, grid_thw) -> features
[N, hidden] on the device.  Without one, a call with pixel_values raises -- there is no CPU or PyTorch fallback."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable

import torch

from ..llama.language import Model as LanguageModel


@dataclass
class ModelArgs:
    """The ensemble-level fields of intern/ensemble.py's config that this path reads."""
    image_token_id: int = 151655
    video_token_id: int = 151656


class Model:
    def __init__(self, config: ModelArgs, language_model: LanguageModel,
                 vision_tower: Callable[[torch.Tensor, torch.Tensor | None], torch.Tensor] | None = None):
        self.config = config
        self.language_model = language_model
        self.vision_tower = vision_tower

    # ------------------------------------------------------------------ ensemble.py:33-60
    def get_input_embeddings(self, input_ids: torch.Tensor, pixel_values: torch.Tensor | None = None,
                             image_grid_thw: torch.Tensor | None = None) -> torch.Tensor:
        """[1, L] ids (+ image) -> [1, L, hidden] embeddings with the image features scattered in."""
        inputs_embeds = self.language_model.embed(input_ids).unsqueeze(0)
        if pixel_values is None:
            return inputs_embeds
        if self.vision_tower is None:
            raise NotImplementedError("no vision tower bound: the HIP vision tower is SURVEY.md 8 row f3 (not built); "
                                      "pass vision_tower= or precomputed features via merge_image_features()")
        hidden_states = self.vision_tower(pixel_values, image_grid_thw)
        if hidden_states.dim() == 2:
            hidden_states = hidden_states[None]
        return self._merge_input_ids_with_image_features(hidden_states, inputs_embeds, input_ids)

    # ------------------------------------------------------------------ ensemble.py:62-91
    def _merge_input_ids_with_image_features(self, image_features: torch.Tensor, inputs_embeds: torch.Tensor,
                                             input_ids: torch.Tensor) -> torch.Tensor:
        """Rows of `inputs_embeds` at the image-token positions (video-token positions when there is no image token) are
        replaced, in order, by the rows of `image_features` (batch 1)."""
        ids = input_ids.reshape(1, -1).to(inputs_embeds.device)
        positions = ids == self.config.image_token_id
        n = int(positions.sum().item())
        if n == 0:
            positions = ids == self.config.video_token_id
            n = int(positions.sum().item())
        if n > 0:
            feats = image_features.to(device=inputs_embeds.device, dtype=inputs_embeds.dtype)
            if feats.dim() == 2:
                feats = feats[None]
            if feats.shape[1] != n:
                # the reference's indexed assignment raises on a shape mismatch as well
                raise ValueError(f"{n} image tokens in the prompt but {feats.shape[1]} image feature rows")
            # ascending positions: what argsort of the 0/1 mask yields for the trailing n entries (ensemble.py:83-86)
            idx = torch.nonzero(positions[0], as_tuple=False).reshape(-1)
            inputs_embeds = inputs_embeds.clone()
            inputs_embeds[0, idx] = feats[0]
        return inputs_embeds

    def merge_image_features(self, input_ids: torch.Tensor, image_features: torch.Tensor) -> torch.Tensor:
        """Embeddings for a prompt whose image features were computed elsewhere."""
        return self._merge_input_ids_with_image_features(image_features, self.language_model.embed(input_ids).unsqueeze(0), input_ids)

    # ------------------------------------------------------------------ ensemble.py:93-108
    def __call__(self, input_ids: torch.Tensor, pixel_values: torch.Tensor | None = None, cache=None, **kwargs) -> torch.Tensor:
        image_grid_thw = kwargs.pop("image_grid_thw", None)
        video_grid_thw = kwargs.pop("video_grid_thw", None)
        grid_thw = image_grid_thw if image_grid_thw is not None else video_grid_thw
        if pixel_values is None:
            return self.language_model(input_ids.reshape(1, -1), cache=cache)
        inputs_embeds = self.get_input_embeddings(input_ids, pixel_values, grid_thw)
        return self.language_model(None, cache=cache, inputs_embeds=inputs_embeds)

    # ------------------------------------------------------------------ ensemble.py:110-121
    @property
    def layers(self):
        return self.language_model.layers

    @property
    def head_dim(self):
        return self.language_model.head_dim

    @property
    def n_kv_heads(self):
        return self.language_model.n_kv_heads

    def make_cache(self):
        return self.language_model.make_cache()

    # the engine's fused fast path (engine/inference_engine.py of this package) talks to the text tower
    @property
    def device(self):
        return self.language_model.device

    def step(self, ids, cache, graph: bool = True):
        return self.language_model.step(ids, cache, graph)

    def step_embeds(self, inputs_embeds, cache):
        return self.language_model.step_embeds(inputs_embeds, cache)
