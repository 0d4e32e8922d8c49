"""InferenceEngine: the decode hot loop on MI355X.

Host-side mirror of engine/inference_engine.py of the reference for the path SURVEY.md 8a names:
generate_step (:228-297), generate (:175-226), make_sampler (:299-317), make_processors (:319-335).
Out of scope here: the tokenizer / chat template, the PSE structuring engine (third-party,
absent) and the Interaction value objects -- `generate` works on token ids; the PSE hooks are optional
injected callables that default to identity.

Array type: torch.Tensor on the ROCm device instead of mx.array.
"""
from __future__ import annotations

from collections.abc import Callable, Generator, Iterator

import torch

from .. import hip_ops
from ..cache import PromptCache
from ..logits_processors import repetition_penalty_logits_processor
from ..models import load
from ..samplers import make_sampler

Sampler = Callable[[torch.Tensor], torch.Tensor]
LogitsProcessor = Callable[[object, torch.Tensor], torch.Tensor]
ModelOutput = tuple[int, dict]


def check_generate_mask(mask, n_heads: int | None = None) -> None:
    """generate_step(mask=array).  The reference forwards ONE array to the prompt pass and to every later single-token call
    (inference_engine.py:246-249), and mx.fast.scaled_dot_product_attention broadcasts it against scores [1, H, L, S] -- L = the prompt
    length, then 1; S growing by one per step.  The only arrays that broadcast against all of those are constant per head: shape
    (), (1,)*k or [.., H | 1, 1, 1].  Such a mask cannot change which keys a query sees: a boolean one must be all True (False hides EVERY
    key of a row: the softmax of an empty row is NaN upstream), an additive one shifts every score of a row by the same finite amount,
    which the softmax cancels.  Accepted and applied as what they are -- nothing; anything else is refused with the reason the reference
    itself would fail with (a broadcast error at the first decode step, or NaN logits)."""
    m = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(mask)
    shape = tuple(m.shape)
    if len(shape) > 4 or any(d != 1 for d in shape[-2:]) or (len(shape) == 4 and shape[0] != 1) or \
            (len(shape) >= 3 and shape[-3] != 1 and (n_heads is None or shape[-3] != n_heads)):
        raise ValueError(f"generate_step(mask=array of shape {shape}): one array is forwarded to the prompt pass and to every single-token step "
                         "(inference_engine.py:246-249), so it must broadcast against scores [1, H, L, S] for every L and S -- shape [.., H | 1, 1, 1]; "
                         "a [L, S] causal mask belongs to Model.__call__(inputs, mask=...) for one call")
    if m.dtype == torch.bool:
        if not bool(m.all()):
            raise ValueError("generate_step(mask=boolean array): a False entry hides every key of its rows (softmax of an empty row)")
    elif not bool(torch.isfinite(m.float()).all()):
        raise ValueError("generate_step(mask=additive array): a non-finite entry hides every key of its rows (softmax of an empty row)")


class InferenceEngine:
    """One model, one PromptCache, not re-entrant -- like the reference (server/app.py:23,35)."""

    def __init__(self, model_path: str | None = None, *, model=None, stop_tokens=(), structuring_engine=None):
        """model_path: local checkpoint directory (models/utils.py layout).  `model=` injects an already built
        proxy_inference_engine_amd.models.llama.Model (synthetic checkpoints, tests, benchmarks)."""
        if model is None:
            if model_path is None:
                raise ValueError("InferenceEngine needs a model_path or a model")
            llm = load(model_path)
            model, self.hf_tokenizer, self.tokenizer_config = llm.model, llm.hf_tokenizer, llm.tokenizer_config
        else:
            self.hf_tokenizer, self.tokenizer_config = None, {}
        self.model = model
        self.prompt_cache = PromptCache()
        self.stop_tokens = set(int(t) for t in stop_tokens)
        self.structuring_engine = structuring_engine  # optional PSE-like object (inference_engine.py:37-41)
        self.samplers: dict[str, Sampler] = {}
        self.logits_processors: dict[str, list[LogitsProcessor]] = {}

    # ------------------------------------------------------------------ samplers / processors
    def make_sampler(self, **kwargs) -> Sampler:
        """inference_engine.py:299-317.  NB the reference's default is temp=1.0 (sampling); greedy needs temp=0."""
        sampler = make_sampler(temp=kwargs.get("temp", 1.0), top_p=kwargs.get("top_p", 1.0), top_k=kwargs.get("top_k", -1),
                               min_p=kwargs.get("min_p", 0.0), min_tokens_to_keep=kwargs.get("min_tokens_to_keep", 1))
        if self.structuring_engine is None:
            return sampler
        wrapped = lambda x: self.structuring_engine.sample(x, sampler)  # noqa: E731
        return wrapped

    def make_processors(self, **kwargs) -> list[LogitsProcessor]:
        """inference_engine.py:319-335: [PSE process_logits] + optional repetition penalty."""
        procs: list[LogitsProcessor] = []
        if self.structuring_engine is not None:
            procs.append(self.structuring_engine.process_logits)
        if kwargs.get("repetition_penalty", 1.0) != 1.0:
            procs.append(repetition_penalty_logits_processor(float(kwargs.get("repetition_penalty", 1.0)),
                                                             int(kwargs.get("context_size", 60))))
        return procs

    def prepare_engine(self, prompt_ids, **inference_kwargs):
        """The sampler / processor half of prepare_engine (inference_engine.py:47-94); tokenisation and the
        state machine are outside this build, so `prompt_ids` are already token ids."""
        self.prompt_cache.load_cached_prompt(prompt_ids)
        self.samplers["root"] = self.make_sampler(**inference_kwargs)
        self.logits_processors["root"] = self.make_processors(**inference_kwargs)
        return prompt_ids

    # ------------------------------------------------------------------ the hot loop
    def generate_step(self, prompt_ids, pixel_values=None, mask=None) -> Iterator[tuple[torch.Tensor, torch.Tensor]]:
        """Yields (next_token_id[1] int32, logprobs[V] fp32) per step, forever (inference_engine.py:228-297).
        Prefill of the non-cached prompt suffix, then one forward per token; all device work is queued
        asynchronously, the consumer synchronises when it reads a token (generate() does, like `.tolist()` :202)."""
        if mask is not None and not (isinstance(mask, str) and mask == "causal"):
            check_generate_mask(mask, getattr(getattr(self.model, "args", None), "num_attention_heads", None))
        if pixel_values is not None and not hasattr(self.model, "get_input_embeddings"):
            raise TypeError("pixel_values need a VLM ensemble (models/intern/ensemble.py: Model) as the engine's model")
        if "root" not in self.samplers:
            self.prepare_engine(prompt_ids, temp=0)
        dev = self.model.device

        def _inference(ids: torch.Tensor, fed_back: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
            """One forward + tail.  fed_back: `ids` is the token the previous call returned (still in the decoder's
            device-side state), so nothing has to be copied or read back."""
            state = "root"
            if self.structuring_engine is not None:
                state = self.structuring_engine.get_current_state() or "root"
                if state not in self.logits_processors:
                    state = "root"
            sampler = self.samplers[state]
            procs = self.logits_processors.get(state) or []
            if pixel_values is not None and not fed_back:
                # the prompt of a VLM request (:246-252): text embeddings with the image features scattered in, through the
                # text tower.  The reference passes pixel_values on every later step too, where a single new token holds
                # no image token and the vision tower's output is discarded (ensemble.py:62-91); those steps skip it here.
                embeds = self.model.get_input_embeddings(ids.reshape(1, -1), pixel_values)
                if not procs:
                    tok, logprobs, _ = self.model.step_embeds(embeds, self.prompt_cache.cache)
                    self.prompt_cache.update(ids)
                    if getattr(sampler, "is_greedy", False):
                        return tok, logprobs
                    return sampler(logprobs[None]).reshape(1).to(torch.int32), logprobs
                logits = self.model.language_model(None, cache=self.prompt_cache.cache, inputs_embeds=embeds)
                last = logits[:, -1, :]
                self.prompt_cache.update(ids)
                for proc in procs:
                    last = proc(self.prompt_cache.computed_ids, last)
                tok, logprobs = hip_ops.logprobs_argmax(last)
                if getattr(sampler, "is_greedy", False):
                    return tok, logprobs
                return sampler(logprobs[None]), logprobs
            if not procs:
                # fused step: hipGraph replay for L == 1, log-softmax (+ greedy argmax) in the HIP tail, no host sync
                greedy_fused = getattr(sampler, "is_greedy", False)
                tok, logprobs, _ = self.model.step(None if (fed_back and greedy_fused) else ids, self.prompt_cache.cache)
                self.prompt_cache.update(ids)                                  # :255 (device ids resolve lazily)
                if greedy_fused:
                    return tok, logprobs
                # stochastic sampler (:271): drawn on the device; the token tensor feeds the next step without a read-back
                return sampler(logprobs[None]).reshape(1).to(torch.int32), logprobs
            logits = self.model(ids[None], cache=self.prompt_cache.cache)      # :252
            last = logits[:, -1, :]                                            # :254
            self.prompt_cache.update(ids)                                      # :255
            for proc in procs:                                                 # :257-266
                last = proc(self.prompt_cache.computed_ids, last)
            tok, logprobs = hip_ops.logprobs_argmax(last)                      # :268-269 + greedy argmax, HIP tail
            if getattr(sampler, "is_greedy", False):
                return tok, logprobs
            return sampler(logprobs[None]), logprobs                           # :271

        if len(self.prompt_cache.cache) == 0:
            self.prompt_cache.create_kv_cache(self.model)                      # :274-275
        todo = self.prompt_cache(prompt_ids)                                   # :277
        host_ids = [int(t) for t in (todo.tolist() if isinstance(todo, torch.Tensor) else todo)]
        next_token, logprobs = _inference(torch.tensor(host_ids, dtype=torch.int32))   # :278 (host ids: no read-back)
        step_count = 0
        while True:
            if step_count > 0:
                next_token, logprobs = _inference(next_token, fed_back=True)   # :288
            yield next_token, logprobs
            step_count += 1

    def generate(self, prompt_ids, **inference_kwargs) -> Generator[ModelOutput, None, str]:
        """Stop-token / max_completion_tokens loop (inference_engine.py:175-226).  Yields (token_id, logprobs_map);
        the generator's return value is the stop reason ("stop" | "length" | "tool_calls")."""
        max_completion_tokens = inference_kwargs.get("max_completion_tokens", -1)
        collect_logprobs = inference_kwargs.get("logprobs", False)
        top_logprobs: int = inference_kwargs.get("top_logprobs", 0)
        logprobs_map: dict[int, float] = {}
        stop_reason = "stop"
        token_count = 0
        for new_tokens, new_logprobs in self.generate_step(prompt_ids):
            token_count += new_tokens.numel()
            if collect_logprobs:
                logprobs_map = get_top_logprobs(new_logprobs, top_logprobs)
            stopped = False
            for token_id in new_tokens.tolist():
                if token_id in self.stop_tokens:
                    stopped = True
                    break
                if collect_logprobs and token_id not in logprobs_map:
                    logprobs_map[token_id] = float(new_logprobs[token_id].item())
                yield token_id, logprobs_map
            if stopped:
                # the reference `break`s only the inner loop and keeps generating (inference_engine.py:204-206);
                # that is an endless loop once a stop token appears, so the outer loop ends here instead
                break
            if self.structuring_engine is not None and self.structuring_engine.has_reached_accept_state:
                stop_reason = "tool_calls"
                break
            if max_completion_tokens > 0 and token_count >= max_completion_tokens:
                stop_reason = "length"
                break
        return stop_reason


def get_top_logprobs(logprobs: torch.Tensor, top_k: int) -> dict[int, float]:
    """engine/utils.py:4-48: the top_k (token -> logprob) pairs, sorted by decreasing logprob."""
    if top_k == 0:
        return {}
    if logprobs.dim() == 2:
        logprobs = logprobs.squeeze(0)
    elif logprobs.dim() != 1:
        raise ValueError(f"Expected 1D or 2D array, got {logprobs.dim()}D")
    top_k = min(top_k, logprobs.shape[0])
    if logprobs.shape[0] == 0:
        return {}
    vals, idx = torch.topk(logprobs, top_k)
    return {int(i): float(v) for i, v in zip(idx.tolist(), vals.tolist())}
