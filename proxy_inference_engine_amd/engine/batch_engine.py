"""Continuous batching on top of the multi-sequence decode step (SURVEY.md 8 row f2's "unlocks multi-sequence batching").

The reference declares this shape -- a `Scheduler` that forms `BatchDetails` from prefill- and decode-state sequences over the
paged pool (src/pie_core/include/engine/batch_details.hpp:10-88, scheduler.hpp) -- without a body (`Scheduler::step` is
empty, the Python engine serves one sequence).  This is the smallest complete loop over the pieces that exist here: requests
wait in a queue, join the batch when a slot and enough pages are free -- their prompt rows ride the decode step of the sequences
in flight (`Model.step_mixed`: both kinds of sequence in one pass over the weights, as BatchDetails holds both), or run as a
prompt pass of their own when nothing is decoding -- every step decodes all active sequences with one pass over the weights
(`Model.step_batch`), finished sequences leave and return their pages.  Greedy by default; `sampler` maps a [B, V] log-probability block to B token ids."""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import Callable, Iterable

import torch

from ..cache.kv_cache.paged import TOKEN_CAPACITY_PER_PAGE


@dataclass
class _Active:
    request: int
    cache: list
    token: torch.Tensor           # [1] int32 on the device: the input of the next step
    generated: list = field(default_factory=list)


class BatchedEngine:
    def __init__(self, model, num_pages: int = 1024, max_batch: int = 32, stop_tokens: Iterable[int] = (),
                 sampler: Callable[[torch.Tensor], torch.Tensor] | None = None, batch_prefill: bool = True, max_prefill_rows: int = 4096,
                 kv_dtype: torch.dtype | None = None, kv_scales=None, mixed: bool = True, prefill_chunk: int | None = None, share_prefix: bool = False):
        """kv_dtype=torch.int8 (+ kv_scales = (k, v) float16 [n_layers, n_kv_heads]): the pool holds the reference KVPage's int8 pages with
        per-head scales (page.hpp:25-32) -- half the cache bytes per token, so twice the sequences / context per pool."""
        self.model = model
        self.pool = model.enable_paged_kv(num_pages=num_pages, kv_dtype=kv_dtype, kv_scales=kv_scales)
        self._i8 = self.pool.dtype == torch.int8   # int8 pages are written by the batch paths only: lone prompts go through prefill_batch too
        self.max_batch = max_batch
        self.stop_tokens = set(int(t) for t in stop_tokens)
        self.sampler = sampler
        self.batch_prefill = batch_prefill          # admit several waiting prompts with one pass over the weights
        self.max_prefill_rows = max_prefill_rows    # prompt tokens per such pass
        self.mixed = mixed            # prompts admitted while sequences are decoding join THAT pass (Model.step_mixed) instead of a pass of their own
        # chunked prefill: an admitted prompt is fed at most this many rows per pass (all filling prompts together), every pass also decoding the
        # sequences in flight -- a long prompt no longer stalls them for its whole length.  T pages only (a continuing prompt reads T pages).
        self.prefill_chunk = None if (prefill_chunk is None or self._i8) else max(1, int(prefill_chunk))
        # shared prompt prefix (a system prompt): the whole pages of the requests' longest common prefix are computed ONCE and shared by
        # reference count (KVPage::add_ref, page.hpp:55-68 -- PagedSequence.fork); every request then feeds only its own suffix, as a prompt
        # continuing a cached prefix.  T pages only.
        self.share_prefix = bool(share_prefix) and not self._i8
        self.shared_pages = 0         # pages of the last generate()'s shared prefix
        self.steps = 0                # batched decode steps taken (for throughput accounting)
        self.mixed_passes = 0         # ... of which carried prompt rows as well

    def _pages_for(self, n_tokens: int) -> int:
        return (n_tokens + TOKEN_CAPACITY_PER_PAGE - 1) // TOKEN_CAPACITY_PER_PAGE

    def generate(self, prompts: list, max_new_tokens: int) -> list[list[int]]:
        """Token ids generated for every prompt (in order), at most max_new_tokens each, ending early at a stop token."""
        if max_new_tokens < 1:
            return [[] for _ in prompts]
        for p in prompts:
            if self._pages_for(len(p) + max_new_tokens) > self.pool.size():
                raise ValueError("a prompt does not fit the page pool")
        pending = deque(enumerate(prompts))
        active: list[_Active] = []
        out: list[list[int]] = [[] for _ in prompts]
        reserved = 0                  # pages promised to the active sequences for their full length
        need = {}
        root, P = None, 0             # the sequence holding the shared prefix, its length (whole pages)
        self.shared_pages = 0
        if self.share_prefix and len(prompts) > 1:
            lcp = min(len(p) for p in prompts) - 1                         # every request keeps at least one token of its own
            first = prompts[0]
            for p in prompts[1:]:
                n = 0
                while n < lcp and p[n] == first[n]:
                    n += 1
                lcp = n
            P = lcp // TOKEN_CAPACITY_PER_PAGE * TOKEN_CAPACITY_PER_PAGE
            if P and self._pages_for(P) + self._pages_for(max(len(p) for p in prompts) - P + max_new_tokens) <= self.pool.size():
                root = self.model.make_cache()
                self.model.step_mixed(None, [], [list(first[:P])], [root])    # one pass over the prefix; its pages are shared from here on
                reserved = self.shared_pages = P // TOKEN_CAPACITY_PER_PAGE
            else:
                P = 0

        def new_cache():
            if root is None:
                return self.model.make_cache()
            seq = root[0].page_manager.fork()                               # whole pages: add_ref only, nothing is copied
            return [type(root[0])(seq, i) for i in range(len(root))]
        filling: list = []            # chunked prefill: [request, prompt, cache, rows done] of admitted prompts not yet fully in their pages
        while pending or active or filling:
            # every active sequence holds one token not yet recorded (from its prompt or from the last pass): record, retire
            if active:
                host = torch.cat([a.token for a in active]).tolist()      # one read-back per pass for the stop / length checks
                keep = []
                for a, t in zip(active, host):
                    a.generated.append(int(t))
                    if int(t) in self.stop_tokens or len(a.generated) >= max_new_tokens:
                        out[a.request] = a.generated
                        a.cache[0].page_manager.release()
                        reserved -= need.pop(a.request)
                    else:
                        keep.append(a)
                active = keep
            # admit while there is a slot and the pool can hold the request to its end; the admitted prompts run as ONE pass
            batch = []
            rows = 0
            while pending and len(active) + len(batch) + len(filling) < self.max_batch:
                idx, prompt = pending[0]
                n_pages = self._pages_for(len(prompt) + max_new_tokens) - P // TOKEN_CAPACITY_PER_PAGE
                if reserved + n_pages > self.pool.size() or (rows > 0 and rows + len(prompt) - P > self.max_prefill_rows):
                    break
                pending.popleft()
                need[idx] = n_pages
                reserved += n_pages
                rows += len(prompt) - P
                batch.append((idx, prompt[P:] if P else prompt, new_cache()))
            if self.prefill_chunk:
                filling += [[idx, prompt, cache, 0] for idx, prompt, cache in batch]
                batch = []
            if not active and not batch and not filling:
                if pending:
                    raise RuntimeError("no request fits the page pool")  # unreachable after the check above
                break
            if filling:
                # one pass: every decoding sequence's step + the next rows of the filling prompts, oldest first, prefill_chunk rows in all (decode rows included)
                take, budget = [], max(1, self.prefill_chunk - len(active))   # the pass's rows INCLUDING the decode rows: a chunk of 256 + 8 decode rows would cost the GEMMs a second 256-row tile
                for f in filling:
                    n = min(len(f[1]) - f[3], budget)
                    if n > 0:
                        take.append((f, n))
                        budget -= n
                nxt, logprobs, _ = self.model.step_mixed(torch.cat([a.token for a in active]) if active else None, [a.cache for a in active],
                                                         [f[1][f[3]:f[3] + n] for f, n in take], [f[2] for f, _ in take])
                self.steps += 1
                self.mixed_passes += bool(active)
                nb = len(active)
                if self.sampler is not None:
                    # the sampler sees generation steps only: the decode rows and the rows of prompts whose LAST chunk is in this pass; the row
                    # of a prompt that is still filling is discarded (its greedy id stands in), so a seeded sampler's stream and a stateful
                    # sampler's history do not depend on the chunk size
                    live = list(range(nb)) + [nb + i for i, (f, n) in enumerate(take) if f[3] + n == len(f[1])]
                    if live:
                        idx = torch.tensor(live, dtype=torch.long, device=logprobs.device)
                        nxt = nxt.clone()
                        nxt[idx] = self.sampler(logprobs[idx]).reshape(-1).to(torch.int32)
                for i, a in enumerate(active):
                    a.token = nxt[i:i + 1]
                for i, (f, n) in enumerate(take):
                    f[3] += n
                    if f[3] == len(f[1]):                                 # its last chunk: the row's token is the request's first
                        active.append(_Active(f[0], f[2], nxt[nb + i:nb + i + 1]))
                filling = [f for f in filling if f[3] < len(f[1])]
                continue
            # Worth it while the decode rows do not push the prompt rows into another 256-row GEMM tile: 8 sequences + a prompt of
            # 8 / 64 / 192 / 256 / 384 rows on the 8B model take 0.61 / 0.76 / 0.75 / 0.97 / 0.84 of a prompt pass + a step, but
            # 508 / 2048 rows take 1.08 / 1.06 (516 rows are three 256-row tiles' worth of work for the GEMMs, 508 are two).
            n_mix = len(active) + rows
            if active and batch and self.mixed and (n_mix <= 512 or (n_mix + 255) // 256 == (rows + 255) // 256):
                # the admitted prompts ride the decode step of the sequences in flight: one pass over the weights for both
                nxt, logprobs, _ = self.model.step_mixed(torch.cat([a.token for a in active]), [a.cache for a in active],
                                                         [p for _, p, _ in batch], [c for _, _, c in batch])
                self.steps += 1
                self.mixed_passes += 1
                if self.sampler is not None:
                    nxt = self.sampler(logprobs).reshape(-1).to(torch.int32)
                nb = len(active)
                for i, a in enumerate(active):
                    a.token = nxt[i:i + 1]
                for i, (idx, _, cache) in enumerate(batch):
                    active.append(_Active(idx, cache, nxt[nb + i:nb + i + 1]))
                continue
            joined = []
            if self._i8 and batch and (len(batch) == 1 or not self.batch_prefill):
                for idx, prompt, cache in batch:
                    toks, logprobs, _ = self.model.prefill_batch([prompt], [cache])
                    if self.sampler is not None:
                        toks = self.sampler(logprobs).reshape(-1).to(torch.int32)
                    joined.append(_Active(idx, cache, toks[:1].clone()))
            elif P and batch:                                            # suffixes behind the shared prefix: prompts continuing a cached prefix
                toks, logprobs, _ = self.model.step_mixed(None, [], [p for _, p, _ in batch], [c for _, _, c in batch])
                if self.sampler is not None:
                    toks = self.sampler(logprobs).reshape(-1).to(torch.int32)
                for i, (idx, _, cache) in enumerate(batch):
                    joined.append(_Active(idx, cache, toks[i:i + 1].clone()))
            elif len(batch) == 1 or (batch and not self.batch_prefill):
                for idx, prompt, cache in batch:
                    ids = torch.as_tensor(prompt, dtype=torch.int32).reshape(-1)
                    tok, logprobs, _ = self.model.step(ids.to(self.model.device), cache)
                    if self.sampler is not None:
                        tok = self.sampler(logprobs[None]).reshape(1).to(torch.int32)
                    joined.append(_Active(idx, cache, tok.reshape(1).clone()))
            elif batch:
                toks, logprobs, _ = self.model.prefill_batch([p for _, p, _ in batch], [c for _, _, c in batch])
                if self.sampler is not None:
                    toks = self.sampler(logprobs).reshape(-1).to(torch.int32)
                for i, (idx, _, cache) in enumerate(batch):
                    joined.append(_Active(idx, cache, toks[i:i + 1].clone()))
            if active:
                nxt, logprobs, _ = self.model.step_batch(torch.cat([a.token for a in active]), [a.cache for a in active])
                self.steps += 1
                if self.sampler is not None:
                    nxt = self.sampler(logprobs).reshape(-1).to(torch.int32)
                for i, a in enumerate(active):
                    a.token = nxt[i:i + 1]
            active += joined
        if root is not None:
            root[0].page_manager.release()
        return out
