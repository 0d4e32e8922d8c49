"""Host logic of the continuous-batching loop (engine/batch_engine.py) on the CPU: a stub model whose next token is a deterministic function
of the sequence's whole token history, so HOW the scheduler batches a request -- alone, in a prompt pass with others, riding a decode step,
in chunks, behind shared prefix pages -- can never change WHAT it generates.  The page pool is the real native allocator (CPU slab)."""
from collections import Counter

import numpy as np
import pytest
import torch

from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator, PagedKVCache, PagedSequence, TOKEN_CAPACITY_PER_PAGE
from proxy_inference_engine_amd.engine.batch_engine import BatchedEngine

V = 211


def next_token(hist) -> int:
    h = 0
    for i, t in enumerate(hist):
        h = (h * 1000003 + (i + 1) * (int(t) + 7)) & 0xFFFFFFFF
    return (h >> 5) % V


def alone(prompt, max_new, stop):
    hist, out = list(prompt), []
    for _ in range(max_new):
        t = next_token(hist)
        out.append(t)
        if t in stop:
            break
        hist.append(t)
    return out


class StubModel:
    def __init__(self, n_layers: int = 2):
        self.layers = [None] * n_layers
        self.device = torch.device("cpu")
        self.calls = Counter()
        self.max_rows = 0
        self.pool = None

    # ---- the Model surface BatchedEngine uses
    def enable_paged_kv(self, num_pages, kv_dtype=None, kv_scales=None):
        self.pool = PageAllocator(num_pages, 1, 8, dtype=torch.bfloat16, device="cpu")
        if kv_dtype == torch.int8:
            self.pool.dtype = torch.int8   # bookkeeping only (writing the pages' scales needs the GPU): what the engine and this stub branch on
        self.by_first_page = {}
        return self.pool

    def make_cache(self):
        seq = PagedSequence(self.pool, 4)
        return [PagedKVCache(seq, i) for i in range(len(self.layers))]

    def _hist(self, seq):
        if not hasattr(seq, "_hist"):
            # a forked sequence: it shares its first page with the sequence it was forked from, whose history it continues
            assert seq.offset > 0 and seq.pages and seq.pages[0] in self.by_first_page, "a sequence with cached positions nobody wrote"
            seq._hist = list(self.by_first_page[seq.pages[0]][:seq.offset])
        assert len(seq._hist) == seq.offset, "history and cache length disagree"
        return seq._hist

    def _feed(self, cache, ids):
        seq = cache[0].page_manager
        hist = self._hist(seq) if seq.offset else seq.__dict__.setdefault("_hist", [])
        seq.reserve(len(ids))
        hist.extend(int(t) for t in ids)
        seq.advance(len(ids))
        self.by_first_page.setdefault(seq.pages[0], hist)
        return next_token(hist)

    def _out(self, toks):
        nxt = torch.tensor(toks, dtype=torch.int32)
        logprobs = torch.full((len(toks), V), -30.0)
        logprobs[torch.arange(len(toks)), nxt.long()] = 0.0
        return nxt, logprobs, None

    def step(self, ids, cache):
        self.calls["step"] += 1
        tok = self._feed(cache, ids.reshape(-1).tolist())
        nxt, lp, _ = self._out([tok])
        return nxt, lp[0], None

    def step_batch(self, tokens, caches):
        self.calls["step_batch"] += 1
        assert len({id(c[0].page_manager) for c in caches}) == len(caches) == tokens.numel()
        self.max_rows = max(self.max_rows, len(caches))
        return self._out([self._feed(c, [t]) for c, t in zip(caches, tokens.tolist())])

    def prefill_batch(self, prompts, caches):
        self.calls["prefill_batch"] += 1
        assert all(c[0].offset == 0 for c in caches), "prefill_batch takes fresh caches"
        self.max_rows = max(self.max_rows, sum(len(p) for p in prompts))
        return self._out([self._feed(c, p) for c, p in zip(caches, prompts)])

    def step_mixed(self, tokens, decode_caches, prompts, prompt_caches):
        self.calls["step_mixed"] += 1
        assert len(prompts) == len(prompt_caches) and all(len(p) >= 1 for p in prompts)
        assert all(c[0].offset >= 1 for c in decode_caches)
        every = list(decode_caches) + list(prompt_caches)
        assert len({id(c[0].page_manager) for c in every}) == len(every)
        if any(c[0].offset > 0 for c in prompt_caches):
            assert self.pool.dtype != torch.int8, "a prompt continuing a cached prefix needs T pages"
            self.calls["continuing"] += sum(c[0].offset > 0 for c in prompt_caches)
        self.max_rows = max(self.max_rows, len(decode_caches) + sum(len(p) for p in prompts))
        toks = [self._feed(c, [t]) for c, t in zip(decode_caches, tokens.tolist())] if decode_caches else []
        return self._out(toks + [self._feed(c, p) for c, p in zip(prompt_caches, prompts)])


def requests(seed, n, lo=1, hi=150, prefix=()):
    rng = np.random.default_rng(seed)
    return [list(prefix) + rng.integers(0, V, int(k)).tolist() for k in rng.integers(lo, hi, n)]


@pytest.mark.parametrize("kw", [dict(), dict(mixed=False), dict(batch_prefill=False), dict(prefill_chunk=16), dict(prefill_chunk=1), dict(prefill_chunk=300),
                                dict(mixed=False, prefill_chunk=40), dict(max_prefill_rows=64), dict(kv_dtype=torch.int8), dict(kv_dtype=torch.int8, prefill_chunk=16)])
@pytest.mark.parametrize("slots,pages", [(1, 6), (3, 9), (8, 40)])
def test_every_request_gets_the_tokens_it_gets_alone(kw, slots, pages):
    prompts = requests(7, 17)
    stop = {3, 77}
    new = 9
    want = [alone(p, new, stop) for p in prompts]
    model = StubModel()
    eng = BatchedEngine(model, num_pages=pages, max_batch=slots, stop_tokens=stop, **kw)
    assert eng.generate(prompts, new) == want
    assert eng.pool.get_num_free_pages() == eng.pool.size()                          # every page came back
    assert model.max_rows <= max(eng.max_prefill_rows, max(len(p) for p in prompts)) + slots
    if kw.get("prefill_chunk") and kw.get("kv_dtype") != torch.int8:
        assert model.calls["prefill_batch"] == 0 and model.calls["step"] == 0       # every prompt row went through the chunked passes
        assert model.max_rows <= max(kw["prefill_chunk"], slots)                    # a pass never exceeds its row budget (decode rows included)
    if kw.get("kv_dtype") == torch.int8:
        assert model.calls["continuing"] == 0 and eng.prefill_chunk is None          # int8 pages: fresh prompts and decode rows only
    if kw.get("mixed") is False and not kw.get("prefill_chunk"):
        assert model.calls["step_mixed"] == 0
    # a second generate() on the same engine starts clean
    assert eng.generate(prompts[:3], 2) == [alone(p, 2, stop) for p in prompts[:3]]


@pytest.mark.parametrize("kw", [dict(), dict(prefill_chunk=24), dict(mixed=False)])
def test_shared_prefix_pages_are_computed_once_and_change_nothing(kw):
    rng = np.random.default_rng(3)
    system = rng.integers(0, V, 150).tolist()
    prompts = requests(11, 12, lo=1, hi=60, prefix=system)
    prompts[5] = system[:140] + [V - 1, V - 2]                                       # shares only 140 tokens: the common prefix shrinks to two pages
    stop, new = {5}, 7
    want = [alone(p, new, stop) for p in prompts]
    model = StubModel()
    eng = BatchedEngine(model, num_pages=12, max_batch=4, stop_tokens=stop, share_prefix=True, **kw)
    assert eng.generate(prompts, new) == want
    assert eng.shared_pages == 2 and eng.pool.get_num_free_pages() == eng.pool.size()
    assert model.calls["continuing"] >= len(prompts)                                 # every request fed only its suffix, behind the shared pages
    # without sharing the same pool holds fewer requests at once: more passes
    plain = StubModel()
    eng2 = BatchedEngine(plain, num_pages=12, max_batch=4, stop_tokens=stop, **kw)
    assert eng2.generate(prompts, new) == want and eng2.shared_pages == 0
    assert eng.steps <= eng2.steps
    # nothing to share: one request, no common whole page, or int8 pages
    assert BatchedEngine(StubModel(), num_pages=12, max_batch=4, share_prefix=True).generate(prompts[:1], 3) == [alone(prompts[0], 3, set())]
    short = requests(5, 4, lo=70, hi=90)
    e3 = BatchedEngine(StubModel(), num_pages=12, max_batch=4, share_prefix=True)
    assert e3.generate(short, 3) == [alone(p, 3, set()) for p in short] and e3.shared_pages == 0
    e4 = BatchedEngine(StubModel(), num_pages=12, max_batch=4, share_prefix=True, kv_dtype=torch.int8)
    assert e4.generate(prompts[:4], 3) == [alone(p, 3, set()) for p in prompts[:4]] and e4.shared_pages == 0


def test_admission_limits_and_errors():
    model = StubModel()
    eng = BatchedEngine(model, num_pages=4, max_batch=8)
    with pytest.raises(ValueError, match="does not fit"):
        eng.generate([list(range(250))], 10)                                         # 260 positions = 5 pages > 4
    assert eng.generate([[1, 2, 3]], 0) == [[]]
    # three requests of 2 pages each in a 4-page pool: two at a time, the third waits for a retirement
    prompts = requests(2, 3, lo=100, hi=110)
    assert eng.generate(prompts, 4) == [alone(p, 4, set()) for p in prompts]
    assert model.max_rows <= 2 * 110 + 2
    # a sampler replaces the greedy choice: here it picks (argmax + 1) % V, which feeds back into every history
    def sampler(logprobs):
        return (logprobs.argmax(dim=-1) + 1) % V
    eng = BatchedEngine(StubModel(), num_pages=8, max_batch=3, sampler=sampler, prefill_chunk=32)
    got = eng.generate(prompts, 3)

    def alone_sampled(p):
        hist, out = list(p), []
        for _ in range(3):
            t = (next_token(hist) + 1) % V
            out.append(t)
            hist.append(t)
        return out
    assert got == [alone_sampled(p) for p in prompts]
    assert TOKEN_CAPACITY_PER_PAGE == 64
