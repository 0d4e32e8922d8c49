"""Paged KV on the GPU: the page pool's slab, pie_paged_kv_append and pie_paged_attn_decode.
The reference has only placeholders for the kernel (src/pie_core/src/layers/attention.cpp:71-83,
src/kernels/paged_attention.metal:6-23), so the parity statement is the domain property: attention over pages equals
attention over the same rows gathered contiguously -- checked against the CPU oracle's sdpa and, bit for bit where the
split geometry coincides, against the contiguous decode kernel."""
import numpy as np
import pytest
import torch

from oracle import pie_oracle as po
from tests._util import assert_dot_close, to_bits, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from proxy_inference_engine_amd import hip_ops
    return hip_ops


def _tdt(dt):
    return torch.bfloat16 if dt == "bfloat16" else torch.float16


def _fill(ops, alloc, rng, lens, Hkv, D, dt, max_blocks):
    """Allocates pages for sequences of the given lengths in an interleaved order (so every table is scattered), appends
    the rows token by token in batches through pie_paged_kv_append; returns (block_table, k_rows, v_rows per sequence)."""
    B = len(lens)
    table = np.zeros((B, max_blocks), np.int32)
    need = [(n + 63) // 64 for n in lens]
    for j in range(max(need)):
        for s in rng.permutation(B):
            if j < need[s]:
                pid = alloc.allocate_page()
                assert pid is not None
                table[s, j] = pid
    ks = [po.round_T(rng.standard_normal((n, Hkv, D)), dt) for n in lens]
    vs = [po.round_T(rng.standard_normal((n, Hkv, D)), dt) for n in lens]
    bt = torch.from_numpy(table).cuda()
    for t in range(max(lens)):
        kb = np.zeros((B, Hkv, D), np.float32)
        vb = np.zeros((B, Hkv, D), np.float32)
        pos = np.full(B, -1, np.int32)
        for s in range(B):
            if t < lens[s]:
                kb[s], vb[s], pos[s] = ks[s][t], vs[s][t], t
        ops.paged_kv_append(to_dev(po.to_bits(kb, dt), dt), to_dev(po.to_bits(vb, dt), dt), alloc.slab[0], alloc.size(), bt,
                            torch.from_numpy(pos).cuda())
    return bt, table, ks, vs


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("Hq,Hkv,D,lens", [
    (8, 2, 128, [1, 63, 64, 65, 300]),
    (32, 8, 128, [777, 5, 1500]),
    (6, 2, 64, [129, 0, 64, 31]),          # an idle slot (context 0) in the batch
    (4, 4, 64, [200]),
    (16, 8, 128, [(7 * i) % 90 for i in range(64)]),   # 64 short sequences x 8 kv-heads: ONE split each, the kernel writes the result itself (no combine launch); idle slots among them
])
def test_paged_attention_vs_oracle(ops, dt, Hq, Hkv, D, lens):
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    rng = np.random.default_rng(Hq * 100 + len(lens))
    max_blocks = (max(lens) + 63) // 64 + 1
    n_pages = sum((n + 63) // 64 for n in lens) + 3
    alloc = PageAllocator(n_pages, Hkv, D, dtype=_tdt(dt), device="cuda")
    alloc.slab.view(torch.int16).fill_(0x7BFF if dt == "float16" else 0x7F00)    # poison: stale rows are huge (f16 65504 / bf16 1.7e38)
    bt, table, ks, vs = _fill(ops, alloc, rng, lens, Hkv, D, dt, max_blocks)
    B = len(lens)
    q = po.round_T(rng.standard_normal((B, Hq, D)), dt)
    got = ops.paged_attention_decode(to_dev(po.to_bits(q, dt), dt), alloc.slab[0], n_pages, bt,
                                     torch.tensor(lens, dtype=torch.int32, device="cuda"), Hkv, D ** -0.5)
    got = to_bits(got)
    for s, n in enumerate(lens):
        if n == 0:
            assert not got[s].any(), "an idle slot must produce zeros"
            continue
        k = np.ascontiguousarray(ks[s].transpose(1, 0, 2))       # [Hkv, n, D]
        v = np.ascontiguousarray(vs[s].transpose(1, 0, 2))
        want = po.sdpa(q[s][:, None, :], k, v, D ** -0.5, None, dt, True, T=n)
        # a weighted mean that cancels to nearly zero has a tiny ulp of its own: the ulp is taken at max(|want|, max / 128)
        assert_dot_close(po.from_bits(got[s], dt), po.round_T(want, dt), dt, max_frac=0.03, what=f"paged seq {s} len {n} {Hq}/{Hkv} D{D} {dt}")
    # the pages hold exactly the appended rows, in the reference's logical [64, heads, head_dim] view (page.hpp:29-30)
    s = int(np.argmax(lens))
    page = alloc.get_page(int(table[s, 0]))
    rows = min(64, lens[s])
    assert np.array_equal(to_bits(page.key_cache()[:rows].contiguous()), po.to_bits(ks[s][:rows], dt))
    assert np.array_equal(to_bits(page.value_cache()[:rows].contiguous()), po.to_bits(vs[s][:rows], dt))


def test_paged_equals_contiguous_kernel_bitwise(ops):
    """Same rows, same split geometry (4 splits): the paged and the contiguous decode kernels must agree bit for bit."""
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    dt, Hq, Hkv, D, n = "bfloat16", 32, 8, 128, 250
    rng = np.random.default_rng(11)
    # 16 x 8 x 4 = 512 workgroups -> 4 splits, the contiguous op's choice for 128 <= T < 512
    lens = [n] * 16
    alloc = PageAllocator(16 * 4 + 1, Hkv, D, device="cuda")
    bt, table, ks, vs = _fill(ops, alloc, rng, lens, Hkv, D, dt, 4)
    q = po.round_T(rng.standard_normal((16, Hq, D)), dt)
    got = ops.paged_attention_decode(to_dev(po.to_bits(q, dt), dt), alloc.slab[0], alloc.size(), bt,
                                     torch.tensor(lens, dtype=torch.int32, device="cuda"), Hkv, D ** -0.5)
    for s in (0, 7, 15):
        k = np.zeros((Hkv, 256, D), np.float32)
        v = np.zeros((Hkv, 256, D), np.float32)
        k[:, :n], v[:, :n] = ks[s].transpose(1, 0, 2), vs[s].transpose(1, 0, 2)
        ref = ops.scaled_dot_product_attention(to_dev(po.to_bits(q[s], dt), dt).view(1, Hq, 1, D), to_dev(po.to_bits(k, dt), dt).view(1, Hkv, 256, D),
                                               to_dev(po.to_bits(v, dt), dt).view(1, Hkv, 256, D), D ** -0.5, T=n)
        assert np.array_equal(to_bits(got[s]).reshape(-1), to_bits(ref).reshape(-1))


def test_shared_prefix_pages(ops):
    """Two sequences sharing their first pages by reference count (add_ref): both read the same rows; freeing one keeps
    the shared pages alive for the other (page_allocator.cpp:81-92)."""
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    dt, Hq, Hkv, D = "bfloat16", 8, 2, 128
    rng = np.random.default_rng(5)
    alloc = PageAllocator(8, Hkv, D, device="cuda")
    bt, table, ks, vs = _fill(ops, alloc, rng, [128], Hkv, D, dt, 3)          # the shared 2-page prefix
    shared = [int(table[0, 0]), int(table[0, 1])]
    for pid in shared:
        alloc.add_ref(pid)
    tail_a, tail_b = alloc.allocate_page(), alloc.allocate_page()
    table2 = np.array([shared + [tail_a], shared + [tail_b]], np.int32)
    bt2 = torch.from_numpy(table2).cuda()
    ka, kb = (po.round_T(rng.standard_normal((1, Hkv, D)), dt) for _ in range(2))
    va, vb = (po.round_T(rng.standard_normal((1, Hkv, D)), dt) for _ in range(2))
    ops.paged_kv_append(to_dev(po.to_bits(np.concatenate([ka, kb]), dt), dt), to_dev(po.to_bits(np.concatenate([va, vb]), dt), dt),
                        alloc.slab[0], alloc.size(), bt2, torch.tensor([128, 128], dtype=torch.int32, device="cuda"))
    q = po.round_T(rng.standard_normal((2, Hq, D)), dt)
    got = to_bits(ops.paged_attention_decode(to_dev(po.to_bits(q, dt), dt), alloc.slab[0], alloc.size(), bt2,
                                             torch.tensor([129, 129], dtype=torch.int32, device="cuda"), Hkv, D ** -0.5))
    for s, (kt, vt) in enumerate(((ka, va), (kb, vb))):
        k = np.ascontiguousarray(np.concatenate([ks[0], kt]).transpose(1, 0, 2))
        v = np.ascontiguousarray(np.concatenate([vs[0], vt]).transpose(1, 0, 2))
        want = po.sdpa(q[s][:, None, :], k, v, D ** -0.5, None, dt, True, T=129)
        assert_dot_close(po.from_bits(got[s], dt), po.round_T(want, dt), dt, max_frac=0.03, what=f"shared prefix seq {s}")
    free0 = alloc.get_num_free_pages()
    for pid in shared + [tail_a]:                                             # sequence A ends
        alloc.free_page(pid)
    assert alloc.get_num_free_pages() == free0 + 1                            # only its private tail came back
    assert all(alloc.get_page(pid).get_ref_count() == 1 for pid in shared)


def test_paged_argument_errors(ops):
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    alloc = PageAllocator(2, 2, 128, device="cuda")
    q = torch.zeros(1, 8, 128, dtype=torch.bfloat16, device="cuda")
    bt = torch.zeros(1, 1, dtype=torch.int32, device="cuda")
    with pytest.raises(TypeError):
        ops.paged_attention_decode(q, alloc.slab[0], 2, bt.long(), torch.ones(1, dtype=torch.int32, device="cuda"), 2, 1.0)
    with pytest.raises(ValueError):
        ops.paged_attention_decode(q, alloc.slab[0], 3, bt, torch.ones(1, dtype=torch.int32, device="cuda"), 2, 1.0)   # slab too small
    with pytest.raises(ValueError):
        ops.paged_attention_decode(q, alloc.slab[0], 2, bt, torch.ones(2, dtype=torch.int32, device="cuda"), 2, 1.0)


# ---------------------------------------------------------------------------- the decoder on paged KV
import json  # noqa: E402


def _tiny(golden_dir, **kw):
    from tests.test_gpu_decode import build
    g = np.load(golden_dir / "tiny_llama_w4_bf16.npz")
    cfg = json.loads(str(g["config_json"]))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    return g, cfg, build(cfg, w, **kw)


def _run(model, cache, prompt, steps):
    """Prompt (batched prefill) then `steps` greedy steps; returns the tokens and every step's logits bits."""
    tok, _, logits = model.step(torch.from_numpy(prompt.astype(np.int32)).cuda(), cache)
    toks, bits = [int(tok.item())], [to_bits(logits).copy()]
    for _ in range(steps):
        tok, _, logits = model.step(None, cache)
        toks.append(int(tok.item()))
        bits.append(to_bits(logits).copy())
    return toks, bits


def test_decoder_on_pages_is_bitwise_the_contiguous_decoder(golden_dir):
    """Same model, same prompt: KV in scattered 64-token pages vs per-layer contiguous buffers.  Both run the same
    kernels on the same rows with the same split geometry (capacity <= 1024 either way), so prompt logits and 200 greedy
    steps -- crossing three page boundaries and one block-table growth (2 -> 4 blocks) -- must agree bit for bit."""
    g, cfg, model = _tiny(golden_dir)
    prompt = g["prompt"]
    want_t, want_b = _run(model, model.make_cache(), prompt, 200)
    pool = model.enable_paged_kv(num_pages=24, max_blocks=2)
    for _ in range(5):                       # scatter: the sequence's pages will not be 0, 1, 2, ...
        pool.allocate_page()
    pool.free_page(1), pool.free_page(3)
    cache = model.make_cache()
    got_t, got_b = _run(model, cache, prompt, 200)
    assert got_t == want_t
    for i, (a, b) in enumerate(zip(got_b, want_b)):
        assert np.array_equal(a, b), f"logits differ at step {i}"
    seq = cache[0].page_manager
    n = len(prompt) + 200
    assert seq.offset == n and len(seq.pages) == (n + 63) // 64 and seq.pages[:2] == [3, 1]
    assert seq.max_blocks >= len(seq.pages) > 2
    assert [pool.get_page(p).num_tokens() for p in seq.pages] == [64] * (n // 64) + ([n % 64] if n % 64 else [])
    # the reference's protocol on top: state gathers the rows, trim / reuse give pages back
    k0, v0 = cache[0].state
    assert k0.shape == (1, cfg["num_key_value_heads"], n, cfg["hidden_size"] // cfg["num_attention_heads"])
    free = pool.get_num_free_pages()
    assert cache[0].trim(n - 70) == n - 70 and seq.offset == 70 and len(seq.pages) == 2
    assert pool.get_num_free_pages() == free + (n + 63) // 64 - 2


def test_decoder_on_pages_long_context_forced_splits(golden_dir):
    """Past capacity 1024 the attention plan switches to many splits + the combine launch; with the split count pinned
    (kv_splits) the paged and contiguous decoders still see identical geometry: 1100 positions, bitwise."""
    g, cfg, model = _tiny(golden_dir, kv_splits=8)
    rng = np.random.default_rng(3)
    prompt = rng.integers(0, cfg["vocab_size"], 1050).astype(np.int32)
    want_t, want_b = _run(model, model.make_cache(), prompt, 50)
    model.enable_paged_kv(num_pages=40, max_blocks=4)
    got_t, got_b = _run(model, model.make_cache(), prompt, 50)
    assert got_t == want_t
    assert all(np.array_equal(a, b) for a, b in zip(got_b, want_b))


def test_engine_generates_on_pages_with_prefix_reuse(golden_dir):
    """InferenceEngine end to end on the paged cache: golden greedy tokens, then a second request sharing the prompt:
    PromptCache.reuse_cache trims to the common prefix (pages behind it return to the pool) and one token is re-processed."""
    from proxy_inference_engine_amd import InferenceEngine
    from tests.test_gpu_decode import margin_bound
    g, cfg, model = _tiny(golden_dir)
    pool = model.enable_paged_kv(num_pages=16)
    eng = InferenceEngine(model=model)
    eng.prepare_engine(g["prompt"], temp=0)
    gen = eng.generate_step(torch.from_numpy(g["prompt"]))
    mb = margin_bound(po.from_bits(g["prefill_last_logits"], "bfloat16"))
    n = len(g["tokens"])
    safe = int(np.argmax(g["margins"] < mb)) if (g["margins"] < mb).any() else n
    first = None
    for i in range(n):
        tok, _ = next(gen)
        first = int(tok.item()) if i == 0 else first
        if i < safe:
            assert int(tok.item()) == int(g["tokens"][i]), f"step {i}"
    from proxy_inference_engine_amd.cache.kv_cache import PagedKVCache
    assert isinstance(eng.prompt_cache.cache[0], PagedKVCache)
    used = pool.size() - pool.get_num_free_pages()
    assert used == (len(g["prompt"]) + n - 1 + 63) // 64
    again = next(eng.generate_step(torch.from_numpy(g["prompt"])))[0]
    assert int(again.item()) == first
    assert eng.prompt_cache.cache[0].offset == len(g["prompt"])
    assert pool.size() - pool.get_num_free_pages() == (len(g["prompt"]) + 63) // 64


def test_pool_exhaustion_is_an_error_not_a_fault(golden_dir):
    g, cfg, model = _tiny(golden_dir)
    model.enable_paged_kv(num_pages=2)
    cache = model.make_cache()
    with pytest.raises(RuntimeError, match="exhausted"):
        model.step(torch.zeros(200, dtype=torch.int32, device="cuda"), cache)
    assert cache[0].offset == 0 and cache[0].page_manager.allocator.get_num_free_pages() == 2


def test_forked_sequence_shares_full_pages(golden_dir):
    """fork(): full pages shared by reference count, the partial page copied; both continuations produce what an
    unshared sequence with the same history produces."""
    from proxy_inference_engine_amd.cache.kv_cache import PagedKVCache
    g, cfg, model = _tiny(golden_dir)
    rng = np.random.default_rng(9)
    prompt = rng.integers(0, cfg["vocab_size"], 150).astype(np.int32)            # 2 full pages + 22 rows
    pool = model.enable_paged_kv(num_pages=16)
    a = model.make_cache()
    model.step(torch.from_numpy(prompt).cuda(), a)
    seq_b = a[0].page_manager.fork()
    b = [PagedKVCache(seq_b, i) for i in range(len(a))]
    assert seq_b.pages[:2] == a[0].page_manager.pages[:2] and seq_b.pages[2] != a[0].page_manager.pages[2]
    assert all(pool.get_page(p).get_ref_count() == 2 for p in seq_b.pages[:2])
    nxt_a = torch.tensor([5], dtype=torch.int32, device="cuda")
    nxt_b = torch.tensor([9], dtype=torch.int32, device="cuda")
    la = to_bits(model.step(nxt_a, a)[2]).copy()
    lb = to_bits(model.step(nxt_b, b)[2]).copy()
    for tail, got in ((5, la), (9, lb)):
        ref = model.make_cache()
        model.step(torch.from_numpy(prompt).cuda(), ref)
        want = to_bits(model.step(torch.tensor([tail], dtype=torch.int32, device="cuda"), ref)[2]).copy()
        assert np.array_equal(got, want)
        ref[0].page_manager.release()
    seq_b.release()
    assert all(pool.get_page(p).get_ref_count() == 1 for p in a[0].page_manager.pages[:2])


# ---------------------------------------------------------------------------- multi-sequence decode step (continuous batching)
def _oracle_seq(cfg, w, prompt, steps, regime_rows):
    """Oracle run of one sequence: prompt, then `steps` teacher-forced steps along its own greedy choices; the batched step
    multiplies B rows at once, so its Linears are in MLX's many-row regime from 6 sequences on (oracle qmm_min_rows)."""
    orc = po.OracleLlama(cfg, w, "bfloat16")
    cache = [po.OracleKVCache() for _ in orc.layers]
    logits = [orc.forward(prompt, cache)[-1]]
    po.set_qmm_min_rows(regime_rows)
    try:
        for _ in range(steps):
            logits.append(orc.forward(np.array([int(np.argmax(logits[-1]))]), cache)[0])
    finally:
        po.set_qmm_min_rows(6)
    return logits


@pytest.mark.parametrize("B", [2, 3, 5, 8, 32, 40])
def test_batched_decode_step_matches_each_sequence_alone(golden_dir, B):
    """B sequences of different lengths decode together (pie_decoder_step_batch): the weights stream once per step, every row
    has its own position, pages and attention span.  Each row's logits must be what the oracle gives for that sequence alone
    (teacher-forced along the oracle's greedy tokens, margins permitting), with the oracle in ITS OWN default regime for that
    many rows: below 6 rows row-by-row exact fp32 (MLX's qmv; here the multi-row streaming GEMV k_w4s_gemv_rows), from 6 rows the
    many-row regime (weights dequantised to T, MFMA: the few-row kernel up to 32 rows, the many-row GEMM beyond)."""
    from tests._util import assert_vec_close
    from tests.test_gpu_decode import margin_bound
    g, cfg, model = _tiny(golden_dir)
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    rng = np.random.default_rng(B)
    steps = 4
    lens = [int(rng.integers(7, 150)) for _ in range(B)]
    lens[0], lens[-1] = 63, 64                                   # a sequence that crosses a page boundary during the steps, one that just did
    prompts = [rng.integers(0, cfg["vocab_size"], n).astype(np.int32) for n in lens]
    pool = model.enable_paged_kv(num_pages=4 * B + 4, max_blocks=2)
    caches, toks = [], []
    for p in prompts:
        c = model.make_cache()
        tok, _, _ = model.step(torch.from_numpy(p).cuda(), c)
        caches.append(c)
        toks.append(int(tok.item()))
    # the oracle multiplies ONE sequence at a time: it is told which regime B rows would be in (its switch sits at 6 rows); it is slow: the first six rows
    want = [_oracle_seq(cfg, w, p, steps, 6 if B < 6 else 1) for p in prompts[:6]]
    alive = [True] * len(want)
    tokens = torch.tensor(toks, dtype=torch.int32, device="cuda")
    for st in range(steps):
        # teacher-force the oracle's greedy token where we compare, the model's own elsewhere
        feed = tokens.clone()
        for i in range(len(want)):
            feed[i] = int(np.argmax(want[i][st]))
        tokens, logprobs, logits = model.step_batch(feed, caches)
        assert tokens.shape == (B,) and logprobs.shape == (B, cfg["vocab_size"])
        got = logits.float().cpu().numpy()
        for i in range(len(want)):
            assert_vec_close(got[i], want[i][st + 1], "bfloat16", what=f"B={B} step {st} row {i} (len {lens[i]})")
            top2 = np.sort(want[i][st + 1])[-2:]
            if top2[1] - top2[0] > margin_bound(want[i][st + 1]):
                assert int(tokens[i].item()) == int(np.argmax(want[i][st + 1]))
        lp = logprobs.double().exp().sum(dim=1).cpu().numpy()
        assert np.all(np.abs(lp - 1.0) < 1e-4)
    for c, n in zip(caches, lens):
        assert c[0].offset == n + steps
    assert pool.size() - pool.get_num_free_pages() == sum((n + steps + 63) // 64 for n in lens)


def test_batched_step_equals_single_sequence_batched_path(golden_dir, knobs):
    """A batch of one sequence against a one-token prompt on a contiguous cache (knob prefill_min at its minimum):
    same Linears, different attention kernel (paged split-KV decode vs the MFMA prompt kernel) -- equal within rounding."""
    from tests._util import assert_vec_close
    g, cfg, model = _tiny(golden_dir)
    prompt = g["prompt"].astype(np.int32)
    ref_cache = model.make_cache()
    model.step(torch.from_numpy(prompt).cuda(), ref_cache)
    knobs("prefill_min", 1)
    want = model(torch.tensor([[77]]).cuda(), cache=ref_cache)[0, -1]
    knobs("prefill_min", None)
    model.enable_paged_kv(num_pages=8)
    c = model.make_cache()
    model.step(torch.from_numpy(prompt).cuda(), c)
    _, _, logits = model.step_batch(torch.tensor([77], dtype=torch.int32), [c])
    assert_vec_close(logits[0].float().cpu().numpy(), want.float().cpu().numpy(), "bfloat16", what="batch of one")
    with pytest.raises(TypeError):
        model.step_batch(torch.tensor([1]), [ref_cache])


def test_continuous_batching_engine(golden_dir):
    """BatchedEngine: ten requests, at most four in flight, a pool too small for all of them at once -- requests wait, join as
    others retire, every one finishes with the tokens it gets when served alone (same kernels; only the attention split
    geometry varies with the batch, so low-margin steps may differ: first tokens exact, overall agreement high), stop tokens
    end a request early, and every page returns to the pool."""
    from proxy_inference_engine_amd.engine import BatchedEngine
    g, cfg, model = _tiny(golden_dir)
    rng = np.random.default_rng(21)
    prompts = [rng.integers(0, cfg["vocab_size"], int(n)).tolist() for n in rng.integers(3, 90, 10)]
    new = 6
    eng = BatchedEngine(model, num_pages=7, max_batch=4)
    got = eng.generate(prompts, new)
    assert [len(t) for t in got] == [new] * 10
    assert eng.pool.get_num_free_pages() == eng.pool.size()
    assert 0 < eng.steps < 10 * new                       # fewer batched steps than tokens: sequences really shared steps
    alone = BatchedEngine(model, num_pages=7, max_batch=1).generate(prompts, new)
    assert [t[0] for t in got] == [t[0] for t in alone]
    same = sum(a == b for x, y in zip(got, alone) for a, b in zip(x, y))
    assert same >= 0.85 * 10 * new, f"{same} of {10 * new} tokens agree"
    # a stop token ends its request (the token is reported, like the reference's loop does before breaking)
    stop = got[3][2]
    eng2 = BatchedEngine(model, num_pages=7, max_batch=4, stop_tokens=[stop])
    got2 = eng2.generate(prompts, new)
    assert got2[3] == got[3][:got[3].index(stop) + 1]
    assert all(len(t) <= new for t in got2)
    # the request that took the freed slot joined while three others were decoding: its prompt rode their step (Model.step_mixed);
    # with mixed=False it gets a prompt pass of its own -- same tokens up to low-margin steps (a different GEMM row count)
    assert eng2.mixed_passes > 0
    plain = BatchedEngine(model, num_pages=7, max_batch=4, stop_tokens=[stop], mixed=False)
    got_plain = plain.generate(prompts, new)
    assert plain.mixed_passes == 0 and [len(t) for t in got_plain] == [len(t) for t in got2]
    n_tok = sum(len(t) for t in got2)
    assert sum(a == b for x, y in zip(got2, got_plain) for a, b in zip(x, y)) >= 0.85 * n_tok
    # chunked prefill: admitted prompts are fed 24 rows per pass while the sequences in flight keep decoding (prompts continuing their own
    # cached prefix: pie_decoder_step_mixed's chunks) -- every request completes with (up to low-margin steps) the same tokens, pages drain
    chunked = BatchedEngine(model, num_pages=7, max_batch=4, stop_tokens=[stop], prefill_chunk=24)
    got_c = chunked.generate(prompts, new)
    assert [len(t) for t in got_c] == [len(t) for t in got2] and chunked.pool.get_num_free_pages() == chunked.pool.size()
    assert chunked.steps > eng2.steps                      # the prompts took several passes each
    assert sum(a == b for x, y in zip(got2, got_c) for a, b in zip(x, y)) >= 0.85 * n_tok
    with pytest.raises(ValueError, match="does not fit"):
        BatchedEngine(model, num_pages=2, max_batch=2).generate([list(range(200))], 4)


def test_engine_shares_the_pages_of_a_common_prompt_prefix(golden_dir):
    """BatchedEngine(share_prefix=True): eight requests start with the same 140 tokens (a system prompt).  The two whole pages of that prefix are
    computed once and shared by reference count (KVPage::add_ref, page.hpp:55-68); every request feeds only its suffix, as a prompt continuing a
    cached prefix.  Same tokens as the engine that prefills every prompt whole (up to low-margin steps), fewer pages in use, the pool drains;
    also together with chunked prefill."""
    from proxy_inference_engine_amd.engine import BatchedEngine
    g, cfg, model = _tiny(golden_dir)
    rng = np.random.default_rng(91)
    V = cfg["vocab_size"]
    system = rng.integers(0, V, 140).tolist()
    prompts = [system + rng.integers(0, V, int(n)).tolist() for n in rng.integers(1, 40, 8)]
    new = 5
    plain = BatchedEngine(model, num_pages=16, max_batch=4)
    want = plain.generate(prompts, new)
    assert plain.shared_pages == 0
    eng = BatchedEngine(model, num_pages=16, max_batch=4, share_prefix=True)
    got = eng.generate(prompts, new)
    assert eng.shared_pages == 2 and eng.pool.get_num_free_pages() == eng.pool.size()
    assert [len(t) for t in got] == [new] * 8
    assert sum(a == b for x, y in zip(got, want) for a, b in zip(x, y)) >= 0.85 * 8 * new
    # a pool too small for four whole prompts (3 pages each) serves four requests at once when they share the prefix: 2 + 4 * 1 pages
    tight = BatchedEngine(model, num_pages=7, max_batch=4, share_prefix=True)
    got_t = tight.generate(prompts, new)
    assert sum(a == b for x, y in zip(got_t, want) for a, b in zip(x, y)) >= 0.85 * 8 * new and tight.pool.get_num_free_pages() == 7
    narrow = BatchedEngine(model, num_pages=7, max_batch=4)             # without sharing only two whole prompts fit at a time: more steps
    narrow.generate(prompts, new)
    assert tight.steps < narrow.steps
    both = BatchedEngine(model, num_pages=16, max_batch=4, share_prefix=True, prefill_chunk=16)
    got_b = both.generate(prompts, new)
    assert sum(a == b for x, y in zip(got_b, want) for a, b in zip(x, y)) >= 0.85 * 8 * new and both.pool.get_num_free_pages() == 16


def test_several_prompts_in_one_pass(golden_dir):
    """pie_decoder_prefill_batch: prompts of 1..150 tokens concatenated into one pass.  Every prompt's last-position logits, its
    cache rows and the decode steps that follow must be what the single-prompt path gives for it alone (same GEMM contract;
    the attention kernel differs: segment-masked rows of this pass instead of the paged cache)."""
    from tests._util import assert_vec_close
    g, cfg, model = _tiny(golden_dir)
    rng = np.random.default_rng(17)
    lens = [1, 5, 33, 64, 150, 7, 97]
    prompts = [rng.integers(0, cfg["vocab_size"], n).astype(np.int32) for n in lens]
    pool = model.enable_paged_kv(num_pages=40)
    ref_logits, ref_caches = [], []
    for p in prompts:
        c = model.make_cache()
        _, _, lg = model.step(torch.from_numpy(p).cuda(), c)
        ref_logits.append(lg.float().cpu().numpy().copy())
        ref_caches.append(c)
    caches = [model.make_cache() for _ in prompts]
    toks, logprobs, logits = model.prefill_batch([p.tolist() for p in prompts], caches)
    assert toks.shape == (len(lens),) and [c[0].offset for c in caches] == lens
    got = logits.float().cpu().numpy()
    for i, n in enumerate(lens):
        # a 1- or 5-token prompt alone runs the GEMV regime (exact affine sums); in the batch it rides the many-row regime
        assert_vec_close(got[i], ref_logits[i], "bfloat16", c_max=6.0 if n < 6 else 4.0, c_rms=5.0 if n < 6 else 4.0, what=f"prompt {i} ({n} tokens)")
        k_ref, v_ref = ref_caches[i][1].state
        k_got, v_got = caches[i][1].state
        assert_vec_close(k_got.float().cpu().numpy().ravel(), k_ref.float().cpu().numpy().ravel(), "bfloat16", what=f"layer-1 keys of prompt {i}")
        assert_vec_close(v_got.float().cpu().numpy().ravel(), v_ref.float().cpu().numpy().ravel(), "bfloat16", what=f"layer-1 values of prompt {i}")
    # both sets of caches continue identically through the multi-sequence step
    feed = torch.tensor([3, 1, 4, 1, 5, 9, 2], dtype=torch.int32)
    _, _, la = model.step_batch(feed, caches, graph=False)
    la = la.float().cpu().numpy().copy()
    _, _, lb = model.step_batch(feed, ref_caches, graph=False)
    lb = lb.float().cpu().numpy()
    for i in range(len(lens)):
        assert_vec_close(la[i], lb[i], "bfloat16", what=f"decode after batched prefill, prompt {i}")
    with pytest.raises(ValueError, match="fresh"):
        model.prefill_batch([[1, 2]], [caches[0]])


def test_mixed_prompt_and_decode_pass_vs_oracle(golden_dir):
    """pie_decoder_step_mixed (batch_details.hpp:10-88: prefill- and decode-state sequences in one BatchDetails): three decoding sequences
    (one crossing a page boundary with this very token, one that just did) and fresh prompts of 9 and 70 tokens share ONE pass over the
    weights; then all five decode while a one-token prompt joins.  Every output row must be what the oracle gives for that sequence alone
    with its Linears in the many-row regime (the pass multiplies all rows at once); caches, offsets and page accounting as if the
    sequences had been served separately."""
    from tests._util import assert_vec_close
    from tests.test_gpu_decode import margin_bound
    g, cfg, model = _tiny(golden_dir)
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    rng = np.random.default_rng(5)
    V = cfg["vocab_size"]
    dlens, plens = [63, 64, 20], [9, 70]
    dprompts = [rng.integers(0, V, n).astype(np.int32) for n in dlens]
    pprompts = [rng.integers(0, V, n).astype(np.int32) for n in plens]
    late = rng.integers(0, V, 1).astype(np.int32)
    pool = model.enable_paged_kv(num_pages=16, max_blocks=2)
    dcaches = []
    for p in dprompts:
        c = model.make_cache()
        model.step(torch.from_numpy(p).cuda(), c)
        dcaches.append(c)
    # oracle: each sequence alone; from the mixed pass on every product is in the many-row regime
    orcs = []
    for p in dprompts + pprompts + [late]:
        orc = po.OracleLlama(cfg, w, "bfloat16")
        orcs.append((orc, [po.OracleKVCache() for _ in orc.layers]))
    want0 = [orcs[i][0].forward(dprompts[i], orcs[i][1])[-1] for i in range(3)]
    feed1 = [int(np.argmax(x)) for x in want0]
    po.set_qmm_min_rows(1)
    try:
        want1 = [orcs[i][0].forward(np.array([feed1[i]]), orcs[i][1])[0] for i in range(3)]
        want1 += [orcs[3 + j][0].forward(pprompts[j], orcs[3 + j][1])[-1] for j in range(2)]
        feed2 = [int(np.argmax(x)) for x in want1]
        want2 = [orcs[i][0].forward(np.array([feed2[i]]), orcs[i][1])[0] for i in range(5)]
        want2.append(orcs[5][0].forward(late, orcs[5][1])[-1])
    finally:
        po.set_qmm_min_rows(6)

    def check(got, nxt, want, what):
        for i, wv in enumerate(want):
            assert_vec_close(got[i], wv, "bfloat16", what=f"{what} row {i}")
            top2 = np.sort(wv)[-2:]
            if top2[1] - top2[0] > margin_bound(wv):
                assert int(nxt[i].item()) == int(np.argmax(wv)), f"{what} row {i}: greedy token"

    pcaches = [model.make_cache() for _ in pprompts]
    nxt, logprobs, logits = model.step_mixed(torch.tensor(feed1, dtype=torch.int32), dcaches, [p.tolist() for p in pprompts], pcaches)
    assert nxt.shape == (5,) and logits.shape == (5, V) and logprobs.shape == (5, V)
    check(logits.float().cpu().numpy(), nxt, want1, "mixed pass 1")
    assert np.all(np.abs(logprobs.double().exp().sum(dim=1).cpu().numpy() - 1.0) < 1e-4)
    assert [c[0].offset for c in dcaches + pcaches] == [64, 65, 21, 9, 70]
    lcache = model.make_cache()
    nxt, logprobs, logits = model.step_mixed(torch.tensor(feed2, dtype=torch.int32), dcaches + pcaches, [late.tolist()], [lcache])
    check(logits.float().cpu().numpy(), nxt, want2, "mixed pass 2")
    assert [c[0].offset for c in dcaches + pcaches + [lcache]] == [65, 66, 22, 10, 71, 1]
    assert pool.size() - pool.get_num_free_pages() == sum((n + 63) // 64 for n in [65, 66, 22, 10, 71, 1])
    # the sequences continue through the plain multi-sequence step, whose rows must agree with the oracle's next step
    feed3 = [int(np.argmax(x)) for x in want2]
    po.set_qmm_min_rows(1)
    try:
        want3 = [orcs[i][0].forward(np.array([feed3[i]]), orcs[i][1])[0] for i in range(6)]
    finally:
        po.set_qmm_min_rows(6)
    nxt, _, logits = model.step_batch(torch.tensor(feed3, dtype=torch.int32), dcaches + pcaches + [lcache], graph=False)
    check(logits.float().cpu().numpy(), nxt, want3, "step after the mixed passes")
    # argument errors
    with pytest.raises(ValueError, match="one token per"):
        model.step_mixed(torch.tensor([1, 2], dtype=torch.int32), [dcaches[0]], [[1, 2]], [model.make_cache()])
    with pytest.raises(ValueError, match="distinct"):
        model.step_mixed(torch.tensor([1, 2], dtype=torch.int32), [dcaches[0], dcaches[0]], [], [])


def test_prompt_chunks_and_shared_prefix_suffixes_in_a_mixed_pass(golden_dir):
    """step_mixed with prompts that CONTINUE a cached prefix (pie_decoder_step_mixed's chunks): a 150-token prompt fed as chunks of 64 + 50 + 36
    rows, each riding the step of two decoding sequences -- the chunk's last-position logits must be the oracle's logits of the WHOLE prompt
    at that position (causal attention over the cached pages + the chunk's own rows, across a page boundary), the cache afterwards the
    single-pass cache.  Then the sequence is forked (two shared prefix pages, KVPage ref counts page.hpp:55-68) and both branches take
    different suffixes in ONE pass; each must match the oracle's continuation.  int8 pools refuse such prompts by name."""
    from proxy_inference_engine_amd.cache.kv_cache.paged import PagedKVCache
    from tests._util import assert_vec_close
    g, cfg, model = _tiny(golden_dir)
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    rng = np.random.default_rng(77)
    V = cfg["vocab_size"]
    prompt = rng.integers(0, V, 150).astype(np.int32)
    orc = po.OracleLlama(cfg, w, "bfloat16")
    ocache = [po.OracleKVCache() for _ in orc.layers]
    want = orc.forward(prompt, ocache)                                    # logits at all 150 positions (many-row regime)
    pool = model.enable_paged_kv(num_pages=24, max_blocks=4)
    dcaches = []
    for n in (20, 63):
        c = model.make_cache()
        model.step(torch.from_numpy(rng.integers(0, V, n).astype(np.int32)).cuda(), c)
        dcaches.append(c)
    feed = torch.tensor([5, 9], dtype=torch.int32)
    c = model.make_cache()
    for lo, hi in ((0, 64), (64, 114), (114, 150)):
        nxt, _, logits = model.step_mixed(feed, dcaches, [prompt[lo:hi].tolist()], [c])
        assert c[0].offset == hi and logits.shape == (3, V)
        assert_vec_close(logits[2].float().cpu().numpy(), want[hi - 1], "bfloat16", what=f"chunk [{lo}, {hi}): logits of its last position")
    ref = model.make_cache()
    model.step(torch.from_numpy(prompt).cuda(), ref)                      # the whole prompt in one pass
    for l in range(len(ref)):
        for got_t, ref_t in zip(c[l].state, ref[l].state):
            assert_vec_close(got_t.float().cpu().numpy().ravel(), ref_t.float().cpu().numpy().ravel(), "bfloat16", what=f"layer {l} cache after the chunks")
    # fork: two full pages shared, the partly filled third copied; both branches continue with their own suffix in one pass
    seq_b = c[0].page_manager.fork()
    b = [PagedKVCache(seq_b, i) for i in range(len(c))]
    assert seq_b.pages[:2] == c[0].page_manager.pages[:2] and all(pool.get_page(p).get_ref_count() == 2 for p in seq_b.pages[:2])
    suf_a, suf_b = rng.integers(0, V, 7).astype(np.int32), rng.integers(0, V, 70).astype(np.int32)
    ocache_b = [po.OracleKVCache() for _ in orc.layers]
    orc.forward(prompt, ocache_b)
    po.set_qmm_min_rows(1)                                                # every row of the pass multiplies in the many-row regime
    try:
        want_a = orc.forward(suf_a, ocache)[-1]
        want_b = orc.forward(suf_b, ocache_b)[-1]
    finally:
        po.set_qmm_min_rows(6)
    _, _, logits = model.step_mixed(None, [], [suf_a.tolist(), suf_b.tolist()], [c, b])
    assert c[0].offset == 157 and b[0].offset == 220
    assert_vec_close(logits[0].float().cpu().numpy(), want_a, "bfloat16", what="7-token suffix behind the shared prefix")
    assert_vec_close(logits[1].float().cpu().numpy(), want_b, "bfloat16", what="70-token suffix behind the shared prefix")
    assert all(pool.get_page(p).get_ref_count() == 2 for p in seq_b.pages[:2])
    with pytest.raises(ValueError, match="fresh"):
        model.prefill_batch([[1, 2]], [c])
    seq_b.release()
    model.enable_paged_kv(num_pages=8, kv_dtype=torch.int8)
    c8 = model.make_cache()
    model.prefill_batch([[1, 2, 3, 4, 5, 6, 7]], [c8])
    with pytest.raises(ValueError, match="T pages"):
        model.step_mixed(None, [], [[8, 9]], [c8])
    model.enable_paged_kv(num_pages=8)   # back to T pages
    model.step(torch.tensor([1, 2, 3], dtype=torch.int32).cuda(), model.make_cache())


def test_mixed_pass_on_int8_pages_matches_the_separate_passes(golden_dir):
    """step_mixed on a pool of int8 pages: the decode rows read the quantised pages (paged_i8 kernel), the prompt rows this pass's own T rows.
    Against the same sequences served by a prompt pass + a multi-sequence step of their own on the same pool format: logits within the
    rounding of two GEMM shapes, the pages' int8 codes of the prompts identical (they do not depend on the other rows), greedy tokens equal."""
    from tests._util import assert_vec_close
    g, cfg, model = _tiny(golden_dir)
    rng = np.random.default_rng(29)
    V = cfg["vocab_size"]
    dprompts = [rng.integers(0, V, n).astype(np.int32).tolist() for n in (40, 64, 7, 90, 12, 33)]
    pprompts = [rng.integers(0, V, n).astype(np.int32).tolist() for n in (17, 65)]
    L, Hkv = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
    ks = torch.full((L, Hkv), 0.05, dtype=torch.float16)
    vs = torch.full((L, Hkv), 0.02, dtype=torch.float16)
    feed = torch.tensor([3, 1, 4, 1, 5, 9], dtype=torch.int32)
    model.enable_paged_kv(num_pages=32, kv_dtype=torch.int8, kv_scales=(ks, vs))
    ref_d = [model.make_cache() for _ in dprompts]
    model.prefill_batch(dprompts, ref_d)
    ref_p = [model.make_cache() for _ in pprompts]
    tp, _, lp = model.prefill_batch(pprompts, ref_p)
    tp, lp = tp.clone(), lp.float().cpu().numpy().copy()
    td, _, ld = model.step_batch(feed, ref_d, graph=False)          # six rows: the many-row regime, like the mixed pass
    td, ld = td.clone(), ld.float().cpu().numpy().copy()
    dc = [model.make_cache() for _ in dprompts]
    model.prefill_batch(dprompts, dc)
    pc = [model.make_cache() for _ in pprompts]
    nxt, _, logits = model.step_mixed(feed, dc, pprompts, pc)
    got = logits.float().cpu().numpy()
    for i in range(6):
        assert_vec_close(got[i], ld[i], "bfloat16", c_max=6.0, c_rms=5.0, what=f"decode row {i}")
    for j in range(2):
        assert_vec_close(got[6 + j], lp[j], "bfloat16", c_max=6.0, c_rms=5.0, what=f"prompt {j}")
        k_ref, v_ref = ref_p[j][0].state
        k_got, v_got = pc[j][0].state
        assert k_got.dtype == torch.int8 and torch.equal(k_got, k_ref) and torch.equal(v_got, v_ref)          # layer 0: independent of attention
    agree = int((nxt.cpu() == torch.cat([td, tp]).cpu()).sum())
    assert agree >= 7, f"{agree} of 8 greedy tokens agree"
    model.enable_paged_kv(num_pages=8)   # back to T pages
    model.step(torch.tensor([1, 2, 3], dtype=torch.int32).cuda(), model.make_cache())


def test_prompt_pass_then_few_sequence_step_with_a_large_vocabulary():
    """the serving order `prefill_batch` of S <= 5 prompts, then `step_batch` of B <= S sequences, with a vocabulary whose
    lm_head GEMV runs more than 256 waves (V = 4096: 2048 waves).  The prompt pass sizes the tail partials at 256 per row; the fused
    few-sequence step writes one per GEMV wave -- the buffer's capacity is tracked in entries now (it was tracked in rows: an 8 x overrun).
    Every row must equal that sequence served alone, and a neighbouring allocation made right after the prompt pass must stay intact."""
    from proxy_inference_engine_amd.models.llama import Model, ModelArgs
    from proxy_inference_engine_amd.models.utils import synthetic_checkpoint
    from tests._util import assert_vec_close
    cfg = {"model_type": "llama", "hidden_size": 256, "num_hidden_layers": 2, "intermediate_size": 704, "num_attention_heads": 4,
           "num_key_value_heads": 2, "rms_norm_eps": 1e-5, "vocab_size": 4096, "rope_theta": 10000.0, "max_position_embeddings": 2048,
           "tie_word_embeddings": False, "quantization": {"group_size": 64, "bits": 4}}
    model = Model(ModelArgs(**cfg), synthetic_checkpoint(cfg, seed=5, lm_head_gain=4.0))
    rng = np.random.default_rng(3)
    prompts = [rng.integers(0, cfg["vocab_size"], n).astype(np.int32) for n in (9, 40, 23)]
    model.enable_paged_kv(num_pages=16)
    alone, alone_next = [], []
    for p in prompts:  # each sequence alone: prompt, then one decode step on its own greedy token
        c = model.make_cache()
        tok, _, _ = model.step(torch.from_numpy(p).cuda(), c)
        alone_next.append(int(tok.item()))
        _, lp, lg = model.step(None, c)
        alone.append((lg.float().cpu().numpy().copy(), lp.cpu().numpy().copy()))
        c[0].page_manager.release()
    caches = [model.make_cache() for _ in prompts]
    toks, _, _ = model.prefill_batch([p.tolist() for p in prompts], caches)   # sizes the partials for a prompt pass: 256 per row
    guard = torch.full((1 << 20,), 0x5A, dtype=torch.uint8, device="cuda")   # neighbours of whatever the pass allocated
    feed = torch.tensor(alone_next, dtype=torch.int32, device="cuda")
    _, logprobs, logits = model.step_batch(feed, caches, graph=False)
    torch.cuda.synchronize()
    assert bool((guard == 0x5A).all())
    got, got_lp = logits.float().cpu().numpy(), logprobs.cpu().numpy()
    for i in range(len(prompts)):
        assert_vec_close(got[i], alone[i][0], "bfloat16", what=f"row {i} of the step after the prompt pass")
        assert abs(float(np.exp(got_lp[i].astype(np.float64)).sum()) - 1.0) < 1e-4
        assert np.abs(got_lp[i] - alone[i][1]).max() < 0.05


# ---------------------------------------------------------------- int8 pages with per-head scales (page.hpp:25-32)
@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("Hq,Hkv,D,lens", [
    (8, 2, 128, [1, 63, 64, 65, 300]),
    (32, 8, 128, [777, 5, 1500]),
    (6, 2, 64, [129, 0, 64, 31]),          # an idle slot (context 0) in the batch
    (16, 2, 128, [200, 70]),               # 8 q-heads per kv-head: the 2-wave form
    (4, 4, 64, [200]),
    (16, 8, 128, [(11 * i) % 70 for i in range(128)]),   # 128 short sequences x 8 kv-heads: one split each, written directly; idle slots among them
])
def test_int8_pages_append_and_attention_vs_oracle(ops, dt, Hq, Hkv, D, lens):
    """The reference page's own storage: int8 K / V blocks + float16 per-head scales.  Appended rows equal the oracle's quantiser bit for
    bit (every page has its own scales, some small enough to clip), attention over the pages follows the oracle's sdpa on the dequantised
    rows, and a fresh pool's scales are ones (page.hpp:31-32)."""
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    rng = np.random.default_rng(Hq * 100 + len(lens) + 7)
    max_blocks = (max(lens) + 63) // 64 + 1
    n_pages = sum((n + 63) // 64 for n in lens) + 3
    alloc = PageAllocator(n_pages, Hkv, D, dtype=torch.int8, device="cuda")
    assert alloc.page_bytes % 256 == 0 and alloc.page_bytes >= 2 * 64 * Hkv * D + 4 * Hkv
    fresh = alloc.get_page(n_pages - 1)
    assert torch.all(fresh.key_cache_scale() == 1) and torch.all(fresh.value_cache_scale() == 1) and fresh.key_cache_scale().shape == (Hkv, 1)
    assert fresh.key_cache().dtype == torch.int8 and fresh.key_cache().shape == (64, Hkv, D)
    # per-page scales: K rows are N(0,1): 1/32 covers +-4 sigma, 1/64 clips the tails; V likewise
    ksc = rng.choice(np.array([1 / 32, 1 / 48, 1 / 64], np.float16), size=(n_pages, Hkv))
    vsc = rng.choice(np.array([1 / 32, 1 / 40, 1 / 64], np.float16), size=(n_pages, Hkv))
    for pg in range(n_pages):
        ops.page_i8_set_scales(alloc.slab[0], n_pages, Hkv, D, torch.from_numpy(ksc[pg]).cuda(), torch.from_numpy(vsc[pg]).cuda(),
                               torch.tensor([pg], dtype=torch.int32, device="cuda"))
    B = len(lens)
    table = np.zeros((B, max_blocks), np.int32)
    need = [(n + 63) // 64 for n in lens]
    for j in range(max(need)):
        for s in rng.permutation(B):
            if j < need[s]:
                table[s, j] = alloc.allocate_page()
    ks = [po.round_T(rng.standard_normal((n, Hkv, D)), dt) for n in lens]
    vs = [po.round_T(rng.standard_normal((n, Hkv, D)), dt) for n in lens]
    bt = torch.from_numpy(table).cuda()
    for t in range(max(lens)):
        kb, vb, pos = np.zeros((B, Hkv, D), np.float32), np.zeros((B, Hkv, D), np.float32), np.full(B, -1, np.int32)
        for s in range(B):
            if t < lens[s]:
                kb[s], vb[s], pos[s] = ks[s][t], vs[s][t], t
        ops.paged_kv_append_i8(to_dev(po.to_bits(kb, dt), dt), to_dev(po.to_bits(vb, dt), dt), alloc.slab[0], alloc.size(), bt, torch.from_numpy(pos).cuda())
    q = po.round_T(rng.standard_normal((B, Hq, D)), dt)
    got = to_bits(ops.paged_attention_decode_i8(to_dev(po.to_bits(q, dt), dt), alloc.slab[0], n_pages, bt,
                                                torch.tensor(lens, dtype=torch.int32, device="cuda"), Hkv, D ** -0.5))
    clipped = 0
    for s, n in enumerate(lens):
        if n == 0:
            assert not got[s].any(), "an idle slot must produce zeros"
            continue
        pages = table[s, np.arange(n) // 64]                       # page of every token
        kq = po.kv_i8_quantize(ks[s], ksc[pages])                   # [n, Hkv, D] with scales [n, Hkv]
        vq = po.kv_i8_quantize(vs[s], vsc[pages])
        clipped += int((np.abs(kq) == 127).sum())
        for j in range(need[s]):                                    # the stored bytes, in the reference's logical [64, heads, head_dim] view
            page, rows = alloc.get_page(int(table[s, j])), min(64, n - 64 * j)
            assert np.array_equal(page.key_cache()[:rows].cpu().numpy(), kq[64 * j: 64 * j + rows]), f"K bytes of seq {s} block {j}"
            assert np.array_equal(page.value_cache()[:rows].cpu().numpy(), vq[64 * j: 64 * j + rows]), f"V bytes of seq {s} block {j}"
            assert np.array_equal(page.key_cache_scale().cpu().numpy()[:, 0], ksc[table[s, j]])
        k = np.ascontiguousarray(po.kv_i8_dequantize(kq, ksc[pages]).transpose(1, 0, 2))    # [Hkv, n, D] fp32
        v = np.ascontiguousarray(po.kv_i8_dequantize(vq, vsc[pages]).transpose(1, 0, 2))
        want = po.sdpa(q[s][:, None, :], k, v, D ** -0.5, None, dt, True, T=n)
        assert_dot_close(po.from_bits(got[s], dt), po.round_T(want, dt), dt, max_frac=0.03, what=f"int8 paged seq {s} len {n} {Hq}/{Hkv} D{D} {dt}")
    assert clipped > 0 or max(lens) < 100, "the small scales were meant to exercise the clamp"


def test_int8_pages_default_scale_is_one_like_the_reference(ops):
    """KVPage's constructor sets both scales to ones (page.hpp:31-32): with them the stored byte is rint(x) clamped."""
    from proxy_inference_engine_amd.cache.kv_cache.paged import PageAllocator
    dt, Hkv, D = "bfloat16", 2, 64
    alloc = PageAllocator(2, Hkv, D, dtype=torch.int8, device="cuda")
    pid = alloc.allocate_page()
    x = po.round_T(np.array([[-300.0, -2.5, -1.5, -0.5, 0.49, 0.5, 1.5, 2.5] * (D // 8)] * Hkv, np.float32) * np.ones((1, 1, 1), np.float32), dt)
    bt = torch.tensor([[pid]], dtype=torch.int32, device="cuda")
    ops.paged_kv_append_i8(to_dev(po.to_bits(x, dt), dt), to_dev(po.to_bits(x, dt), dt), alloc.slab[0], alloc.size(), bt, torch.tensor([3], dtype=torch.int32, device="cuda"))
    row = alloc.get_page(pid).key_cache()[3].cpu().numpy()
    assert np.array_equal(row[0, :8], np.array([-127, -2, -2, 0, 0, 0, 2, 2], np.int8))          # round-half-even, clamp at -127
    assert np.array_equal(row, po.kv_i8_quantize(x[0], np.ones(Hkv, np.float16)))


def test_decoder_batch_paths_on_int8_pages(golden_dir):
    """PIE_OPT_KV_I8: prefill_batch / step_batch on a pool of int8 pages.  The prompt pass attends to its own T rows, so its logits are
    bit-identical to the T-page run and EVERY layer's pages must hold exactly the oracle's quantisation of that run's K / V rows; the decode
    step then reads the quantised rows back: layer 0's appended row is again exact (it does not depend on attention), the logits stay within
    the quantisation noise of the T-page step, and the greedy tokens agree."""
    from tests._util import assert_vec_close
    g, cfg, model = _tiny(golden_dir)
    rng = np.random.default_rng(23)
    lens = [5, 33, 64, 97]
    prompts = [rng.integers(0, cfg["vocab_size"], n).astype(np.int32).tolist() for n in lens]
    L, Hkv = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
    model.enable_paged_kv(num_pages=24)
    ct = [model.make_cache() for _ in prompts]
    _, _, lg_t = model.prefill_batch(prompts, ct)
    lg_t = lg_t.clone()
    rows = [[tuple(t.float().cpu().numpy()[0].transpose(1, 0, 2) for t in ct[i][l].state) for l in range(L)] for i in range(len(lens))]  # [n, Hkv, D]
    amax = np.zeros((2, L, Hkv), np.float32)
    for i in range(len(lens)):
        for l in range(L):
            for w in range(2):
                amax[w, l] = np.maximum(amax[w, l], np.abs(rows[i][l][w]).max(axis=(0, 2)))
    ks, vs = (torch.from_numpy((amax[w] * 1.25 / 127).astype(np.float16)) for w in range(2))   # headroom for the decode rows
    feed = torch.tensor([3, 1, 4, 1], dtype=torch.int32)
    tok_t, _, la_t = model.step_batch(feed, ct, graph=False)
    tok_t, la_t = tok_t.clone(), la_t.float().cpu().numpy().copy()
    new_t = [tuple(t.float().cpu().numpy()[0].transpose(1, 0, 2)[-1] for t in ct[i][0].state) for i in range(len(lens))]          # layer 0, the step's row

    pool = model.enable_paged_kv(num_pages=24, kv_dtype=torch.int8, kv_scales=(ks, vs))
    c8 = [model.make_cache() for _ in prompts]
    _, _, lg_8 = model.prefill_batch(prompts, c8)
    assert torch.equal(lg_8, lg_t), "the prompt pass reads its own T rows: identical logits"
    ksn, vsn = ks.numpy(), vs.numpy()
    for i, n in enumerate(lens):
        for l in range(L):
            k8, v8 = (t.cpu().numpy()[0].transpose(1, 0, 2) for t in c8[i][l].state)                                             # int8 [n, Hkv, D]
            assert k8.dtype == np.int8
            assert np.array_equal(k8, po.kv_i8_quantize(rows[i][l][0], np.broadcast_to(ksn[l], (n, Hkv)))), f"K codes, prompt {i} layer {l}"
            assert np.array_equal(v8, po.kv_i8_quantize(rows[i][l][1], np.broadcast_to(vsn[l], (n, Hkv)))), f"V codes, prompt {i} layer {l}"
    page = pool.get_page(c8[0][0].page_manager.pages[0])
    assert np.array_equal(page.key_cache_scale(1).cpu().numpy()[:, 0], ksn[1])
    tok_8, _, la_8 = model.step_batch(feed, c8, graph=False)
    for i in range(len(lens)):
        k8, v8 = (t.cpu().numpy()[0].transpose(1, 0, 2)[-1] for t in c8[i][0].state)
        assert np.array_equal(k8, po.kv_i8_quantize(new_t[i][0], ksn[0])) and np.array_equal(v8, po.kv_i8_quantize(new_t[i][1], vsn[0]))
        assert_vec_close(la_8.float().cpu().numpy()[i], la_t[i], "bfloat16", c_max=24.0, c_rms=16.0, what=f"int8-page decode step, sequence {i}")
    assert torch.equal(tok_8.cpu(), tok_t.cpu())
    model.enable_paged_kv(num_pages=8)   # back to T pages
    model.step(torch.tensor([1, 2, 3], dtype=torch.int32).cuda(), model.make_cache())


def test_single_sequence_step_on_int8_pages(golden_dir):
    """The single-sequence decoder step (Model.step / InferenceEngine: one replayed hipGraph) on a pool of int8 pages (page.hpp:25-32): the q|k|v
    launch's RoPE + append epilogue writes the new T row into a staging page, k_paged_kv_append_i8 quantises it into the sequence's page, the
    int8-page attention reads the codes back.  Against the same sequence on T pages: layer 0's stored codes are exactly the oracle's quantisation
    of the T-page run's rows (they do not depend on attention), later layers' within one code, logits within the quantisation noise, greedy
    tokens equal; prompt (iterated steps) + graph-replayed steps across a page boundary; then through InferenceEngine."""
    from proxy_inference_engine_amd import InferenceEngine
    from tests._util import assert_vec_close
    g, cfg, model = _tiny(golden_dir)
    L, Hkv = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
    prompt = np.random.default_rng(29).integers(0, cfg["vocab_size"], 58).astype(np.int32)
    steps = 12                                                              # 58 + 12 positions: crosses the 64-row page boundary
    model.enable_paged_kv(num_pages=8)
    ct = model.make_cache()
    tok, _, lg = model.step(torch.from_numpy(prompt[:5]).cuda(), ct)        # 5 rows: iterated decode steps on both page formats
    for t in prompt[5:]:
        tok, _, lg = model.step(torch.tensor([t], dtype=torch.int32).cuda(), ct)
    ref = [(int(tok.item()), lg.float().cpu().numpy().copy())]
    for _ in range(steps):
        tok, _, lg = model.step(None, ct)
        ref.append((int(tok.item()), lg.float().cpu().numpy().copy()))
    rows = [tuple(t.float().cpu().numpy()[0].transpose(1, 0, 2) for t in ct[l].state) for l in range(L)]   # [n, Hkv, D]
    amax = np.stack([np.stack([np.abs(rows[l][w]).max(axis=(0, 2)) for l in range(L)]) for w in range(2)])
    ks, vs = (torch.from_numpy((amax[w] * 1.25 / 127).astype(np.float16)) for w in range(2))
    ct[0].page_manager.release()

    model.enable_paged_kv(num_pages=8, kv_dtype=torch.int8, kv_scales=(ks, vs))
    c8 = model.make_cache()
    tok, _, lg = model.step(torch.from_numpy(prompt[:5]).cuda(), c8)
    for t in prompt[5:]:
        tok, _, lg = model.step(torch.tensor([t], dtype=torch.int32).cuda(), c8)
    got = [(int(tok.item()), lg.float().cpu().numpy().copy())]
    for i in range(steps):
        tok, _, lg = model.step(None, c8, graph=i >= 2)                      # eager, then the captured graph
        got.append((int(tok.item()), lg.float().cpu().numpy().copy()))
    n = len(prompt) + steps
    assert c8[0].offset == n
    ksn, vsn = ks.numpy(), vs.numpy()
    k8, v8 = (t.cpu().numpy()[0].transpose(1, 0, 2) for t in c8[0].state)    # layer 0: int8 [n, Hkv, D]
    assert k8.dtype == np.int8 and k8.shape[0] == n
    same_tokens = [a[0] == b[0] for a, b in zip(got, ref)]
    upto = same_tokens.index(False) if False in same_tokens else len(same_tokens)     # rows are comparable while both runs fed the same tokens
    m = min(len(prompt) + upto, n)
    assert np.array_equal(k8[:m], po.kv_i8_quantize(rows[0][0][:m], np.broadcast_to(ksn[0], (m, Hkv)))), "layer-0 K codes"
    assert np.array_equal(v8[:m], po.kv_i8_quantize(rows[0][1][:m], np.broadcast_to(vsn[0], (m, Hkv)))), "layer-0 V codes"
    k81 = c8[1].state[0].cpu().numpy()[0].transpose(1, 0, 2)[:m].astype(np.int32)
    assert np.abs(k81 - po.kv_i8_quantize(rows[1][0][:m], np.broadcast_to(ksn[1], (m, Hkv))).astype(np.int32)).max() <= 2, "layer-1 K codes (behind one int8 attention)"
    for i in range(upto):
        assert_vec_close(got[i][1], ref[i][1], "bfloat16", c_max=24.0, c_rms=16.0, what=f"int8-page step {i}")
    assert upto >= 4, f"greedy tokens diverged after {upto} steps: {same_tokens}"
    c8[0].page_manager.release()
    # the engine on such a pool: prompt, then greedy steps -- the same tokens as the model-level run above
    eng = InferenceEngine(model=model)
    eng.prepare_engine(prompt, temp=0)
    gen = eng.generate_step(torch.from_numpy(prompt.astype(np.int64)))
    etoks = [int(next(gen)[0].item()) for _ in range(upto)]
    assert etoks == [t for t, _ in got[:upto]]
    eng.prompt_cache.cache[0].page_manager.release()
    # a FRESH multi-token prompt on int8 pages goes through the several-prompts pass as a batch of one (not L decode steps): its pages hold the
    # codes of the batched pass's rows (MLX's qmm regime: within one code of the row-by-row run's), the first token and the following
    # graph-replayed steps continue from them
    cf = model.make_cache()
    tok, _, lg = model.step(torch.from_numpy(prompt).cuda(), cf)
    assert cf[0].offset == len(prompt)
    kf = cf[0].state[0].cpu().numpy()[0].transpose(1, 0, 2).astype(np.int32)
    assert np.abs(kf - k8[:len(prompt)].astype(np.int32)).max() <= 1, "layer-0 K codes of the batched prompt pass"
    assert_vec_close(lg.float().cpu().numpy(), got[0][1], "bfloat16", c_max=24.0, c_rms=16.0, what="int8-page batched prompt")
    ftoks = [int(tok.item())]
    for i in range(upto - 1):
        tok, _, lg = model.step(None, cf)
        ftoks.append(int(tok.item()))
    assert cf[0].offset == len(prompt) + upto - 1
    assert ftoks[:2] == [t for t, _ in got[:2]], (ftoks, [t for t, _ in got[:upto]])
    cf[0].page_manager.release()
    # a prompt of caller-made embeddings (the VLM text tower's entry, intern/ensemble.py:106-108) on int8 pages: its rows run as decode steps,
    # each copied into the residual stream -- the same rows as the token prompt's, so logits at every position and the pages' codes are equal
    ca, cb = model.make_cache(), model.make_cache()
    ids = torch.from_numpy(prompt[:20]).cuda()
    la = model(ids[None], cache=ca).clone()
    lb = model(None, cache=cb, inputs_embeds=model.embed(ids)[None])
    assert ca[0].offset == cb[0].offset == 20 and torch.equal(la, lb)
    for l in range(L):
        assert all(torch.equal(x, y) for x, y in zip(ca[l].state, cb[l].state))
    nb = model.step(torch.tensor([7], dtype=torch.int32).cuda(), cb)[2].clone()
    assert torch.equal(model.step(torch.tensor([7], dtype=torch.int32).cuda(), ca)[2], nb)
    model.enable_paged_kv(num_pages=8)   # back to T pages


def test_continuous_batching_engine_on_int8_pages(golden_dir):
    """BatchedEngine(kv_dtype=torch.int8): more requests than slots, prompts of mixed length (a lone admission goes through prefill_batch
    as well: int8 pages are written by the batch paths only).  With scales from the T-page run's amax the generated ids equal the T-page
    engine's for most requests; every request completes and the pool drains."""
    from proxy_inference_engine_amd.engine.batch_engine import BatchedEngine
    g, cfg, model = _tiny(golden_dir)
    rng = np.random.default_rng(31)
    prompts = [rng.integers(0, cfg["vocab_size"], n).tolist() for n in (9, 40, 3, 70, 21)]
    ref = BatchedEngine(model, num_pages=16, max_batch=2).generate(prompts, 6)
    L, Hkv = cfg["num_hidden_layers"], cfg["num_key_value_heads"]
    # calibration: amax of K / V over one prompt on T pages
    model.enable_paged_kv(num_pages=8)
    c = model.make_cache()
    model.step(torch.tensor(prompts[3], dtype=torch.int32).cuda(), c)
    ks = torch.stack([c[l].state[0].float().abs().amax(dim=(0, 2, 3)) for l in range(L)]) * (2.0 / 127)
    vs = torch.stack([c[l].state[1].float().abs().amax(dim=(0, 2, 3)) for l in range(L)]) * (2.0 / 127)
    eng = BatchedEngine(model, num_pages=16, max_batch=2, kv_dtype=torch.int8, kv_scales=(ks.half(), vs.half()))
    got = eng.generate(prompts, 6)
    assert [len(t) for t in got] == [6] * len(prompts) and eng.pool.get_num_free_pages() == eng.pool.size()
    same = sum(a == b for a, b in zip(got, ref))
    assert same >= len(prompts) - 1, f"int8 pages changed the greedy ids of {len(prompts) - same} of {len(prompts)} requests"
    assert all(a[0] == b[0] for a, b in zip(got, ref)), "the first token comes from the prompt pass, which reads its own T rows"
