"""-m gpu: the parity gaps VERDICT r1 listed -- cache growth at unaligned offsets on the device, the engine's generate() loop and
its logits-processor branch, the TP=8 shard shapes of the 70B layer, prefix reuse after a diverging request, a regrown paged
sequence in the batched step, and the persistent one-launch step against the launch sequence (bit for bit).
All through the C ABI, against the CPU oracle (tolerances: tests/test_gpu_decode.py)."""
import json

import numpy as np
import pytest
import torch

from oracle import pie_oracle as po
from tests._util import EPS, assert_bits_close, assert_vec_close, codes_dev, to_bits, to_dev
from tests.test_gpu_decode import build, device_weights, margin_bound

pytestmark = pytest.mark.gpu
DT = "bfloat16"


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = np.load(golden_dir / "tiny_llama_w4_bf16.npz")
    cfg = json.loads(str(g["config_json"]))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    return g, cfg, w, build(cfg, w)


@pytest.mark.parametrize("chunks", [(520, 300), (700, 400)])
def test_cache_growth_at_unaligned_offset_matches_reference_capacity(tiny, chunks):
    """reusable.py:125-129: a multi-token turn appended at an offset that is not a multiple of 256 grows the buffers from the
    OFFSET (520 then 300 -> 1024, not 1280).  The capacity also picks the attention plan, so the logits are checked too."""
    g, cfg, w, model = tiny
    rng = np.random.default_rng(sum(chunks))
    orc = po.OracleLlama(cfg, w, DT)
    ocache = [po.OracleKVCache() for _ in orc.layers]
    cache = model.make_cache()
    want = None
    for n in chunks:
        ids = rng.integers(0, cfg["vocab_size"], n)
        want = orc.forward(ids, ocache)[-1]
        tok, lp, logits = model.step(torch.from_numpy(ids).to(torch.int32).cuda(), cache)
        assert (cache[0].offset, cache[0].capacity) == (ocache[0].offset, ocache[0].keys.shape[2])
    if chunks == (520, 300):
        assert cache[0].capacity == 1024
    assert_vec_close(logits.float().cpu().numpy(), want, DT, what=f"logits after {chunks}")
    t = rng.integers(0, cfg["vocab_size"], 1)
    want = orc.forward(t, ocache)[0]
    _, _, logits = model.step(torch.from_numpy(t).to(torch.int32).cuda(), cache)
    assert_vec_close(logits.float().cpu().numpy(), want, DT, what="decode step after the growth")


def _oracle_generate(cfg, w, prompt, n, penalty=1.0, context_size=60):
    """generate_step of the oracle with the repetition-penalty processor of logits_processors/__init__.py applied to the last
    logits (engine/inference_engine.py:252-271); the history is the prompt cache's computed_ids (prompt + fed tokens)."""
    orc = po.OracleLlama(cfg, w, DT)
    cache = [po.OracleKVCache() for _ in orc.layers]
    hist = [int(t) for t in prompt]
    ids = np.asarray(prompt)
    out = []
    for _ in range(n):
        last = np.array(orc.forward(ids, cache)[-1], np.float32)
        if penalty != 1.0:
            idx = np.array(hist[-context_size:])
            sel = last[idx]
            last[idx] = np.where(sel < 0, sel * penalty, sel / penalty).astype(np.float32)
            last = po.from_bits(po.to_bits(last, DT), DT)  # the processor works on T logits
        tok, lp = po.logprobs_argmax(last)
        out.append((int(tok), lp, last))
        ids = np.array([tok])
        hist.append(int(tok))
    return out


def test_engine_generate_stop_length_and_logprobs(tiny):
    """InferenceEngine.generate (engine/inference_engine.py:175-226): the token stream equals generate_step's (itself checked
    against the oracle), a stop token ends it with "stop" WITHOUT being yielded, max_completion_tokens ends it with "length",
    the logprobs / top_logprobs maps hold the step's log-probabilities."""
    from proxy_inference_engine_amd import InferenceEngine
    g, cfg, w, model = tiny
    prompt = torch.from_numpy(g["prompt"])
    n = 8
    eng = InferenceEngine(model=model)
    eng.prepare_engine(prompt, temp=0)
    gen = eng.generate_step(prompt)
    ref = []
    for _ in range(n):
        tok, lp = next(gen)
        ref.append((int(tok.item()), lp.float().cpu().numpy().copy()))
    oref = _oracle_generate(cfg, w, g["prompt"], 1)
    assert ref[0][0] == oref[0][0] == int(g["tokens"][0])             # anchored on the oracle / the golden fixture
    assert_vec_close(ref[0][1], oref[0][1], DT, what="first logprobs")

    def run(eng, **kw):
        gen = eng.generate(prompt, **kw)
        toks, maps = [], []
        while True:
            try:
                t, m = next(gen)
            except StopIteration as stop:
                return toks, maps, stop.value
            toks.append(t), maps.append(dict(m))

    eng = InferenceEngine(model=model)
    eng.prepare_engine(prompt, temp=0)
    toks, maps, reason = run(eng, max_completion_tokens=6, logprobs=True, top_logprobs=3)
    assert reason == "length" and toks == [r[0] for r in ref[:6]]
    for i, m in enumerate(maps):
        assert len(m) == 3 and toks[i] in m                            # the greedy token is the top-1 entry
        top3 = np.sort(ref[i][1])[-3:][::-1]
        assert np.array_equal(np.array(sorted(m.values(), reverse=True), np.float32), top3.astype(np.float32))
    # a stop token: the first token value that appears later in the stream ends it before being yielded
    stop_at = 3
    stop_tok = ref[stop_at][0]
    first = next(i for i, r in enumerate(ref) if r[0] == stop_tok)
    eng = InferenceEngine(model=model, stop_tokens=[stop_tok])
    eng.prepare_engine(prompt, temp=0)
    toks, _, reason = run(eng, max_completion_tokens=50)
    assert reason == "stop" and toks == [r[0] for r in ref[:first]]
    # logprobs=True with top_logprobs = 0: only the chosen token's entry
    eng = InferenceEngine(model=model)
    eng.prepare_engine(prompt, temp=0)
    toks, maps, reason = run(eng, max_completion_tokens=2, logprobs=True)
    assert reason == "length" and toks == [r[0] for r in ref[:2]]
    assert all(list(m.keys()) == [t] and np.float32(m[t]) == ref[i][1][t] for i, (t, m) in enumerate(zip(toks, maps)))


def test_engine_repetition_penalty_branch(tiny):
    """repetition_penalty != 1 takes _inference's logits-processor branch (engine/inference_engine.py:257-266, 319-335): full
    Model.__call__, the processor on logits[:, -1, :] with the prompt cache's history, then the HIP log-softmax tail."""
    from proxy_inference_engine_amd import InferenceEngine
    g, cfg, w, model = tiny
    prompt = torch.from_numpy(g["prompt"])
    ref = _oracle_generate(cfg, w, g["prompt"], 6, penalty=1.8, context_size=20)
    plain = _oracle_generate(cfg, w, g["prompt"], 6)
    eng = InferenceEngine(model=model)
    eng.prepare_engine(prompt, temp=0, repetition_penalty=1.8, context_size=20)
    gen = eng.generate_step(prompt)
    checked = 0
    for i, (otok, olp, olast) in enumerate(ref):
        tok, lp = next(gen)
        assert_vec_close(lp.float().cpu().numpy(), olp, DT, what=f"penalised logprobs step {i}")
        top2 = np.sort(olast)[-2:]
        if top2[1] - top2[0] > margin_bound(olast):
            assert int(tok.item()) == otok
            checked += 1
        else:
            break
    assert checked >= 1
    assert any(not np.allclose(a[1], b[1]) for a, b in zip(ref, plain))  # the penalty changed something


def test_prefix_reuse_after_a_diverging_request(tiny):
    """A -> B -> A: after B diverged at k the caches hold B's rows beyond k.  The third request must re-process A[k:] and give the
    tokens of a cold run (the reference, which never cuts computed_ids, would attend over B's rows: prompt_cache.py:52-76)."""
    from proxy_inference_engine_amd import InferenceEngine
    g, cfg, w, model = tiny
    rng = np.random.default_rng(9)
    A = np.concatenate([g["prompt"], rng.integers(0, cfg["vocab_size"], 10)])
    B = np.concatenate([g["prompt"][:12], rng.integers(0, cfg["vocab_size"], 17)])

    def first_tokens(eng, ids, n=4):
        gen = eng.generate_step(torch.from_numpy(ids))
        return [int(next(gen)[0].item()) for _ in range(n)], [next(gen)[1].float().cpu().numpy() for _ in range(1)]

    cold = InferenceEngine(model=model)
    cold.prepare_engine(A, temp=0)
    want, want_lp = first_tokens(cold, A)
    eng = InferenceEngine(model=model)
    eng.prepare_engine(A, temp=0)
    first_tokens(eng, A)
    first_tokens(eng, B)
    assert eng.prompt_cache.computed_ids[:12] == [int(t) for t in g["prompt"][:12]]
    got, got_lp = first_tokens(eng, A)
    assert got == want
    # the warm run re-processes A[12:] on top of a 12-row prefix (another GEMM row count than the cold 34-row prompt): close, not identical
    assert_vec_close(got_lp[0], want_lp[0], DT, what="logprobs after A -> B -> A")


# ---------------------------------------------------------------- TP = 8 shard shapes of the Llama-3-70B layer, on ONE card
# (BASELINE.json configs[4]; partitioning of proxy_inference_engine_amd/tp.py: column-parallel q|k|v and gate|up, row-parallel
# o_proj and down split on multiples of the 64-wide group, vocab-parallel lm_head).  A rank's slices, as random int4 matrices:
TP8 = {"qkv": (1280, 8192, False), "o_proj": (8192, 1024, True), "gate_up": (7168, 8192, False), "down": (8192, 3584, True),
       "lm_head": (16032, 8192, False)}


@pytest.mark.parametrize("name", list(TP8))
def test_llama70b_tp8_shard_shapes(name):
    """o_proj K = 1024 is half a 2048-wide W4S slice (ragged lanes), down K = 3584 is 1.75 slices, lm_head 16,032 rows is no
    multiple of the wave geometry.  Column-parallel products are rounded (pie_qgemv_w4g64 vs the oracle's qmv regime); the
    row-parallel ones are the UN-rounded fp32 partials that the all-reduce sums (pie_qgemv_w4g64_f32) vs a float64 affine sum."""
    from proxy_inference_engine_amd import hip_ops as ops
    N, K, partial = TP8[name]
    rng = np.random.default_rng(N + K)
    wf = po.round_T(rng.standard_normal((N, K)) * 0.03, DT)
    wq, s, b = po.quantize(wf, 64, 4, DT)
    x = po.round_T(rng.standard_normal((1, K)), DT)
    wd = ops.repack_w4s(codes_dev(wq), to_dev(s, DT), to_dev(b, DT))
    xd = to_dev(po.to_bits(x, DT), DT)
    if not partial:
        got = ops.quantized_matmul(xd, wd)
        want = po.quantized_matmul(x, wq, s, b, dtype=DT)
        assert_bits_close(to_bits(got), po.to_bits(want, DT), what=f"TP8 {name} [{N}x{K}]")
        return
    got = ops.quantized_matmul_partial(xd, wd).cpu().numpy().astype(np.float64).reshape(-1)
    q = np.zeros((N, K), np.float64)
    for i in range(8):
        q[:, i::8] = (wq >> np.uint32(4 * i)) & np.uint32(0xF)
    sf, bf = po.from_bits(s, DT).astype(np.float64), po.from_bits(b, DT).astype(np.float64)
    wfull = q * np.repeat(sf, 64, axis=1) + np.repeat(bf, 64, axis=1)
    xv = x.astype(np.float64).reshape(-1)
    want = wfull @ xv
    bound = 2.0 ** -20 * (np.abs(wfull) @ np.abs(xv))                  # fp32 accumulation of K terms, generous
    assert np.all(np.abs(got - want) <= bound + 1e-30), f"TP8 {name}: worst {np.abs(got - want).max()} vs bound {bound.min()}"


def test_batched_step_after_trim_and_regrow(tiny):
    """A PagedSequence that is truncated and regrown holds the same NUMBER of pages in a different order (LIFO free list).  The
    batched step caches its device block table; it must notice (the key was (id, page count))."""
    g, cfg, w, _ = tiny
    model = build(cfg, w)
    model.enable_paged_kv(num_pages=32)
    rng = np.random.default_rng(4)
    orc = po.OracleLlama(cfg, w, DT)
    prompts = [rng.integers(0, cfg["vocab_size"], n) for n in (150, 70)]
    caches, ocaches = [], []
    for p in prompts:
        c = model.make_cache()
        model.step(torch.from_numpy(p).to(torch.int32).cuda(), c)
        oc = [po.OracleKVCache() for _ in orc.layers]
        orc.forward(p, oc)
        caches.append(c), ocaches.append(oc)
    toks = rng.integers(0, cfg["vocab_size"], 2)
    model.step_batch(torch.from_numpy(toks).to(torch.int32), caches)   # fills the table cache
    for oc, t in zip(ocaches, toks):
        orc.forward(np.array([t]), oc)
    # sequence 0: trim to 60 tokens (frees its 2nd and 3rd page), regrow by a 90-token turn -> same page count, other ids
    before = list(caches[0][0].page_manager.pages)
    for layer_cache in caches[0]:
        layer_cache.trim(caches[0][0].offset - 60)
    for oc in ocaches[0]:
        oc.trim(oc.offset - 60)
    turn = rng.integers(0, cfg["vocab_size"], 90)
    model.step(torch.from_numpy(turn).to(torch.int32).cuda(), caches[0])
    orc.forward(turn, ocaches[0])
    after = list(caches[0][0].page_manager.pages)
    assert len(after) == len(before) and after != before, (before, after)
    toks = rng.integers(0, cfg["vocab_size"], 2)
    nxt, lp, logits = model.step_batch(torch.from_numpy(toks).to(torch.int32), caches)
    for i, (oc, t) in enumerate(zip(ocaches, toks)):
        want = orc.forward(np.array([t]), oc)[0]
        assert_vec_close(logits[i].float().cpu().numpy(), want, DT, what=f"sequence {i} after trim + regrow")


def test_generate_step_with_a_mask_array(tiny):
    """generate_step(prompt, mask=array) (inference_engine.py:246-249: the same array goes to the prompt pass and every step): a per-head
    constant is the only kind that broadcasts against all of them; tokens and logprobs equal the unmasked stream.  A [L, L] mask is refused."""
    from proxy_inference_engine_amd import InferenceEngine
    g, cfg, w, model = tiny
    prompt = torch.from_numpy(g["prompt"])

    def stream(mask):
        eng = InferenceEngine(model=model)
        eng.prepare_engine(prompt, temp=0)
        gen = eng.generate_step(prompt, mask=mask)
        out = []
        for _ in range(5):
            tok, lp = next(gen)
            out.append((int(tok.item()), lp.clone()))
        return out

    ref = stream(None)
    for mk in (torch.zeros(1, 1, 1, 1, device="cuda"), torch.ones(1, cfg["num_attention_heads"], 1, 1, dtype=torch.bool), "causal"):
        got = stream(mk)
        assert [t for t, _ in got] == [t for t, _ in ref] and all(torch.equal(a[1], b[1]) for a, b in zip(got, ref))
    with pytest.raises(ValueError, match="broadcast"):
        next(InferenceEngine(model=model).generate_step(prompt, mask=torch.zeros(len(prompt), len(prompt))))


def test_model_call_accepts_the_causal_mask_it_would_build_itself(tiny):
    """Model.__call__(inputs, mask=...) (language.py:199-204): the reference builds create_attention_mask(h, cache) when mask is None
    and passes a caller's mask through to sdpa; here the causal mask is implicit in the kernels, so that very mask (additive array,
    boolean array or "causal") is accepted and gives the same logits, at offset 0 and behind a cached prefix; any other mask is refused."""
    from proxy_inference_engine_amd.models.base import create_causal_mask
    g, cfg, w, model = tiny
    ids = torch.from_numpy(g["prompt"][:12])[None].cuda()
    ref = model(ids, cache=model.make_cache())
    for mk in ("causal", create_causal_mask(12, 0, device="cuda"), ~(create_causal_mask(12, 0, device="cuda") < 0)):
        assert torch.equal(model(ids, mask=mk, cache=model.make_cache()), ref)
    cache, cache2 = model.make_cache(), model.make_cache()
    model(ids[:, :7], cache=cache)
    model(ids[:, :7], cache=cache2)
    tail = model(ids[:, 7:], mask=create_causal_mask(5, 7, device="cuda"), cache=cache).clone()
    assert torch.equal(tail, model(ids[:, 7:], cache=cache2))
    with pytest.raises(NotImplementedError):
        model(ids, mask=torch.zeros(12, 12, device="cuda"), cache=model.make_cache())      # blocks nothing: bidirectional attention
    with pytest.raises(NotImplementedError):
        model(ids, mask=create_causal_mask(12, 3, device="cuda"), cache=model.make_cache())  # wrong offset
