"""-m gpu: the fused HIP sampler (csrc/sampler.hip, pie_sample) against the reference's sort-based definitions.

samplers/{top_p,min_p,top_k,categorical}.py of the reference filter with argsort / cumsum / argpartition and draw with
mx.random.categorical; the product restates them as one sort-free kernel.  Pinned here: the KEPT SET of every branch equals the numpy
restatement of the reference's definition (tests/test_host_logic.py: _ref_sets) id for id; every draw lies in it; the empirical
distribution follows the renormalised probabilities; seeding restarts the stream and the device-side call counter advances it."""
import numpy as np
import pytest
import torch

from tests.test_host_logic import _ref_sets

pytestmark = pytest.mark.gpu

CASES = [dict(top_p=0.6), dict(top_p=0.95), dict(top_p=0.05), dict(min_p=0.1), dict(min_p=0.3, min_tokens_to_keep=4), dict(min_p=0.9, min_tokens_to_keep=7),
         dict(top_k=1), dict(top_k=5), dict(top_k=50), dict()]


def _logprobs(rng, V, scale=2.0):
    logits = (rng.standard_normal(V) * scale).astype(np.float32)
    return logits - np.float32(np.log(np.exp(logits.astype(np.float64)).sum()))


def _mode(kw):
    if "top_p" in kw:
        return "top_p", kw["top_p"], 0
    if "min_p" in kw:
        return "min_p", kw["min_p"], kw.get("min_tokens_to_keep", 1)
    if "top_k" in kw:
        return "top_k", 0.0, kw["top_k"]
    return "categorical", 0.0, 0


@pytest.mark.parametrize("kw", CASES)
@pytest.mark.parametrize("V,temp", [(64, 0.8), (1000, 1.0), (4099, 0.7), (128256, 1.0)])
def test_kept_set_equals_the_references_definition(kw, V, temp):
    from proxy_inference_engine_amd import hip_ops
    rng = np.random.default_rng(V + int(temp * 10) + len(kw))
    lp = _logprobs(rng, V, 3.0 if V > 10000 else 2.0)
    mode, p, k = _mode(kw)
    tokens, kept, mask = hip_ops.sample(torch.from_numpy(lp).cuda(), mode, temp, p=p, k=k, want_mask=True)
    want = _ref_sets(lp, temp, kw.get("top_p", 0.0), kw.get("min_p", 0.0), kw.get("min_tokens_to_keep", 1), kw.get("top_k", -1))
    got = set(np.nonzero(mask[0].cpu().numpy())[0].tolist())
    if mode == "top_p" and got != want:
        # the reference's cumulative sum is a sequential fp32 sum over the sorted row; the kernel's is exact: the sets may differ by the ids
        # whose cumulative mass is within that rounding of 1 - top_p -- at most a handful of boundary ids, contiguous in the sorted order
        x = lp.astype(np.float64) / temp
        pr = np.exp(x - x.max())
        pr /= pr.sum()
        order = np.argsort(pr, kind="stable")
        cum = np.cumsum(pr[order])
        diff = got ^ want
        pos = {int(i): j for j, i in enumerate(order)}
        assert len(diff) <= 3 and all(abs(cum[pos[i]] - (1 - kw["top_p"])) < 2e-5 for i in diff), (len(diff), sorted(diff)[:5])
    else:
        assert got == want, (len(got), len(want), sorted(got ^ want)[:8])
    assert int(kept.item()) == len(got) and int(tokens.item()) in got


@pytest.mark.parametrize("kw", [dict(top_p=0.6), dict(min_p=0.1), dict(top_k=5), dict()])
def test_distribution_seed_and_counter(kw):
    from proxy_inference_engine_amd import samplers
    from proxy_inference_engine_amd.samplers import make_sampler
    rng = np.random.default_rng(11)
    V, temp, rows = 64, 0.8, 4000
    lp = _logprobs(rng, V)
    sampler = make_sampler(temp=temp, **kw)
    x = torch.from_numpy(lp)[None].repeat(rows, 1).cuda()
    samplers.seed(1234)
    draws = sampler(x)
    assert draws.is_cuda and draws.dtype == torch.int32 and draws.shape == (rows,)
    draws = draws.cpu().numpy()
    allowed = _ref_sets(lp, temp, kw.get("top_p", 0.0), kw.get("min_p", 0.0), kw.get("min_tokens_to_keep", 1), kw.get("top_k", -1))
    assert set(draws.tolist()) <= allowed
    p = np.exp(lp.astype(np.float64) / temp)
    mask = np.zeros(V, bool)
    mask[list(allowed)] = True
    p = np.where(mask, p, 0.0)
    p /= p.sum()
    freq = np.bincount(draws, minlength=V) / rows
    assert 0.5 * np.abs(freq - p).sum() < 0.05
    assert all(freq[i] > 0 for i in allowed if p[i] > 0.01)
    again = sampler(x).cpu().numpy()                  # the device-side call counter moved on: a different draw
    assert not np.array_equal(again, draws)
    samplers.seed(1234)                               # seed() restarts the stream
    assert np.array_equal(sampler(x).cpu().numpy(), draws)
    assert np.array_equal(sampler(x).cpu().numpy(), again)


def test_argument_errors_follow_the_reference():
    from proxy_inference_engine_amd.samplers import make_sampler
    x = torch.zeros((1, 16), device="cuda")
    with pytest.raises(ValueError):
        make_sampler(temp=1.0, top_k=16)(x)           # top_k must be < vocab (top_k.py:20-24)
    with pytest.raises(ValueError):
        make_sampler(temp=1.0, min_p=1.5)(x)          # min_p.py:33-36
    with pytest.raises(ValueError):
        make_sampler(temp=1.0, min_p=0.1, min_tokens_to_keep=0)(x)
