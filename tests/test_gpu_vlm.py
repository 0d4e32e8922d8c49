"""-m gpu: the VLM text-tower entry (SURVEY.md 8d config C4, row f3's language side): prompts given as input embeddings
-- text embeddings with an image-token span replaced by image features -- through pie_decoder_prefill_embeds, against the
oracle's `h = inputs_embeds` path (models/intern/language.py:155-158), then decode steps on top of that cache.
The vision tower itself is not built; features are synthetic rows, as C4 defines."""
import numpy as np
import pytest
import torch

from oracle import pie_oracle as po
from tests._util import assert_vec_close, to_bits, to_dev
from tests.test_gpu_decode import build

pytestmark = pytest.mark.gpu
DT = "bfloat16"
IMG = 151            # image_token_id inside the test vocabulary


def _qv_cfg(layers=2, bits=4):
    return {"model_type": "llama", "hidden_size": 512, "num_hidden_layers": layers, "intermediate_size": 1408,
            "num_attention_heads": 8, "num_key_value_heads": 2, "rms_norm_eps": 1e-6, "vocab_size": 1024,
            "rope_theta": 1000000.0, "max_position_embeddings": 32768, "tie_word_embeddings": False, "attention_bias": True,
            "quantization": {"group_size": 64, "bits": bits}}


def _setup(seed=3, **kw):
    from proxy_inference_engine_amd.models.intern import Model as Ensemble, ModelArgs as EnsembleArgs
    cfg = _qv_cfg(**kw)
    w = po.synth_checkpoint(cfg, seed=seed, dtype=DT, lm_head_gain=4.0)
    lm = build(cfg, w, DT)
    return cfg, w, lm, Ensemble(EnsembleArgs(image_token_id=IMG, video_token_id=IMG + 1), lm), po.OracleLlama(cfg, w, DT)


def _embed(w, ids):
    """nn.QuantizedEmbedding: the dequantised rows (models/utils.py:99-109 quantises the embedding too)."""
    return po.dequantize(w["model.embed_tokens.weight"], w["model.embed_tokens.scales"], w["model.embed_tokens.biases"], dtype=DT)[np.asarray(ids)].copy()


def _prompt(rng, cfg, n_text_a, n_img, n_text_b, tok=IMG):
    ids = np.concatenate([rng.integers(200, cfg["vocab_size"], n_text_a), np.full(n_img, tok), rng.integers(200, cfg["vocab_size"], n_text_b)])
    feats = po.round_T(rng.standard_normal((n_img, cfg["hidden_size"])) * 0.05, DT)
    return ids.astype(np.int64), feats


@pytest.mark.parametrize("n_a,n_img,n_b", [(5, 16, 7), (0, 64, 3), (40, 1, 0), (1, 2, 1)])
def test_prefill_from_merged_embeddings_vs_oracle(n_a, n_img, n_b):
    cfg, w, lm, ens, orc = _setup()
    rng = np.random.default_rng(n_img)
    ids, feats = _prompt(rng, cfg, n_a, n_img, n_b)
    # device: embed + scatter + text tower
    emb = ens.merge_image_features(torch.from_numpy(ids).cuda(), to_dev(po.to_bits(feats, DT), DT))
    assert emb.shape == (1, len(ids), cfg["hidden_size"])
    # oracle: the same merge restated on the host (ensemble.py:62-91), then h = inputs_embeds
    table = _embed(w, ids)
    table[ids == IMG] = feats
    assert np.array_equal(to_bits(emb[0]), po.to_bits(table, DT)), "merged embeddings must be exact (gather + scatter)"
    ocache = [po.OracleKVCache() for _ in orc.layers]
    want = orc.forward(None, ocache, inputs_embeds=table)
    cache = lm.make_cache()
    logits = lm(None, cache=cache, inputs_embeds=emb)
    assert logits.shape == (1, len(ids), cfg["vocab_size"]) and cache[0].offset == len(ids)
    for l in range(len(ids)):
        assert_vec_close(logits[0, l].float().cpu().numpy(), want[l], DT, what=f"position {l}")
    # decode continues on the cache the embeddings filled
    tok = int(np.argmax(want[-1]))
    for _ in range(3):
        want1 = orc.forward(np.array([tok]), ocache)[0]
        got_tok, _, got = lm.step(torch.tensor([tok], dtype=torch.int32, device="cuda"), cache)
        assert_vec_close(got.float().cpu().numpy(), want1, DT, what="decode after embeds")
        tok = int(np.argmax(want1))


def test_ensemble_call_and_video_token_fallback():
    cfg, w, lm, ens, orc = _setup(seed=8, layers=1)
    rng = np.random.default_rng(1)
    ids, feats = _prompt(rng, cfg, 4, 9, 4, tok=IMG + 1)                 # no image token: the video token id is used (ensemble.py:72-76)
    calls = []

    def tower(pixel_values, grid_thw):
        calls.append((tuple(pixel_values.shape), grid_thw))
        return to_dev(po.to_bits(feats, DT), DT)                          # [N, hidden]; the ensemble adds the batch axis (:53-54)

    ens.vision_tower = tower
    pix = torch.zeros(9 * 4, 1176, device="cuda")
    logits = ens(torch.from_numpy(ids)[None].cuda(), pixel_values=pix, cache=None, image_grid_thw="grid")
    assert calls == [((36, 1176), "grid")]
    table = _embed(w, ids)
    table[ids == IMG + 1] = feats
    want = orc.forward(None, [po.OracleKVCache() for _ in orc.layers], inputs_embeds=table)
    assert_vec_close(logits[0, -1].float().cpu().numpy(), want[-1], DT, what="ensemble logits")
    # text-only call: plain ids through the same object
    ids2 = rng.integers(200, cfg["vocab_size"], 12)
    want2 = orc.forward(ids2, [po.OracleKVCache() for _ in orc.layers])
    got2 = ens(torch.from_numpy(ids2)[None].cuda())
    assert_vec_close(got2[0, -1].float().cpu().numpy(), want2[-1], DT, what="text-only")
    assert len(ens.layers) == 1 and ens.head_dim == 64 and ens.n_kv_heads == 2


def test_engine_generate_step_with_pixel_values():
    """InferenceEngine.generate_step(prompt_ids, pixel_values=...) (inference_engine.py:228-252): the prompt goes through the
    ensemble once, later steps are ordinary decode steps; greedy tokens follow the oracle while its margins are safe."""
    from proxy_inference_engine_amd import InferenceEngine
    from tests.test_gpu_decode import margin_bound
    cfg, w, lm, ens, orc = _setup(seed=5)
    rng = np.random.default_rng(2)
    ids, feats = _prompt(rng, cfg, 6, 20, 6)
    ens.vision_tower = lambda pv, grid: to_dev(po.to_bits(feats, DT), DT)[None]
    eng = InferenceEngine(model=ens)
    eng.prepare_engine(ids, temp=0)
    gen = eng.generate_step(torch.from_numpy(ids), pixel_values=torch.zeros(4, 4, device="cuda"))
    table = _embed(w, ids)
    table[ids == IMG] = feats
    ocache = [po.OracleKVCache() for _ in orc.layers]
    want = orc.forward(None, ocache, inputs_embeds=table)[-1]
    for step in range(6):
        tok, lp = next(gen)
        top2 = np.sort(want)[-2:]
        if top2[1] - top2[0] <= margin_bound(want):
            break
        assert int(tok.item()) == int(np.argmax(want)), f"step {step}"
        want = orc.forward(np.array([int(tok.item())]), ocache)[0]
    assert step >= 1
    assert eng.prompt_cache.computed_ids[:len(ids)] == [int(i) for i in ids]


def test_embeds_argument_errors():
    cfg, w, lm, ens, orc = _setup(seed=8, layers=1)
    with pytest.raises(ValueError, match="Either inputs or inputs_embeds"):
        lm(None)
    with pytest.raises(ValueError):
        lm(None, inputs_embeds=torch.zeros(1, 4, 100, device="cuda"))
    with pytest.raises(NotImplementedError, match="no vision tower"):
        ens(torch.zeros(1, 4, dtype=torch.int64, device="cuda"), pixel_values=torch.zeros(1, device="cuda"))
    ids = torch.tensor([IMG, IMG, 300], device="cuda")
    with pytest.raises(ValueError, match="2 image tokens"):
        ens.merge_image_features(ids, torch.zeros(3, cfg["hidden_size"], device="cuda"))
