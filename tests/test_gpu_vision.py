"""-m gpu: the vision tower -- its four op-level kernels and the whole Qwen2.5-VL tower on the HIP ops
against oracle/vision_oracle.py (the CPU restatement of models/intern/vision.py), then tower -> ensemble -> text tower."""
import numpy as np
import pytest
import torch

from oracle import pie_oracle as po
from oracle import vision_oracle as vo
from tests._util import EPS, assert_bits_close, assert_dot_close, assert_vec_close, to_bits, to_dev

pytestmark = pytest.mark.gpu
DT = "bfloat16"


@pytest.fixture(scope="module")
def ops():
    from proxy_inference_engine_amd import hip_ops
    return hip_ops


def dev(x, dt=DT):
    return to_dev(po.to_bits(x, dt), dt)


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,bias", [(92, 480, 160, True), (23, 96, 640, True), (64, 214, 1176, False), (130, 160, 212, True), (1, 7, 64, True)])
def test_linear_rows_vs_oracle(ops, dt, M, N, K, bias):
    rng = np.random.default_rng(M + N)
    x = po.round_T(rng.standard_normal((M, K)), dt)
    w = po.round_T(rng.standard_normal((N, K)) / np.sqrt(K), dt)
    b = po.round_T(rng.standard_normal(N) * 0.5, dt) if bias else None
    want = po.linear(x, po.to_bits(w, dt), dt, po.to_bits(b, dt) if bias else None)
    got = ops.linear_rows(dev(x, dt), dev(w, dt), dev(b, dt) if bias else None)
    assert got.shape == (M, N)
    assert_dot_close(got.float().cpu().numpy(), want, dt, max_frac=0.03, mag=np.abs(want).max() if bias else None, what=f"linear {M}x{N}x{K} {dt}")


def test_gelu_vs_oracle(ops):
    x = po.round_T(np.random.default_rng(0).standard_normal(5003) * 3, DT)
    assert_bits_close(to_bits(ops.gelu(dev(x))), po.to_bits(vo.gelu(x, DT), DT), max_ulp=1, what="gelu")


@pytest.mark.parametrize("D,DP", [(80, 128), (64, 64), (32, 64), (128, 128), (36, 64), (2, 64)])   # the last two: scalar fallback kernel
def test_vision_qkv_rope_layout_and_values(ops, D, DP):
    rng = np.random.default_rng(D)
    N, H = 37, 3
    qkv = po.round_T(rng.standard_normal((N, 3, H, D)), DT)
    ang = (rng.standard_normal((N, D // 2)) * 3).astype(np.float32)
    q, k, v = ops.vision_qkv_rope(dev(qkv).view(N, -1), torch.from_numpy(np.cos(ang)).cuda(), torch.from_numpy(np.sin(ang)).cuda(), H, DP)
    assert q.shape == (N, H, DP) and k.shape == (H, N, DP) and v.shape == (H, N, DP)
    wq, wk = vo.rope_vision(qkv[:, 0], ang, DT), vo.rope_vision(qkv[:, 1], ang, DT)
    assert np.array_equal(to_bits(q[..., :D]), po.to_bits(wq, DT))
    assert np.array_equal(to_bits(k[..., :D]), po.to_bits(wk.transpose(1, 0, 2), DT))
    assert np.array_equal(to_bits(v[..., :D]), po.to_bits(qkv[:, 2].transpose(1, 0, 2), DT))
    for t in (q, k, v):
        assert not to_bits(t[..., D:]).any(), "padding columns must be zero"
    # the qkv Linear's bias folded in: same values as adding it to the GEMM output first
    b = po.round_T(rng.standard_normal(3 * H * D) * 0.5, DT)
    qkv_b = po.round_T(qkv.reshape(N, -1) + b, DT).reshape(N, 3, H, D)
    q2, k2, v2 = ops.vision_qkv_rope(dev(qkv).view(N, -1), torch.from_numpy(np.cos(ang)).cuda(), torch.from_numpy(np.sin(ang)).cuda(), H, DP, bias=dev(b))
    assert np.array_equal(to_bits(q2[..., :D]), po.to_bits(vo.rope_vision(qkv_b[:, 0], ang, DT), DT))
    assert np.array_equal(to_bits(k2[..., :D]), po.to_bits(vo.rope_vision(qkv_b[:, 1], ang, DT).transpose(1, 0, 2), DT))
    assert np.array_equal(to_bits(v2[..., :D]), po.to_bits(qkv_b[:, 2].transpose(1, 0, 2), DT))


@pytest.mark.parametrize("M,N", [(33, 212), (5, 3421), (64, 1280), (3, 3424)])
def test_bias_folded_elementwise_ops(ops, M, N):
    rng = np.random.default_rng(N)
    g, u, x = (po.round_T(rng.standard_normal((M, N)) * 2, DT) for _ in range(3))
    bg, bu = (po.round_T(rng.standard_normal(N), DT) for _ in range(2))
    want = po.silu_mul(po.round_T(g + bg, DT), po.round_T(u + bu, DT), DT)
    assert_bits_close(to_bits(ops.bias_silu_mul(dev(g), dev(u), dev(bg), dev(bu))), po.to_bits(want, DT), max_ulp=1, max_frac=0.01, what="bias_silu_mul")
    # gate | up as the column halves of one GEMM output
    gu = dev(np.concatenate([g, u], axis=1))
    assert torch.equal(ops.bias_silu_mul(gu[:, :N], gu[:, N:], dev(bg), dev(bu)), ops.bias_silu_mul(dev(g), dev(u), dev(bg), dev(bu)))
    want = po.add(x, po.round_T(g + bg, DT), DT)
    y = ops.add_bias(dev(x), dev(g), dev(bg))
    assert np.array_equal(to_bits(y), po.to_bits(want, DT))
    if N % 8 == 0:      # the fused residual + RMSNorm pass gives exactly the two-pass values
        nw = dev(po.round_T(1 + 0.1 * rng.standard_normal(N), DT))
        y2, xn = ops.add_bias_rms_norm(dev(x), dev(g), dev(bg), nw, 1e-6)
        assert torch.equal(y2, y) and torch.equal(xn, ops.rms_norm(y, nw, 1e-6))


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,I,K", [(130, 212, 160), (300, 3420, 1280), (64, 64, 64), (1100, 1024, 512)])
def test_fused_gate_up_linear_is_the_gemm_followed_by_bias_silu_mul(ops, dt, M, I, K):
    """linear_rows(..., swiglu=True): gate_proj | up_proj as one GEMM on interleaved rows with the biases and silu(gate) * up in its epilogue
    (MLP, vision.py:196-197) -- what the plain GEMM on the concatenated rows followed by bias_silu_mul gives (the same roundings; the plain
    GEMM may split K, which moves an fp32 sum by an ulp here and there), and its zero-padded form feeds the next GEMM unchanged
    (intermediate_size 3420 is not a multiple of 64)."""
    rng = np.random.default_rng(M + I)
    x = dev(po.round_T(rng.standard_normal((M, K)), dt), dt)
    gw, uw = (dev(po.round_T(rng.standard_normal((I, K)) / np.sqrt(K), dt), dt) for _ in range(2))
    gb, ub = (dev(po.round_T(rng.standard_normal(I) * 0.5, dt), dt) for _ in range(2))
    gu = ops.linear_rows(x, torch.cat([gw, uw], dim=0))
    want = ops.bias_silu_mul(gu[:, :I], gu[:, I:], gb, ub)
    pk = ops.pack_linear(ops.interleave_gate_up(gw, uw))
    got = ops.linear_rows(x, pk, ops.interleave_gate_up(gb, ub), swiglu=True)
    assert got.shape == (M, I)
    assert_bits_close(to_bits(got), to_bits(want), max_ulp=2, max_frac=0.01, what="fused gate|up")
    padded = ops.linear_rows(x, pk, ops.interleave_gate_up(gb, ub), swiglu=True, pad_to=64)
    Ip = -(-I // 64) * 64
    assert padded.shape == (M, Ip) and torch.equal(padded[:, :I], got) and not padded[:, I:].any()
    dw = dev(po.round_T(rng.standard_normal((96, I)) / np.sqrt(I), dt), dt)
    assert torch.equal(ops.linear_rows(padded, ops.pack_linear(dw)), ops.linear_rows(got, dw))


def test_linear_w16m_refuses_what_it_cannot_do(ops):
    """pie_linear_w16m through the C ABI: a K-split shape without its workspace, a fused SiLU * up on an odd column count, an x row stride shorter
    than the padded K, and a host tensor are errors (status + pie_last_error), not silent fallbacks."""
    from proxy_inference_engine_amd import _ffi
    lib = _ffi.load()
    M, N, K = 64, 4096, 14336
    need = int(lib.pie_linear_w16m_workspace(M, N, K))
    assert need > 0 and need % (M * N * 4) == 0                      # whole fp32 slabs
    x = torch.zeros((M, K), dtype=torch.bfloat16, device="cuda")
    pk = ops.pack_linear(torch.zeros((N, K), dtype=torch.bfloat16, device="cuda"))
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    args = (_ffi.p(x), K, _ffi.p(pk.tiles), None, M, N, K, _ffi.PIE_BF16, _ffi.p(y), 0, 0)
    assert lib.pie_linear_w16m(*args, None, 0, _ffi.stream()) != 0 and b"workspace" in lib.pie_last_error()
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    assert lib.pie_linear_w16m(*args, _ffi.p(ws), need, _ffi.stream()) == 0
    torch.cuda.synchronize()
    assert not y.any()
    small = ops.pack_linear(torch.zeros((36, 128), dtype=torch.bfloat16, device="cuda"))
    xs = torch.zeros((8, 128), dtype=torch.bfloat16, device="cuda")
    assert lib.pie_linear_w16m(_ffi.p(xs), 128, _ffi.p(small.tiles), None, 8, 36, 128, _ffi.PIE_BF16, _ffi.p(y), 0, 1, None, 0, _ffi.stream()) != 0
    assert b"N % 8" in lib.pie_last_error()
    assert lib.pie_linear_w16m(_ffi.p(xs), 64, _ffi.p(small.tiles), None, 8, 36, 128, _ffi.PIE_BF16, _ffi.p(y), 0, 0, None, 0, _ffi.stream()) != 0
    assert b"x rows" in lib.pie_last_error()
    with pytest.raises(ValueError, match="device"):
        ops.linear_rows(torch.zeros((8, 128), dtype=torch.bfloat16), small)


def _mask(cu, N, dt):
    m = np.full((N, N), vo.finfo_min(dt), np.float32)
    for i in range(1, len(cu)):
        m[cu[i - 1]:cu[i], cu[i - 1]:cu[i]] = 0.0
    return m


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("H,D,cu", [
    (2, 64, [0, 64, 128, 192]),                       # windows aligned with the 32-row tiles
    (3, 128, [0, 40, 44, 108, 109, 300]),            # ragged: tiles span several segments, a 1-row and a 4-row segment
    (16, 128, [0, 1024]),                            # one full-image segment (two query tiles per workgroup)
    (4, 64, [0, 16, 48, 112, 176, 180, 436, 500]),
])
def test_sdpa_segments_vs_oracle(ops, dt, H, D, cu):
    rng = np.random.default_rng(H * D + len(cu))
    N = cu[-1]
    q = po.round_T(rng.standard_normal((N, H, D)), dt)
    k = po.round_T(rng.standard_normal((H, N, D)), dt)
    v = po.round_T(rng.standard_normal((H, N, D)), dt)
    lo, hi = ops.segment_bounds(cu, "cuda")
    got = ops.sdpa_segments(dev(q, dt), dev(k, dt), dev(v, dt), lo, hi, D ** -0.5)
    want = po.sdpa(np.ascontiguousarray(q.transpose(1, 0, 2)), k, v, D ** -0.5, _mask(cu, N, dt), dt, True).transpose(1, 0, 2)
    assert_dot_close(got.float().cpu().numpy(), po.round_T(want, dt), dt, max_frac=0.05, what=f"segments H{H} D{D} {dt}")


def _tower(cfg, seed, dt=DT):
    from proxy_inference_engine_amd.models.intern.vision import VisionConfig, VisionModel
    w = vo.synth_vision_checkpoint(cfg, seed, dt)
    tw = {k: to_dev(v, dt) for k, v in w.items()}
    return w, VisionModel(VisionConfig(**cfg), tw, dtype=torch.bfloat16 if dt == "bfloat16" else torch.float16)


CFG80 = dict(depth=3, hidden_size=160, intermediate_size=212, out_hidden_size=96, num_heads=2, patch_size=14, in_channels=3,
             spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1])      # head_dim 80, like the 7B tower
CFG64 = dict(CFG80, hidden_size=128, intermediate_size=344, num_heads=2, fullatt_block_indexes=[0, 2])     # head_dim 64, no padding


@pytest.mark.parametrize("cfg,grid", [(CFG80, [(1, 6, 10), (2, 4, 4)]), (CFG64, [(1, 16, 12)]), (CFG80, [(1, 2, 2)])])
def test_vision_tower_vs_oracle(cfg, grid):
    w, model = _tower(cfg, seed=4)
    N = sum(t * h * ww for t, h, ww in grid)
    pix = po.round_T(np.random.default_rng(7).standard_normal((N, 3 * 2 * 14 * 14)), DT)
    want, states = vo.vision_forward(cfg, w, pix, grid, DT, want_states=True)
    got = model(dev(pix), torch.tensor(grid), output_hidden_states=False)
    assert got.shape == (N // 4, cfg["out_hidden_size"])
    assert_vec_close(got.float().cpu().numpy(), want, DT, c_max=6.0, c_rms=5.0, what=f"tower output {grid}")
    # per-block hidden states against the oracle's, and the hipGraph replay against the eager run (bit for bit)
    got2, got_states = model(dev(pix), torch.tensor(grid), output_hidden_states=True)
    assert torch.equal(got2, got) and len(got_states) == len(states)
    for i, (a, b) in enumerate(zip(got_states, states)):
        assert_vec_close(a.float().cpu().numpy(), b, DT, c_max=6.0, c_rms=5.0, what=f"hidden state {i}")
    for _ in range(3):                                       # eager, capture, replay
        rep = model(dev(pix), torch.tensor(grid), graph=True)
        assert torch.equal(rep, got)
    other = po.round_T(np.random.default_rng(8).standard_normal((N, 3 * 2 * 14 * 14)), DT)
    assert torch.equal(model(dev(other), torch.tensor(grid), graph=True), model(dev(other), torch.tensor(grid)))


# BASELINE.json configs[3] at ITS geometry: the Qwen2.5-VL-7B tower's shapes -- hidden 1280, 16 heads of 80 (zero-padded
# to 128 inside the attention kernel), SwiGLU 3420 (not a multiple of 64: the K = 3420 GEMM rows), patch 14 x 14 x 2 frames, 112-pixel windows,
# merger 5120 -> 3584 -- with 2 of its 32 blocks: one windowed, one full-attention (vision.py:87-442).
CFG7B = dict(depth=2, hidden_size=1280, intermediate_size=3420, out_hidden_size=3584, num_heads=16, patch_size=14, in_channels=3,
             spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1])


@pytest.mark.parametrize("grid", [[(1, 16, 16)], [(1, 32, 32)]])
def test_vision_tower_at_the_qwen25vl_7b_geometry(grid):
    w, model = _tower(CFG7B, seed=11)
    N = sum(t * h * ww for t, h, ww in grid)
    pix = po.round_T(np.random.default_rng(N).standard_normal((N, 3 * 2 * 14 * 14)), DT)
    want, states = vo.vision_forward(CFG7B, w, pix, grid, DT, want_states=True)
    got, got_states = model(dev(pix), torch.tensor(grid), output_hidden_states=True)
    assert got.shape == (N // 4, 3584) and len(got_states) == len(states)
    for i, (a, b) in enumerate(zip(got_states, states)):
        assert_vec_close(a.float().cpu().numpy(), b, DT, c_max=6.0, c_rms=5.0, what=f"7B-geometry tower, grid {grid}, hidden state {i}")
    assert_vec_close(got.float().cpu().numpy(), want, DT, c_max=6.0, c_rms=5.0, what=f"7B-geometry tower output, grid {grid}")
    for _ in range(3):                                       # eager, capture, replay: bit for bit
        assert torch.equal(model(dev(pix), torch.tensor(grid), graph=True), got)


def test_image_at_7b_geometry_into_a_256_token_prompt_through_the_engine():
    """configs[3] end to end at geometry: a 16 x 16-patch image through the 7B-shaped tower (64 image tokens), scattered into a 256-token prompt,
    through InferenceEngine.generate_step(prompt_ids, pixel_values=...) (inference_engine.py:228-252; intern/ensemble.py:62-91) into a text tower
    with Qwen2-VL-7B's layer geometry (hidden 3584, 28 / 4 heads of 128, MLP 18944, q/k/v bias; ONE layer and an 8192-row vocabulary to keep the
    CPU oracle in seconds).  First-token logits against the two oracles chained, then greedy tokens while the oracle's margins are safe."""
    from proxy_inference_engine_amd import InferenceEngine
    from proxy_inference_engine_amd.models.intern import Model as Ensemble, ModelArgs as EnsembleArgs
    from tests.test_gpu_decode import build, margin_bound
    tcfg = {"model_type": "llama", "hidden_size": 3584, "num_hidden_layers": 1, "intermediate_size": 18944, "num_attention_heads": 28,
            "num_key_value_heads": 4, "rms_norm_eps": 1e-6, "vocab_size": 8192, "rope_theta": 1000000.0, "max_position_embeddings": 32768,
            "tie_word_embeddings": False, "attention_bias": True, "quantization": {"group_size": 64, "bits": 4}}
    tw = po.synth_checkpoint(tcfg, seed=9, dtype=DT, lm_head_gain=4.0)
    lm = build(tcfg, tw, DT)
    vw, tower = _tower(CFG7B, seed=12)
    grid = [(1, 16, 16)]
    n_img, IMG = 16 * 16 // 4, 7
    rng = np.random.default_rng(13)
    ids = np.concatenate([rng.integers(10, 8000, 96), np.full(n_img, IMG), rng.integers(10, 8000, 256 - 96 - n_img)]).astype(np.int64)
    assert len(ids) == 256
    pix = po.round_T(rng.standard_normal((256, 1176)), DT)
    ens = Ensemble(EnsembleArgs(image_token_id=IMG, video_token_id=IMG + 1), lm, vision_tower=lambda pv, g: tower(pv, torch.tensor(grid)))
    eng = InferenceEngine(model=ens)
    eng.prepare_engine(ids, temp=0)
    gen = eng.generate_step(torch.from_numpy(ids), pixel_values=dev(pix))
    feats = vo.vision_forward(CFG7B, vw, pix, grid, DT)
    table = po.dequantize(tw["model.embed_tokens.weight"], tw["model.embed_tokens.scales"], tw["model.embed_tokens.biases"], dtype=DT)[ids].copy()
    table[ids == IMG] = feats
    orc = po.OracleLlama(tcfg, tw, DT)
    ocache = [po.OracleKVCache() for _ in orc.layers]
    want = orc.forward(None, ocache, inputs_embeds=table, last_only=True)
    checked = 0
    for step in range(4):
        tok, lp = next(gen)
        olp = po.logprobs_argmax(want)[1]
        # the image rows carry the tower's own rounding noise into the text tower: one more factor on the end-to-end bound
        assert float(np.abs(lp.cpu().numpy() - olp).max()) <= 8.0 * EPS[DT] * float(np.abs(want).max()) + 1e-3, f"step {step}: log-probabilities"
        top2 = np.sort(want)[-2:]
        if top2[1] - top2[0] <= 2.0 * margin_bound(want):
            break
        assert int(tok.item()) == int(np.argmax(want)), f"step {step}"
        checked += 1
        want = orc.forward(np.array([int(tok.item())]), ocache, last_only=True)
    assert checked >= 1
    assert eng.prompt_cache.computed_ids[:len(ids)] == [int(i) for i in ids]


def test_vision_tower_mlx_ordered_conv_weight_and_errors():
    from proxy_inference_engine_amd.models.intern.vision import VisionConfig, VisionModel
    w = vo.synth_vision_checkpoint(CFG80, 1, DT)
    tw = {k: to_dev(v, DT) for k, v in w.items()}
    a = VisionModel(VisionConfig(**CFG80), tw)
    tw2 = dict(tw)
    tw2["vision_tower.patch_embed.proj.weight"] = tw["vision_tower.patch_embed.proj.weight"].permute(0, 2, 3, 4, 1).contiguous()   # MLX order
    b = VisionModel(VisionConfig(**CFG80), tw2)
    pa, pb = a.patch_w.tiles, b.patch_w.tiles   # the W16M tiles of the flattened conv weight
    assert torch.equal(pa, pb)
    with pytest.raises(ValueError, match="grid_thw must be provided"):
        a(torch.zeros(4, 1176, device="cuda"))
    with pytest.raises(ValueError, match="do not match"):
        a(torch.zeros(5, 1176, device="cuda"), [(1, 2, 2)])


def test_image_to_text_logits_end_to_end():
    """pixels -> vision tower -> image-token scatter -> text tower, all on the device, against the two oracles chained."""
    from proxy_inference_engine_amd.models.intern import Model as Ensemble, ModelArgs as EnsembleArgs
    from tests.test_gpu_decode import build
    tcfg = {"model_type": "llama", "hidden_size": 256, "num_hidden_layers": 2, "intermediate_size": 704, "num_attention_heads": 4,
            "num_key_value_heads": 2, "rms_norm_eps": 1e-6, "vocab_size": 512, "rope_theta": 1000000.0, "tie_word_embeddings": False,
            "attention_bias": True, "quantization": {"group_size": 64, "bits": 4}}
    vcfg = dict(CFG80, out_hidden_size=256, depth=2)
    tw = po.synth_checkpoint(tcfg, seed=3, dtype=DT, lm_head_gain=4.0)
    lm = build(tcfg, tw, DT)
    vw, tower = _tower(vcfg, seed=6)
    grid = [(1, 4, 6)]
    n_img = 4 * 6 // 4
    rng = np.random.default_rng(5)
    ids = np.concatenate([rng.integers(10, 500, 5), np.full(n_img, 7), rng.integers(10, 500, 4)]).astype(np.int64)
    pix = po.round_T(rng.standard_normal((24, 1176)), DT)
    ens = Ensemble(EnsembleArgs(image_token_id=7, video_token_id=8), lm, vision_tower=tower)
    logits = ens(torch.from_numpy(ids)[None].cuda(), pixel_values=dev(pix), image_grid_thw=torch.tensor(grid))
    feats = vo.vision_forward(vcfg, vw, pix, grid, DT)
    table = po.dequantize(tw["model.embed_tokens.weight"], tw["model.embed_tokens.scales"], tw["model.embed_tokens.biases"], dtype=DT)[ids].copy()
    table[ids == 7] = feats
    orc = po.OracleLlama(tcfg, tw, DT)
    want = orc.forward(None, [po.OracleKVCache() for _ in orc.layers], inputs_embeds=table)
    # the image rows carry the tower's own rounding noise into the text tower: one more factor on the end-to-end bound
    assert_vec_close(logits[0, -1].float().cpu().numpy(), want[-1], DT, c_max=8.0, c_rms=6.0, what="image -> text logits")


def test_vision_tower_and_few_row_gemm_against_golden_fixture(ops, golden_dir):
    """The committed oracle-captured vectors (tests/golden/tiny_vision_bf16.npz): the tower's features and a 17-row int4 product."""
    import json
    from proxy_inference_engine_amd.models.intern.vision import VisionConfig, VisionModel
    from tests._util import codes_dev
    g = np.load(golden_dir / "tiny_vision_bf16.npz")
    cfg = json.loads(str(g["config_json"]))
    tw = {k[2:]: to_dev(g[k], DT) for k in g.files if k.startswith("w:")}
    model = VisionModel(VisionConfig(**cfg), tw)
    got = model(to_dev(g["pixels"], DT), torch.from_numpy(g["grid"]))
    assert_vec_close(got.float().cpu().numpy(), po.from_bits(g["features"], DT), DT, c_max=6.0, c_rms=5.0, what="golden tower features")
    packed = ops.repack_w4s(codes_dev(g["qmm_wq"]), to_dev(g["qmm_scales"], DT), to_dev(g["qmm_biases"], DT))
    y = ops.quantized_matmul_rows(to_dev(g["qmm_x"], DT), packed)
    assert_dot_close(y.float().cpu().numpy(), po.from_bits(g["qmm_y"], DT), DT, max_frac=0.03, what="golden few-row product")
