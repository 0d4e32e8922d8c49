"""-m gpu: each HIP op behind the C ABI against the CPU oracle and the committed golden vectors.

Integer work (codes, packing, argmax index) must be bit-exact; 16-bit float outputs must be within 1 ulp of the
oracle on at most 2 % of elements (fp32 accumulation order differs), fp32 logprobs within 1e-4 absolute."""
import numpy as np
import pytest
import torch

from oracle import pie_oracle as po
from tests._util import assert_bits_close, assert_dot_close, codes_dev, to_bits, to_dev

pytestmark = pytest.mark.gpu
DT = "bfloat16"


@pytest.fixture(scope="module")
def ops():
    from proxy_inference_engine_amd import _ffi, hip_ops
    _ffi.require_gpu()
    assert _ffi.hello() == "pie_core ✓"
    return hip_ops


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir / "ops_bf16.npz")


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
def test_quantize_bit_exact_and_dequantize(ops, dt):
    rng = np.random.default_rng(5)
    w = po.round_T(rng.standard_normal((96, 512)) * 0.02, dt)
    w[3, :64] = 0.0                       # degenerate group: scale = eps path
    w[4, :64] = 0.25                      # constant group
    wq, s, b = po.quantize(w, 64, 4, dt)
    codes, scales, biases = ops.quantize(to_dev(po.to_bits(w, dt), dt))
    assert np.array_equal(codes.cpu().numpy().view(np.uint32), wq)
    assert np.array_equal(to_bits(scales), s) and np.array_equal(to_bits(biases), b)
    deq = ops.dequantize(codes, scales, biases)
    assert np.array_equal(to_bits(deq), po.to_bits(po.dequantize(wq, s, b, 64, 4, dt), dt))


@pytest.mark.parametrize("tag", ["gemv4096", "gemv14336"])
def test_qgemv_golden(ops, gold, tag):
    w = ops.repack_w4s(codes_dev(gold[f"{tag}_wq"]), to_dev(gold[f"{tag}_scales"], DT), to_dev(gold[f"{tag}_biases"], DT))
    y = ops.quantized_matmul(to_dev(gold[f"{tag}_x"], DT), w)
    assert_bits_close(to_bits(y), gold[f"{tag}_y"], what=tag)


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("N,K,M", [(64, 64, 1), (6, 192, 2), (130, 2048, 1), (34, 2112, 3), (256, 5632, 1), (32, 8192, 1), (18, 28672, 1)])
def test_qgemv_shapes_vs_oracle(ops, dt, N, K, M):
    """Ragged shapes: K not a multiple of the 2048 slice, odd pair counts, several rows of x, a linear bias."""
    rng = np.random.default_rng(N * 7 + K)
    w = po.round_T(rng.standard_normal((N, K)) * 0.03, dt)
    wq, s, b = po.quantize(w, 64, 4, dt)
    x = po.round_T(rng.standard_normal((M, K)), dt)
    lb = po.to_bits(rng.standard_normal(N) * 0.1, dt) if N % 4 == 0 else None
    want = po.quantized_matmul(x, wq, s, b, dtype=dt, lin_bias=lb)
    wt = ops.repack_w4s(codes_dev(wq), to_dev(s, dt), to_dev(b, dt), lin_bias=None if lb is None else to_dev(lb, dt))
    got = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), wt)
    assert got.shape == (M, N)
    assert_bits_close(to_bits(got), po.to_bits(want, dt), what=f"qgemv {N}x{K} M={M} {dt}")


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("N,K", [(2, 64), (6, 192), (34, 2112), (130, 2048), (4098, 4096), (258, 5632), (66, 8192), (40, 14336), (4100, 14336), (20, 16384),
                                 (18, 28672)])
def test_qgemv_several_rows_in_one_pass_equal_each_row_alone(ops, dt, N, K):
    """k_w4s_gemv_rows (pie_qgemv_w4g64 from two rows on: one pass over the weights per up to 5 rows) against the batch-1 kernel on each
    row alone: BIT-identical, for 1..8 and 14 K slices (incl. ragged last slices), pair counts that leave waves without a pair, several
    pairs per wave (N = 4098 / 4100 on 2048 waves), row counts that split into chunks of rows (7 -> 4 + 3, 11 -> 4 + 4 + 3; K = 14336:
    chunks of three), a linear bias."""
    rng = np.random.default_rng(N + K)
    w = po.round_T(rng.standard_normal((N, K)) * 0.03, dt)
    wq, s, b = po.quantize(w, 64, 4, dt)
    lb = po.to_bits(rng.standard_normal(N) * 0.1, dt) if N % 4 == 0 else None
    wt = ops.repack_w4s(codes_dev(wq), to_dev(s, dt), to_dev(b, dt), lin_bias=None if lb is None else to_dev(lb, dt))
    x = po.round_T(rng.standard_normal((11, K)) * np.exp(rng.standard_normal((11, 1))), dt)
    xd = to_dev(po.to_bits(x, dt), dt)
    alone = torch.cat([ops.quantized_matmul(xd[i:i + 1], wt) for i in range(11)])
    for M in (2, 3, 4, 5, 7, 11):
        got = ops.quantized_matmul(xd[:M], wt)
        assert got.shape == (M, N)
        assert np.array_equal(to_bits(got), to_bits(alone[:M])), f"{N}x{K} {dt}: {M} rows in one pass differ from the rows alone"


def test_qgemv_row_map_and_linearity(ops):
    """row_map reorders rows; y(x1 + x2) == y(x1) + y(x2) up to rounding (size-independent property)."""
    rng = np.random.default_rng(9)
    N, K = 128, 4096
    w = po.round_T(rng.standard_normal((N, K)) * 0.02, DT)
    wq, s, b = po.quantize(w, 64, 4, DT)
    perm = rng.permutation(N).astype(np.int32)
    wt = ops.repack_w4s(codes_dev(wq), to_dev(s, DT), to_dev(b, DT), row_map=torch.from_numpy(perm))
    x = po.round_T(rng.standard_normal((1, K)), DT)
    got = ops.quantized_matmul(to_dev(po.to_bits(x, DT), DT), wt)
    want = po.quantized_matmul(x, wq, s, b, dtype=DT)[:, perm]
    assert_bits_close(to_bits(got), po.to_bits(want, DT), what="row_map")
    x2 = po.round_T(rng.standard_normal((1, K)), DT)
    xs = po.round_T(x + x2, DT)
    ya, yb, ys = (ops.quantized_matmul(to_dev(po.to_bits(v, DT), DT), wt).float().cpu().numpy() for v in (x, x2, xs))
    wd = po.dequantize(wq, s, b, 64, 4, "float32" if False else DT).astype(np.float64)[perm]   # T-rounded affine weights
    resid = (xs.astype(np.float64) - x - x2) @ wd.T                                              # rounding of x1+x2 to T
    assert np.max(np.abs(ys - ya - yb - resid)) <= 4 * 2.0 ** -8 * np.abs(ys).max()


def test_embedding_exact(ops):
    rng = np.random.default_rng(2)
    w = po.round_T(rng.standard_normal((100, 256)) * 0.02, DT)
    wq, s, b = po.quantize(w, 64, 4, DT)
    ids = np.array([0, 99, 17, 17], np.int32)
    got = ops.embedding(torch.from_numpy(ids).cuda(), codes_dev(wq), to_dev(s, DT), to_dev(b, DT))
    assert np.array_equal(to_bits(got), po.to_bits(po.dequantize(wq, s, b, 64, 4, DT)[ids], DT))


def test_rms_norm_golden(ops, gold):
    y = ops.rms_norm(to_dev(gold["rms_x"], DT), to_dev(gold["rms_w"], DT), float(gold["rms_eps"]))
    assert_bits_close(to_bits(y), gold["rms_y"], what="rms_norm")


def test_rope_golden_and_identity(ops, gold):
    freqs = torch.from_numpy(gold["rope_freqs"]).cuda()
    x = to_dev(gold["rope_x"], DT)
    y = ops.rope(x, 128, offset=int(gold["rope_offset"]), freqs=freqs)
    assert_bits_close(to_bits(y), gold["rope_y"], what="rope")
    assert np.array_equal(to_bits(ops.rope(x, 128, offset=0, freqs=freqs)), gold["rope_x"])   # position 0 = identity
    xl = po.round_T(np.random.default_rng(3).standard_normal((4, 5, 64)), DT)                 # L > 1, D = 64
    f64 = po.llama3_rope_freqs(64, 10000.0)
    got = ops.rope(to_dev(po.to_bits(xl, DT), DT), 64, offset=250, freqs=torch.from_numpy(f64).cuda())
    assert_bits_close(to_bits(got), po.to_bits(po.rope(xl, f64, 250, DT), DT), what="rope L>1")


def test_sdpa_decode_golden(ops, gold):
    q = to_dev(gold["sdpa_q"], DT).view(1, 8, 1, 128)
    k = to_dev(gold["sdpa_k"], DT).view(1, 2, 512, 128)
    v = to_dev(gold["sdpa_v"], DT).view(1, 2, 512, 128)
    o = ops.scaled_dot_product_attention(q, k, v, float(gold["sdpa_scale"]), T=int(gold["sdpa_T"]))
    assert_bits_close(to_bits(o), gold["sdpa_out"], what="sdpa")


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("Hq,Hkv,D,T,cap", [(4, 2, 64, 1, 256), (8, 8, 64, 37, 256), (32, 4, 64, 300, 512), (32, 8, 128, 2500, 2560), (8, 1, 128, 129, 256)])
def test_sdpa_decode_shapes_vs_oracle(ops, dt, Hq, Hkv, D, T, cap):
    rng = np.random.default_rng(Hq + T)
    q = po.round_T(rng.standard_normal((Hq, 1, D)), dt)
    k = po.round_T(rng.standard_normal((Hkv, cap, D)), dt)
    v = po.round_T(rng.standard_normal((Hkv, cap, D)), dt)
    k[:, T:] = 1e4                                                   # stale rows past T must not be attended
    want = po.sdpa(q, k, v, D ** -0.5, None, dt, True, T=T)
    got = ops.scaled_dot_product_attention(to_dev(po.to_bits(q, dt), dt).view(1, Hq, 1, D), to_dev(po.to_bits(k, dt), dt).view(1, Hkv, cap, D),
                                           to_dev(po.to_bits(v, dt), dt).view(1, Hkv, cap, D), D ** -0.5, T=T)
    assert_bits_close(to_bits(got), po.to_bits(want, dt), max_ulp=2 if dt == "bfloat16" else 4, what=f"sdpa {Hq}/{Hkv} D{D} T{T} {dt}")


def test_sdpa_online_softmax_rescale_forced(ops):
    """A late key with a far larger score forces the running-max rescale in every lane group and split."""
    rng = np.random.default_rng(4)
    Hq, Hkv, D, T = 8, 2, 128, 777
    q = po.round_T(rng.standard_normal((Hq, 1, D)), DT)
    k = po.round_T(rng.standard_normal((Hkv, 1024, D)) * 0.1, DT)
    v = po.round_T(rng.standard_normal((Hkv, 1024, D)), DT)
    for pos in (5, 300, 776):
        k[:, pos] = po.round_T(q[::4, 0] * (3.0 + pos / 100), DT)
    want = po.sdpa(q, k, v, D ** -0.5, None, DT, True, T=T)
    got = ops.scaled_dot_product_attention(to_dev(po.to_bits(q, DT), DT).view(1, Hq, 1, D), to_dev(po.to_bits(k, DT), DT).view(1, Hkv, 1024, D),
                                           to_dev(po.to_bits(v, DT), DT).view(1, Hkv, 1024, D), D ** -0.5, T=T)
    assert_bits_close(to_bits(got), po.to_bits(want, DT), max_ulp=2, what="sdpa spike")


def test_silu_mul_add_golden(ops, gold):
    a, b = to_dev(gold["act_a"], DT), to_dev(gold["act_b"], DT)
    assert_bits_close(to_bits(ops.silu_mul(a, b)), gold["act_silu_mul"], max_frac=0.01, what="silu_mul")
    assert np.array_equal(to_bits(ops.add(a, b)), gold["act_add"])
    assert np.array_equal(to_bits(ops.add(a[:13 * 8 + 3].clone(), b[:13 * 8 + 3].clone())), gold["act_add"][:107])   # ragged tail


def test_logits_tail_golden_and_ties(ops, gold):
    tok, lp = ops.logprobs_argmax(to_dev(gold["tail_logits"], DT))
    assert int(tok.item()) == int(gold["tail_token"])
    # fp32 logsumexp over 128256 terms: the oracle's sequential sum itself carries ~1e-5 relative error
    assert np.max(np.abs(lp.cpu().numpy() - gold["tail_logprobs"])) <= 1e-4
    x = np.zeros(5000, np.float32)
    x[[4000, 777, 4999]] = 3.0                                       # ties: first maximal index wins
    tok, lp = ops.logprobs_argmax(to_dev(po.to_bits(x, DT), DT))
    assert int(tok.item()) == 777
    assert abs(float(torch.exp(lp.double()).sum()) - 1.0) < 1e-5


def test_error_conventions(ops):
    """Bad arguments raise ValueError (PIE_E_ARG/SHAPE/ALIGN), like the reference's sampler/loader checks."""
    with pytest.raises(ValueError):
        ops.quantize(torch.zeros((4, 100), dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(ValueError):
        ops.quantize(torch.zeros((4, 128), dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        ops.rms_norm(torch.zeros((1, 64), dtype=torch.bfloat16), torch.zeros(64, dtype=torch.bfloat16), 1e-5)  # CPU tensor
    with pytest.raises(NotImplementedError):
        q = torch.zeros((1, 4, 2, 64), dtype=torch.bfloat16, device="cuda")
        ops.scaled_dot_product_attention(q, q, q, 0.125)


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("N,K,M", [(64, 512, 1), (130, 704, 3), (4096, 4096, 1), (256, 5632, 2)])
def test_dense_gemv_and_embedding_vs_oracle(ops, dt, N, K, M):
    """nn.Linear on the W16S stream (pie_repack_dense + pie_gemv_dense) against orc_linear, incl. K that is not a multiple
    of the 512-wide slice (704, 5632), a row count that is not a multiple of the wave geometry, a bias, and a row map."""
    rng = np.random.default_rng(N + K)
    w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
    x = po.round_T(rng.standard_normal((M, K)), dt)
    b = po.round_T(rng.standard_normal(N), dt)
    wd, xd = to_dev(po.to_bits(w, dt), dt), to_dev(po.to_bits(x, dt), dt)
    got = ops.linear(xd, ops.repack_dense(wd, lin_bias=to_dev(po.to_bits(b, dt), dt)))
    want = po.linear(x, po.to_bits(w, dt), dt, lin_bias=po.to_bits(b, dt))
    assert_dot_close(got.float().cpu().numpy(), want, dt, what=f"dense gemv {N}x{K} M={M} {dt}", mag=want - b[None, :])
    perm = torch.from_numpy(rng.permutation(N).astype(np.int32))
    got = ops.linear(xd, ops.repack_dense(wd, row_map=perm))
    want = po.linear(x, po.to_bits(w[perm.numpy()], dt), dt)
    assert_dot_close(got.float().cpu().numpy(), want, dt, what="dense gemv with row map")
    if K % 8 == 0 and N >= 64:
        ids = torch.tensor([0, N - 1, 5, 5], dtype=torch.int32, device="cuda")
        rows = ops.embedding_dense(ids, wd)
        assert np.array_equal(to_bits(rows), po.to_bits(w[[0, N - 1, 5, 5]], dt))


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
def test_int8_g64_quantize_gemv_embedding_vs_oracle(ops, dt):
    """MLX 8-bit group-64 triplets (config "quantization": {"bits": 8}): the HIP quantiser is bit-identical to the oracle's
    mx.quantize restatement (codes, scales, biases), dequantise and the embedding gather are exact, and the W8S streaming
    GEMV (bytes fed to v_dot2 as bf16 numbers q * 2^-133 / f16 1024+q) is within one ulp of orc_quantized_matmul_t."""
    rng = np.random.default_rng(88)
    for N, K, M in ((96, 256, 2), (4096, 4096, 1), (130, 704, 3)):
        w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
        x = po.round_T(rng.standard_normal((M, K)), dt)
        wq, sc, bi = po.quantize(w, 64, 8, dt)
        codes, scales, biases = ops.quantize(to_dev(po.to_bits(w, dt), dt), bits=8)
        assert np.array_equal(codes.cpu().numpy().view(np.uint32), wq) and np.array_equal(to_bits(scales), sc) and np.array_equal(to_bits(biases), bi)
        deq = ops.dequantize(codes, scales, biases, bits=8)
        assert np.array_equal(to_bits(deq), po.to_bits(po.dequantize(wq, sc, bi, 64, 8, dt), dt))
        want = po.quantized_matmul(x, wq, sc, bi, group_size=64, bits=8, dtype=dt)
        got = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), ops.repack_w8s(codes, scales, biases))
        assert_dot_close(got.float().cpu().numpy(), want, dt, what=f"int8 gemv {N}x{K} M={M} {dt}")
        if N >= 96:
            ids = torch.tensor([1, N - 1, 7], dtype=torch.int32, device="cuda")
            rows = ops.embedding(ids, codes, scales, biases, bits=8)
            assert np.array_equal(to_bits(rows), to_bits(deq[[1, N - 1, 7]]))


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
def test_int2_g64_gemv_vs_oracle(ops, dt):
    """MLX 2-bit group-64 triplets (config "quantization": {"bits": 2}; models/utils.py:96-111 forwards any bits) on native W2S units: one 16-byte
    code piece per lane, 1280 B per row pair x 2048-wide K slice = the checkpoint's own 0.3125 B per weight.  The code pairs are masked IN PLACE and
    fed to v_dot2c as bf16 numbers q * 4^j * 2^-133 (f16: shifted down, 1024 + q), so the GEMV must agree with orc_quantized_matmul_t on the 2-bit
    codes -- few and many row pairs, ragged last K slices (704, 3072), 1-3 rows, a Linear bias and a row map."""
    rng = np.random.default_rng(202)
    for N, K, M in ((96, 256, 2), (4096, 4096, 1), (130, 704, 3), (1024, 3072, 1), (64, 14336, 2)):
        w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
        x = po.round_T(rng.standard_normal((M, K)), dt)
        wq, sc, bi = po.quantize(w, 64, 2, dt)
        assert wq.shape == (N, K // 16) and sc.shape == (N, K // 64)
        codes, scales, biases = ops.quantize(to_dev(po.to_bits(w, dt), dt), bits=2)  # the HIP quantiser at 2 bits: bit-identical to the oracle's mx.quantize
        assert np.array_equal(codes.cpu().numpy().view(np.uint32), wq) and np.array_equal(to_bits(scales), sc) and np.array_equal(to_bits(biases), bi)
        lin_bias = po.round_T(rng.standard_normal(N) * 0.1, dt) if N == 130 else None
        want = po.quantized_matmul(x, wq, sc, bi, group_size=64, bits=2, dtype=dt, lin_bias=None if lin_bias is None else po.to_bits(lin_bias, dt))
        wt = ops.repack_w2s(codes, scales, biases, lin_bias=None if lin_bias is None else to_dev(po.to_bits(lin_bias, dt), dt))
        assert wt.nbytes == (N // 2) * ((K + 2047) // 2048) * 1280
        got = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), wt, group_size=64, bits=2)
        # with a Linear bias the product is rounded to T before the add: a one-ulp landing there is carried into the sum (assert_dot_close: mag)
        assert_dot_close(got.float().cpu().numpy(), want, dt, what=f"int2 gemv {N}x{K} M={M} {dt}", mag=None if lin_bias is None else want - lin_bias[None, :])
        if N == 96:
            perm = torch.from_numpy(rng.permutation(N).astype(np.int32))
            got_p = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), ops.repack_w2s(codes, scales, biases, row_map=perm))
            assert torch.equal(got_p, got[:, perm.long().cuda()])
    # every code value at every position of a word: x = one-hot rows pick single weights, which must be scale * q + bias exactly (up to T rounding)
    N, K = 64, 128
    q = rng.integers(0, 4, (N, K)).astype(np.uint32)
    wq = np.zeros((N, K // 16), np.uint32)
    for k in range(K):
        wq[:, k // 16] |= q[:, k] << np.uint32(2 * (k % 16))
    sc, bi = po.to_bits(np.full((N, K // 64), 0.5), dt), po.to_bits(np.full((N, K // 64), -0.75), dt)
    wt = ops.repack_w2s(codes_dev(wq), to_dev(sc, dt), to_dev(bi, dt))
    for k in (0, 1, 2, 15, 16, 31, 63, 64, 77, 127):
        x = np.zeros((1, K), np.float32)
        x[0, k] = 1.0
        got = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), wt).float().cpu().numpy()[0]
        assert np.array_equal(got, 0.5 * q[:, k].astype(np.float32) - 0.75), f"W2S code position {k} {dt}"


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
def test_int6_g64_gemv_vs_oracle(ops, dt):
    """MLX 6-bit group-64 triplets on native W6S units: the byte-straddling bit stream (four codes to three bytes) is split at load into a low-nibble
    plane (W4S order) and a high-two-bit plane (W2S order), 3328 B per row pair x 2048-wide K slice = the checkpoint's own 0.8125 B per weight; a
    group's dot product is the W4S dot of the low plane + 16 x the W2S dot of the high plane.  Against orc_quantized_matmul_t on the 6-bit codes."""
    rng = np.random.default_rng(606)
    for N, K, M in ((96, 256, 2), (4096, 4096, 1), (130, 704, 3), (1024, 3072, 1), (64, 14336, 2)):
        w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
        x = po.round_T(rng.standard_normal((M, K)), dt)
        wq, sc, bi = po.quantize(w, 64, 6, dt)
        assert wq.shape == (N, 3 * K // 16) and sc.shape == (N, K // 64)
        codes, scales, biases = ops.quantize(to_dev(po.to_bits(w, dt), dt), bits=6)  # the HIP quantiser's 6-bit stream: bit-identical to the oracle's mx.quantize
        assert np.array_equal(codes.cpu().numpy().view(np.uint32), wq) and np.array_equal(to_bits(scales), sc) and np.array_equal(to_bits(biases), bi)
        lin_bias = po.round_T(rng.standard_normal(N) * 0.1, dt) if N == 130 else None
        want = po.quantized_matmul(x, wq, sc, bi, group_size=64, bits=6, dtype=dt, lin_bias=None if lin_bias is None else po.to_bits(lin_bias, dt))
        wt = ops.repack_w6s(codes, scales, biases, lin_bias=None if lin_bias is None else to_dev(po.to_bits(lin_bias, dt), dt))
        assert wt.nbytes == (N // 2) * ((K + 2047) // 2048) * 3328
        got = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), wt, group_size=64, bits=6)
        assert_dot_close(got.float().cpu().numpy(), want, dt, what=f"int6 gemv {N}x{K} M={M} {dt}", mag=None if lin_bias is None else want - lin_bias[None, :])
        if N == 96:
            perm = torch.from_numpy(rng.permutation(N).astype(np.int32))
            got_p = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), ops.repack_w6s(codes, scales, biases, row_map=perm))
            assert torch.equal(got_p, got[:, perm.long().cuda()])
    # every code position of a group with every plane exercised: one-hot activations pick single weights = scale * q + bias exactly
    N, K = 64, 128
    q = rng.integers(0, 64, (N, K)).astype(np.uint64)
    q[:, :4] = np.array([63, 48, 15, 16])
    bits_row = np.zeros((N, K * 6 // 32), np.uint32)
    for k in range(K):
        pos = 6 * k
        v = q[:, k] << np.uint64(pos % 32)
        bits_row[:, pos // 32] |= (v & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        if pos % 32 > 26:
            bits_row[:, pos // 32 + 1] |= (v >> np.uint64(32)).astype(np.uint32)
    sc, bi = po.to_bits(np.full((N, K // 64), 0.25), dt), po.to_bits(np.full((N, K // 64), -2.0), dt)
    wt = ops.repack_w6s(codes_dev(bits_row), to_dev(sc, dt), to_dev(bi, dt))
    for k in (0, 1, 2, 3, 5, 15, 16, 31, 63, 64, 77, 127):
        x = np.zeros((1, K), np.float32)
        x[0, k] = 1.0
        got = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), wt).float().cpu().numpy()[0]
        assert np.array_equal(got, 0.25 * q[:, k].astype(np.float32) - 2.0), f"W6S code position {k} {dt}"


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
def test_g32_gemv_embedding_vs_oracle(ops, dt, bits):
    """MLX group-32 triplets (config "quantization": {"group_size": 32}): the W4S32 / W8S32 streaming GEMV (two scale / bias pairs per lane, one
    per 32-wide half of its codes) against orc_quantized_matmul_t with 32-wide groups -- few and many row pairs, a ragged last K slice (3072),
    1-3 rows, a Linear bias and a row map; the embedding gather is exact."""
    rng = np.random.default_rng(132 + bits)
    for N, K, M in ((96, 256, 2), (4096, 4096, 1), (130, 704, 3), (1024, 3072, 1), (64, 14336, 2)):
        w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
        x = po.round_T(rng.standard_normal((M, K)), dt)
        wq, sc, bi = po.quantize(w, 32, bits, dt)
        assert sc.shape == (N, K // 32)
        codes, scales, biases = codes_dev(wq), to_dev(sc, dt), to_dev(bi, dt)
        lin_bias = po.round_T(rng.standard_normal(N) * 0.1, dt) if N == 130 else None
        want = po.quantized_matmul(x, wq, sc, bi, group_size=32, bits=bits, dtype=dt, lin_bias=None if lin_bias is None else po.to_bits(lin_bias, dt))
        wt = ops.repack_w4s32(codes, scales, biases, lin_bias=None if lin_bias is None else to_dev(po.to_bits(lin_bias, dt), dt), bits=bits)
        assert wt.nbytes == (N // 2) * ((K + 2047) // 2048) * (2560 if bits == 4 else 4608)
        got = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), wt, group_size=32, bits=bits)
        assert_dot_close(got.float().cpu().numpy(), want, dt, what=f"int{bits} g=32 gemv {N}x{K} M={M} {dt}")
        if N == 96:
            perm = torch.from_numpy(rng.permutation(N).astype(np.int32))
            got_p = ops.quantized_matmul(to_dev(po.to_bits(x, dt), dt), ops.repack_w4s32(codes, scales, biases, row_map=perm, bits=bits))
            assert torch.equal(got_p, got[:, perm.long().cuda()])
            ids = torch.tensor([1, N - 1, 7], dtype=torch.int32, device="cuda")
            rows = ops.embedding(ids, codes, scales, biases, bits=bits, group_size=32)
            assert np.array_equal(to_bits(rows), po.to_bits(po.dequantize(wq, sc, bi, 32, bits, dt), dt)[[1, N - 1, 7]])


def test_gemv_random_shape_sweep_all_formats(ops):
    """Seeded sweep over irregular shapes for the five weight formats of the streaming GEMV (int4 / int8 in 64- and 32-wide groups, dense): N any
    even number (incl. fewer row pairs than waves and non-multiples of the wave geometry), K any multiple of 64 up to 9 slices (ragged last
    slice: the zero-padded lanes are not fetched), M 1..3, both dtypes.  Against the oracle's exact qmv / dense forms."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        dt = ("bfloat16", "float16")[case % 2]
        fmt = ("int4", "int8", "dense", "int4g32", "int8g32")[case % 5]
        N = 2 * int(rng.integers(1, 700))
        K = 64 * int(rng.integers(1, 150))
        M = int(rng.integers(1, 4))
        w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
        x = po.round_T(rng.standard_normal((M, K)), dt)
        xd = to_dev(po.to_bits(x, dt), dt)
        if fmt == "dense":
            got = ops.linear(xd, ops.repack_dense(to_dev(po.to_bits(w, dt), dt)))
            want = po.linear(x, po.to_bits(w, dt), dt)
        else:
            bits, group = (4 if fmt.startswith("int4") else 8), (32 if fmt.endswith("g32") else 64)
            wq, sc, bi = po.quantize(w, group, bits, dt)
            if group == 32:
                packed = ops.repack_w4s32(codes_dev(wq), to_dev(sc, dt), to_dev(bi, dt), bits=bits)
            else:
                packed = (ops.repack_w4s if bits == 4 else ops.repack_w8s)(codes_dev(wq), to_dev(sc, dt), to_dev(bi, dt))
            got = ops.quantized_matmul(xd, packed)
            want = po.quantized_matmul(x, wq, sc, bi, group_size=group, bits=bits, dtype=dt)
        assert_dot_close(got.float().cpu().numpy(), want, dt, max_frac=0.03, what=f"case {case}: {fmt} N={N} K={K} M={M} {dt}")


def test_sdpa_decode_random_sweep(ops):
    """Seeded sweep over the decode attention: every GQA ratio 1..8, head_dim 64 / 128, T from 1 to 3000 (1, 4, 16 and 32
    splits), capacity above T with poisoned rows past T, both dtypes."""
    rng = np.random.default_rng(77)
    for case in range(18):
        dt = ("bfloat16", "float16")[case % 2]
        rep = int(rng.integers(1, 9))
        Hkv = int(rng.integers(1, 5))
        D = (64, 128)[case % 3 == 0]
        T = int(rng.choice([1, 2, 31, 33, 127, 129, 500, 1025, 2100, 3000]))
        cap = ((T + 255) // 256) * 256 + 256 * int(rng.integers(0, 2))
        Hq = rep * Hkv
        q = po.round_T(rng.standard_normal((Hq, 1, D)), dt)
        k = po.round_T(rng.standard_normal((Hkv, cap, D)), dt)
        v = po.round_T(rng.standard_normal((Hkv, cap, D)), dt)
        k[:, T:] = 1e4
        v[:, T:] = -1e4
        want = po.sdpa(q, k, v, D ** -0.5, None, dt, True, T=T)
        got = ops.scaled_dot_product_attention(to_dev(po.to_bits(q, dt), dt).view(1, Hq, 1, D), to_dev(po.to_bits(k, dt), dt).view(1, Hkv, cap, D),
                                               to_dev(po.to_bits(v, dt), dt).view(1, Hkv, cap, D), D ** -0.5, T=T)
        assert_bits_close(to_bits(got), po.to_bits(want, dt), max_ulp=2 if dt == "bfloat16" else 4, max_frac=0.05,
                          what=f"case {case}: sdpa {Hq}/{Hkv} D{D} T{T} cap{cap} {dt}")


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("Hq,Hkv,D,L,off,cap", [(8, 2, 128, 70, 0, 256), (6, 2, 128, 33, 100, 256), (8, 1, 64, 45, 19, 256), (4, 4, 64, 1, 200, 256)])
def test_sdpa_prefill_vs_oracle(ops, dt, Hq, Hkv, D, L, off, cap):
    """mx.fast.scaled_dot_product_attention with the causal mask at L > 1 (op level, pie_sdpa_prefill): MFMA flash kernel vs
    the oracle's masked fp32 softmax; stale rows past offset + L are poisoned."""
    rng = np.random.default_rng(Hq * L + off)
    T = off + L
    q = po.round_T(rng.standard_normal((Hq, L, D)), dt)
    k = po.round_T(rng.standard_normal((Hkv, cap, D)), dt)
    v = po.round_T(rng.standard_normal((Hkv, cap, D)), dt)
    k[:, T:] = 1e4
    want = po.sdpa(q, k, v, D ** -0.5, po.causal_mask(L, off, dt), dt, True, T=T)
    got = ops.scaled_dot_product_attention(to_dev(po.to_bits(q, dt), dt).view(1, Hq, L, D), to_dev(po.to_bits(k, dt), dt).view(1, Hkv, cap, D),
                                           to_dev(po.to_bits(v, dt), dt).view(1, Hkv, cap, D), D ** -0.5, mask="causal", T=T)
    assert_dot_close(got.float().cpu().numpy(), want, dt, max_frac=0.05, what=f"sdpa prefill {Hq}/{Hkv} D{D} L{L} off{off} {dt}")


def test_qgemv_partial_fp32_and_row_parallel_sum(ops):
    """pie_qgemv_w4g64_f32 (the un-rounded fp32 row sums of a Linear): two K-halves of an int4 weight, summed and rounded once,
    reproduce the unsharded product -- the row-parallel identity the tensor-parallel step relies on -- and each partial is
    within fp32 summation error of the float64 reference."""
    rng = np.random.default_rng(9)
    N, K, dt = 512, 2048, DT
    w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
    x = po.round_T(rng.standard_normal((1, K)), dt)
    wq, sc, bi = po.quantize(w, 64, 4, dt)
    full = po.quantized_matmul(x, wq, sc, bi, dtype=dt)
    q = ((wq[:, :, None] >> (4 * np.arange(8, dtype=np.uint32))) & 15).reshape(N, K).astype(np.float64)
    deq = np.repeat(po.from_bits(sc, dt).astype(np.float64), 64, axis=1) * q + np.repeat(po.from_bits(bi, dt).astype(np.float64), 64, axis=1)  # exact affine values
    xd = to_dev(po.to_bits(x, dt), dt)
    total = None
    for k0, k1 in ((0, K // 2), (K // 2, K)):
        wsh = ops.repack_w4s(codes_dev(wq[:, k0 // 8:k1 // 8].copy()), to_dev(sc[:, k0 // 64:k1 // 64].copy(), dt), to_dev(bi[:, k0 // 64:k1 // 64].copy(), dt))
        part = ops.quantized_matmul_partial(xd[:, k0:k1].contiguous(), wsh)
        assert part.dtype == torch.float32 and part.shape == (1, N)
        ref = x[:, k0:k1].astype(np.float64) @ deq[:, k0:k1].astype(np.float64).T
        assert np.abs(part.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-6
        total = part if total is None else total + part
    got = total.to(torch.bfloat16)
    assert_dot_close(got.float().cpu().numpy(), full, dt, what="sum of row-parallel partials")


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K", [(6, 64, 64), (32, 96, 4096), (17, 4096, 1408), (1, 32, 192), (9, 6144, 4096), (31, 128, 14336)])
def test_few_row_int4_gemm_vs_oracle_qmm(ops, dt, M, N, K):
    """pie_qgemm_w4m (W4M tiles, in-register dequantisation, MFMA) against the oracle's many-row regime of
    mx.quantized_matmul (weights dequantised to T first); K = 1408 / 192: ragged last W4S slice, fewer groups than waves."""
    rng = np.random.default_rng(M * 7 + N)
    w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
    wq, sc, bi = po.quantize(w, 64, 4, dt)
    x = po.round_T(rng.standard_normal((M, K)), dt)
    packed = ops.repack_w4s(codes_dev(wq), to_dev(sc, dt), to_dev(bi, dt))
    got = ops.quantized_matmul_rows(to_dev(po.to_bits(x, dt), dt), packed)
    want = po.quantized_matmul(x, wq, sc, bi, group_size=64, bits=4, dtype=dt, regime="qmm")
    assert_dot_close(got.float().cpu().numpy(), want, dt, max_frac=0.03, what=f"w4m {M}x{N}x{K} {dt}")
    # the tile copy holds exactly the W4S matrix: multiplying the identity recovers the dequantised weights bit for bit
    if K <= 192:
        eye = np.zeros((min(32, K), K), np.float32)
        eye[np.arange(min(32, K)), np.arange(min(32, K))] = 1.0
        cols = ops.quantized_matmul_rows(to_dev(po.to_bits(eye, dt), dt), packed)          # [32, N]: row j = column j of W
        deq = po.dequantize(wq, sc, bi, dtype=dt)
        assert np.array_equal(to_bits(cols), po.to_bits(deq[:, :min(32, K)].T.copy(), dt))


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K", [(5, 16384 + 96, 1024), (32, 16512, 2048), (19, 16416, 576), (6, 96, 256), (31, 4128, 704), (64, 6144, 4096), (65, 1408, 1408),
                                   (128, 4096, 14336), (150, 512, 320), (160, 28672, 512), (192, 288, 4096), (255, 160, 128), (256, 4096, 4096)])
def test_weight_streaming_int4_gemm_vs_oracle_qmm_and_round2_kernels(ops, knobs, dt, M, N, K):
    """pie_qgemm_w4m up to 256 rows = k_w4r_gemm (w4r_gemm.hpp): every row-block geometry, column counts that are not a multiple of the 128-column
    workgroup tile (a ragged last workgroup), K / 64 that is not a multiple of the K-phases (704 = 11 groups, 320 = 5: the partial last step),
    K-split shapes reduced inside the call -- against the oracle's many-row regime of mx.quantized_matmul (weights dequantised to T first) and
    against round 2's kernels (knob w4r = 0), which differ only in the order of fp32 additions."""
    rng = np.random.default_rng(M + N)
    w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
    wq, sc, bi = po.quantize(w, 64, 4, dt)
    x = po.round_T(rng.standard_normal((M, K)), dt)
    packed = ops.repack_w4s(codes_dev(wq), to_dev(sc, dt), to_dev(bi, dt))
    xd = to_dev(po.to_bits(x, dt), dt)
    got = ops.quantized_matmul_rows(xd, packed)
    want = po.quantized_matmul(x, wq, sc, bi, group_size=64, bits=4, dtype=dt, regime="qmm")
    assert_dot_close(got.float().cpu().numpy(), want, dt, max_frac=0.03, what=f"w4r {M}x{N}x{K}")
    knobs("w4r", 0)
    ref = ops.quantized_matmul_rows(xd, packed)
    assert_dot_close(got.float().cpu().numpy(), ref.float().cpu().numpy().astype(np.float64), dt, max_frac=0.03, what=f"w4r vs round-2 kernels {M}x{N}x{K}")


def test_weight_streaming_int4_gemm_scales_beyond_the_fast_conversions_domain(ops):
    """k_w4r_gemm converts codes with v_fma_mix_f32 on f16-denormal operands, exact while |scale| * 2^24 is finite; a matrix with a scale of
    2^101 is flagged when its tiles are built and converted with plain instructions instead: same results as the oracle either way."""
    dt = "bfloat16"
    rng = np.random.default_rng(7)
    N, K, M = 256, 512, 40
    w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
    wq, sc, bi = po.quantize(w, 64, 4, dt)
    sc = sc.copy()
    bi = bi.copy()
    sc[3, 2] = 0x7200                          # bf16 bits of 2^101: one absurd group -- its row's outputs are huge but finite, every other row is untouched
    bi[3, 2] = 0xF300                          # -2^103
    x = po.round_T(rng.standard_normal((M, K)) * 2.0 ** -20, dt)
    packed = ops.repack_w4s(codes_dev(wq), to_dev(sc, dt), to_dev(bi, dt))
    got = ops.quantized_matmul_rows(to_dev(po.to_bits(x, dt), dt), packed).float().cpu().numpy()
    want = po.quantized_matmul(x, wq, sc, bi, group_size=64, bits=4, dtype=dt, regime="qmm")
    assert np.isfinite(got).all()
    assert_dot_close(got, want, dt, max_frac=0.03, what="wide-scale matrix")


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K", [(33, 64, 64), (64, 256, 512), (100, 96, 4096), (256, 4096, 1408), (300, 6144, 4096), (513, 1024, 14336), (1000, 288, 192),
                                   (512, 2048, 4096), (1300, 8192, 1024), (257, 96, 256)])
def test_many_row_int4_gemm_vs_oracle_qmm(ops, dt, M, N, K):
    """pie_qgemm_w4m beyond 32 rows: the 256-row x 256-column MFMA tile kernel (k_w4l_gemm) that processes prompts -- partial row
    tiles (33, 100, 300, 513, 1000), column counts that leave waves of the last workgroup idle (64, 96, 288), a ragged W4S slice
    (K = 1408, 192), and from 129 rows with K / 64 a multiple of 4 the one-wave-per-SIMD form k_w4l2_gemm (LDS-DMA staged x, AGPR
    accumulators; with and without a K split, ragged last row tile, idle strips) -- against the oracle's many-row regime of mx.quantized_matmul (weights dequantised to T first)."""
    rng = np.random.default_rng(M * 11 + N)
    w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
    wq, sc, bi = po.quantize(w, 64, 4, dt)
    x = po.round_T(rng.standard_normal((M, K)), dt)
    packed = ops.repack_w4s(codes_dev(wq), to_dev(sc, dt), to_dev(bi, dt))
    got = ops.quantized_matmul_rows(to_dev(po.to_bits(x, dt), dt), packed)
    want = po.quantized_matmul(x, wq, sc, bi, group_size=64, bits=4, dtype=dt, regime="qmm")
    assert_dot_close(got.float().cpu().numpy(), want, dt, max_frac=0.03, what=f"w4l {M}x{N}x{K} {dt}")
    if K <= 192:  # the identity product recovers the dequantised matrix bit for bit (rows beyond K are zero rows of x)
        r = min(M, K)
        eye = np.zeros((M, K), np.float32)
        eye[np.arange(r), np.arange(r)] = 1.0
        cols = ops.quantized_matmul_rows(to_dev(po.to_bits(eye, dt), dt), packed)
        deq = po.dequantize(wq, sc, bi, dtype=dt)
        assert np.array_equal(to_bits(cols)[:r], po.to_bits(deq.T[:r].copy(), dt))
        assert not to_bits(cols)[r:].any()


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,bias", [(1, 32, 64, False), (33, 1280, 1176, True), (64, 96, 256, False), (130, 3840, 1280, False), (300, 1280, 3420, True),
                                        (513, 6840, 1280, False), (1000, 3584, 5120, True), (257, 100, 72, True)])
def test_linear_rows_at_the_vision_tower_shapes_vs_oracle(ops, dt, M, N, K, bias):
    """pie_linear_w16m (the hand-written 16-bit MFMA GEMM behind every dense many-row Linear; weights tiled on the spot here): nn.Linear with 16-bit weights on a block of rows at the vision tower's
    shapes (PatchEmbed K = 1176, qkv 3840 x 1280, the MLP's K = 3420 whose rows are not 16-byte aligned, gate|up 6840 columns, the merger
    5120 -> 3584) and ragged N / K / M, with and without bias -- against the oracle's Linear (T x T products, fp32 accumulation, one
    rounding, then the bias)."""
    rng = np.random.default_rng(M + N + K)
    w = po.round_T(rng.standard_normal((N, K)) * 0.05, dt)
    x = po.round_T(rng.standard_normal((M, K)), dt)
    b = po.round_T(rng.standard_normal(N) * 0.5, dt) if bias else None
    wd, xd = to_dev(po.to_bits(w, dt), dt), to_dev(po.to_bits(x, dt), dt)
    got = ops.linear_rows(xd, wd)
    want = po.linear(x, po.to_bits(w, dt), dtype=dt)
    assert_dot_close(got.float().cpu().numpy(), want, dt, max_frac=0.03, what=f"linear {M}x{N}x{K} {dt}")
    if bias:
        with_b = ops.linear_rows(xd, wd, to_dev(po.to_bits(b, dt), dt))
        want_b = po.linear(x, po.to_bits(w, dt), dtype=dt, lin_bias=po.to_bits(b, dt))
        assert_dot_close(with_b.float().cpu().numpy(), want_b, dt, max_frac=0.03, mag=np.abs(want_b).max(), what=f"linear + bias {M}x{N}x{K} {dt}")