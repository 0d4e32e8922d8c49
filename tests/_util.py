"""Shared helpers for the parity tests."""
import numpy as np
import torch

from oracle import pie_oracle as po

TORCH_DT = {"bfloat16": torch.bfloat16, "float16": torch.float16}


def ulp_key(bits: np.ndarray) -> np.ndarray:
    """Maps 16-bit float storage bits to integers whose difference is the distance in ulps."""
    s = bits.astype(np.int32)
    return np.where(s & 0x8000, -(s & 0x7FFF), s & 0x7FFF)


def assert_bits_close(got_bits, want_bits, max_ulp=1, max_frac=0.02, what=""):
    """Results are 16-bit floats: both sides accumulate in fp32 in different orders, so an element may land on
    the neighbouring representable value.  Bound: every element within `max_ulp`, at most `max_frac` of them off."""
    got_bits, want_bits = np.asarray(got_bits).reshape(-1), np.asarray(want_bits).reshape(-1)
    assert got_bits.shape == want_bits.shape, (got_bits.shape, want_bits.shape)
    d = np.abs(ulp_key(got_bits) - ulp_key(want_bits))
    frac = float((d > 0).mean())
    assert d.max() <= max_ulp, f"{what}: max ulp distance {d.max()} > {max_ulp} (mismatch fraction {frac:.4f})"
    allowed = max(int(np.ceil(max_frac * d.size)), 1)          # tiny outputs: one neighbouring-value landing is allowed
    assert int((d > 0).sum()) <= allowed, f"{what}: {frac:.4%} of elements differ (limit {max_frac:.2%})"
    return frac


EPS = {"bfloat16": 2.0 ** -8, "float16": 2.0 ** -11}


def assert_dot_close(got, want, dtype, max_frac=0.02, what="", mag=None):
    """Outputs of long dot products: within one T-ulp of the oracle, where the ulp is taken at max(|want|, max|want| / 128)
    (an element that cancels to nearly zero has a tiny ulp of its own, but its fp32 summation-order error is set by
    the magnitude of the terms), and at most `max_frac` of the elements differ at all.  `mag`: magnitude of an
    intermediate T-rounded value the element went through (e.g. the Linear output before its bias is added): a one-ulp
    landing there is carried into the final value unchanged."""
    got, want = np.asarray(got, np.float64).reshape(-1), np.asarray(want, np.float64).reshape(-1)
    floor = np.abs(want).max() / 128.0
    ref = np.abs(want) if mag is None else np.maximum(np.abs(want), np.abs(np.asarray(mag, np.float64).reshape(-1)))
    tol = 2.0 * EPS[dtype] * np.maximum(ref, floor)
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} elements beyond one ulp, worst {np.abs(got - want)[bad].max():.6g}"
    frac = float((got != want).mean())
    assert frac <= max_frac, f"{what}: {frac:.4%} of elements differ (limit {max_frac:.2%})"
    return frac


def assert_vec_close(got, want, dtype, c_max=4.0, c_rms=4.0, what=""):
    """End-to-end tolerance for activations / logits that passed through many 16-bit rounding points.
    Every op boundary rounds to T (eps = 2^-8 bf16, 2^-11 f16) and the HIP kernels accumulate in fp32 in a
    different order than the oracle, so single-ulp landings compound through the layers (measured on MI355X,
    2 layers: max error 2.5 eps*max|ref|, rms error 1.3-2.3 eps*rms(ref), see scripts/diag_parity.py).  Bound, stated in units of one ulp of the
    largest reference element:   max|got-want| <= c_max * eps * max|want|   and   rms(got-want) <= c_rms * eps * rms(want)."""
    got, want = np.asarray(got, np.float64).reshape(-1), np.asarray(want, np.float64).reshape(-1)
    eps = EPS[dtype]
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max()
    assert err <= c_max * eps * scale, f"{what}: max abs err {err:.5f} > {c_max} * eps * {scale:.3f}"
    rms_w = max(np.sqrt(np.mean(want ** 2)), 1e-30)
    rms_e = np.sqrt(np.mean((got - want) ** 2))
    assert rms_e <= c_rms * eps * rms_w, f"{what}: rms err {rms_e:.6f} > {c_rms} * eps * rms(ref) {rms_w:.4f}"
    return err / (eps * scale)


def to_dev(bits: np.ndarray, dtype: str, device="cuda") -> torch.Tensor:
    """numpy storage bits (uint16) -> device tensor of the 16-bit float dtype."""
    t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(device)
    return t.view(TORCH_DT[dtype])


def to_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def codes_dev(wq: np.ndarray, device="cuda") -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(wq).view(np.int32)).to(device)
