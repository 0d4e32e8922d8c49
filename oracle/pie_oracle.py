"""ctypes/numpy front-end of the CPU oracle (oracle/pie_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED -- see the header of pie_oracle.c.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package never does.

Besides thin wrappers over the C ops this file restates the reference's HOST logic for the path
(paths relative to /root/reference/src/proxy_inference_engine/):
  - OracleKVCache      <- cache/kv_cache/reusable.py:8-254
  - OraclePromptCache  <- cache/prompt_cache.py:13-76
  - generate_step      <- engine/inference_engine.py:228-297
  - make_sampler (greedy) <- samplers/__init__.py:37-38
Arrays are numpy float32 holding values representable in the activation dtype; parameters are kept
as "bits" arrays (uint16 for bfloat16/float16, float32 for float32).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libpie_oracle.so"

DTYPES = {"float32": 0, "bfloat16": 1, "float16": 2}


def build(force: bool = False) -> Path:
    src = _HERE / "pie_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _LIB_PATH


class orc_linear_t(C.Structure):
    _fields_ = [("w", C.c_void_p), ("scales", C.c_void_p), ("biases", C.c_void_p), ("lin_bias", C.c_void_p)]


class orc_llama_t(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("hidden", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int),
        ("n_kv_heads", C.c_int), ("head_dim", C.c_int), ("inter", C.c_int), ("vocab", C.c_int),
        ("group_size", C.c_int), ("bits", C.c_int), ("quantized", C.c_int), ("tie_word_embeddings", C.c_int),
        ("eps", C.c_float),
        ("rope_freqs", C.c_void_p), ("attn_norm", C.c_void_p), ("mlp_norm", C.c_void_p),
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("gate", C.c_void_p), ("up", C.c_void_p), ("down", C.c_void_p),
        ("embed", orc_linear_t), ("final_norm", C.c_void_p), ("lm_head", orc_linear_t),
        ("rope_traditional", C.c_int),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_round.restype = C.c_float
        _lib.orc_round.argtypes = [C.c_float, C.c_int]
        _lib.orc_logprobs_argmax.restype = C.c_int
        _lib.orc_llama_forward.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _dt(dtype) -> int:
    return DTYPES[dtype] if isinstance(dtype, str) else int(dtype)


def set_threads(n: int) -> None:
    lib().orc_set_threads(C.c_int(n))


def max_threads() -> int:
    return lib().orc_max_threads()


def set_qmm_min_rows(n: int) -> None:
    """Row count from which quantised Linears use MLX's qmm form (weights dequantised to T, then a T x T -> fp32
    matmul) instead of the exact fp32 affine sum of the qmv kernels; 0 = never.  Default 6 (pie_oracle.c)."""
    lib().orc_set_qmm_min_rows(C.c_int(n))


def get_qmm_min_rows() -> int:
    return int(lib().orc_get_qmm_min_rows())


# ------------------------------------------------------------------ dtype plumbing (numpy, bit-exact with the C helpers)
def to_bits(x, dtype) -> np.ndarray:
    """float32 values -> storage bits of `dtype` (round-to-nearest-even)."""
    x = _f32(x)
    d = _dt(dtype)
    if d == 0:
        return x.copy()
    out = np.empty(x.shape, np.uint16)
    lib().orc_to_T(_p(x), _p(out), C.c_size_t(x.size), C.c_int(d))
    return out


def from_bits(b, dtype) -> np.ndarray:
    d = _dt(dtype)
    if d == 0:
        return _f32(b).copy()
    b = np.ascontiguousarray(b, dtype=np.uint16)
    out = np.empty(b.shape, np.float32)
    lib().orc_from_T(_p(b), _p(out), C.c_size_t(b.size), C.c_int(d))
    return out


def round_T(x, dtype) -> np.ndarray:
    x = _f32(x).copy()
    lib().orc_round_inplace(_p(x), C.c_size_t(x.size), C.c_int(_dt(dtype)))
    return x


# ------------------------------------------------------------------ ops
def quantize(w, group_size=64, bits=4, dtype="bfloat16"):
    """mx.quantize.  w float32 [N,K] -> (codes uint32 [N,K*bits/32], scales bits, biases bits)."""
    w = _f32(w)
    N, K = w.shape
    assert K % group_size == 0
    d = _dt(dtype)
    wq = np.empty((N, K * bits // 32), np.uint32)
    sb_t = np.float32 if d == 0 else np.uint16
    scales = np.empty((N, K // group_size), sb_t)
    biases = np.empty((N, K // group_size), sb_t)
    lib().orc_quantize(_p(w), N, K, group_size, bits, d, _p(wq), _p(scales), _p(biases))
    return wq, scales, biases


def dequantize(wq, scales, biases, group_size=64, bits=4, dtype="bfloat16"):
    N = wq.shape[0]
    K = wq.shape[1] * 32 // bits
    out = np.empty((N, K), np.float32)
    lib().orc_dequantize(_p(wq), _p(scales), _p(biases), N, K, group_size, bits, _dt(dtype), _p(out))
    return out


def quantized_matmul(x, wq, scales, biases, transpose=True, group_size=64, bits=4, dtype="bfloat16", lin_bias=None, regime="qmv"):
    """mx.quantized_matmul(x, w, scales, biases, transpose=True, ...) (Appendix A.2).  x [..., K] -> [..., N].
    regime: "qmv" = exact fp32 affine sum (few rows), "qmm" = weights dequantised to T first (MLX's matrix kernels)."""
    assert transpose, "the hot path only uses transpose=True (nn.QuantizedLinear)"
    x = _f32(x)
    K = x.shape[-1]
    N = wq.shape[0]
    assert wq.shape[1] * 32 // bits == K
    M = x.size // K
    y = np.empty((M, N), np.float32)
    fn = lib().orc_quantized_matmul_t if regime == "qmv" else lib().orc_quantized_matmul_dequant
    fn(_p(x), M, _p(wq), _p(scales), _p(biases), N, K, group_size, bits, _dt(dtype), _p(lin_bias), _p(y))
    return y.reshape(*x.shape[:-1], N)


def linear(x, w_bits, dtype="bfloat16", lin_bias=None):
    x = _f32(x)
    N, K = w_bits.shape
    M = x.size // K
    y = np.empty((M, N), np.float32)
    lib().orc_linear(_p(x), M, _p(w_bits), N, K, _dt(dtype), _p(lin_bias), _p(y))
    return y.reshape(*x.shape[:-1], N)


def rms_norm(x, w_bits, eps, dtype="bfloat16"):
    x = _f32(x)
    H = x.shape[-1]
    y = np.empty_like(x)
    lib().orc_rms_norm(_p(x), x.size // H, H, _p(w_bits), C.c_float(eps), _dt(dtype), _p(y))
    return y


def rope(x, freqs, offset=0, dtype="bfloat16", traditional=False):
    """mx.fast.rope(x[..., heads, L, D], D, traditional, base=None, scale=1.0, offset, freqs)."""
    x = _f32(x)
    L, D = x.shape[-2:]
    heads = x.size // (L * D)
    freqs = _f32(freqs)
    y = np.empty_like(x)
    lib().orc_rope_ex(_p(x), heads, L, D, _p(freqs), int(offset), _dt(dtype), int(traditional), _p(y))
    return y


def llama3_rope_freqs(D, base, max_len=8192.0, factor=1.0, low=1.0, high=1.0):
    f = np.empty(D // 2, np.float32)
    lib().orc_llama3_rope_freqs(D, C.c_float(base), C.c_float(max_len), C.c_float(max_len), C.c_float(factor),
                                C.c_float(low), C.c_float(high), _p(f))
    return f


def sdpa(q, k, v, scale, mask=None, dtype="bfloat16", fused=True, T=None):
    """q [Hq,L,D]; k,v [Hkv,cap,D] (first T rows valid, default cap); mask [L,T] additive or None."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    Hq, L, D = q.shape
    Hkv, cap, _ = k.shape
    T = cap if T is None else T
    m = None if mask is None else _f32(mask)
    out = np.empty_like(q)
    lib().orc_sdpa(_p(q), _p(k), _p(v), Hq, Hkv, L, T, cap, D, C.c_float(scale), _p(m), _dt(dtype), int(bool(fused)), _p(out))
    return out


# ------------------------------------------------------------------ int8 KV pages (src/pie_core/include/engine/page.hpp:25-32,109-117)
# The reference declares the storage -- int8 key / value blocks [64, heads, head_dim] and float16 per-head scales [heads, 1] that start
# as ones ("head-wise quant for now") -- and neither a quantiser nor a reader; the arithmetic below is the product's definition
# (csrc/paged_i8.hip), restated: parity for this piece is oracle == HIP, not oracle == reference.
def kv_i8_quantize(x, scale):
    """x [..., heads, D] (values already rounded to T), scale [..., heads] float16 -> int8: clamp(rint(x / s), -127, 127) in float32."""
    s = np.asarray(scale, np.float16).astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.rint(np.asarray(x, np.float32) / s)   # float32 division, round-half-even
    q = np.where(np.isnan(q), np.float32(0), np.clip(q, -127, 127))
    return q.astype(np.int8)


def kv_i8_dequantize(q, scale):
    """fp32(q) * fp32(s): what the attention multiplies with -- no rounding to T in between."""
    return np.asarray(q, np.int8).astype(np.float32) * np.asarray(scale, np.float16).astype(np.float32)[..., None]


def causal_mask(L, offset, dtype="bfloat16"):
    m = np.empty((L, offset + L), np.float32)
    lib().orc_causal_mask(L, offset, _dt(dtype), _p(m))
    return m


def silu_mul(a, b, dtype="bfloat16"):
    a, b = _f32(a), _f32(b)
    y = np.empty_like(a)
    lib().orc_silu_mul(_p(a), _p(b), C.c_size_t(a.size), _dt(dtype), _p(y))
    return y


def add(a, b, dtype="bfloat16"):
    a, b = _f32(a), _f32(b)
    y = np.empty_like(a)
    lib().orc_add(_p(a), _p(b), C.c_size_t(a.size), _dt(dtype), _p(y))
    return y


def logprobs_argmax(logits):
    logits = _f32(logits).reshape(-1)
    lp = np.empty_like(logits)
    tok = lib().orc_logprobs_argmax(_p(logits), logits.size, _p(lp))
    return int(tok), lp


# ------------------------------------------------------------------ the model (models/llama/language.py)
_LINEARS = ("q", "k", "v", "o", "gate", "up", "down")
_LIN_NAMES = {
    "q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj", "o": "self_attn.o_proj",
    "gate": "mlp.gate_proj", "up": "mlp.up_proj", "down": "mlp.down_proj",
}


class OracleLlama:
    """Holds an MLX-layout checkpoint (dict name -> numpy bits array) and runs orc_llama_forward.

    `config` uses the HF/MLX config.json keys ModelArgs reads (models/llama/language.py:13-29).
    """

    def __init__(self, config: dict, weights: dict, dtype: str = "bfloat16"):
        self.config = dict(config)
        self.dtype = dtype
        self.weights = {k: np.ascontiguousarray(v) for k, v in weights.items()}
        c = self.config
        self.hidden = c["hidden_size"]
        self.n_layers = c["num_hidden_layers"]
        self.n_heads = c["num_attention_heads"]
        self.n_kv_heads = c.get("num_key_value_heads") or self.n_heads
        self.head_dim = c.get("head_dim") or self.hidden // self.n_heads
        self.inter = c["intermediate_size"]
        self.vocab = c["vocab_size"]
        self.tie = bool(c.get("tie_word_embeddings", True))  # language.py:29 default True
        q = c.get("quantization") or {}
        self.group_size, self.bits = q.get("group_size", 64), q.get("bits", 4)
        self.quantized = bool(q)
        rs = c.get("rope_scaling") or {}
        max_len = float(c.get("max_position_embeddings") or 8192)
        self.freqs = llama3_rope_freqs(self.head_dim, float(c.get("rope_theta", 10000.0)), max_len,
                                       float(rs.get("factor", 1.0)), float(rs.get("low_freq_factor", 1.0)),
                                       float(rs.get("high_freq_factor", 1.0)))
        self._keep = []
        self._struct = self._build_struct()

    @property
    def layers(self):
        return list(range(self.n_layers))

    def _lin(self, prefix: str) -> orc_linear_t:
        w = self.weights
        t = orc_linear_t()
        t.w = w[f"{prefix}.weight"].ctypes.data
        has_q = f"{prefix}.scales" in w  # models/utils.py:99-109 class_predicate
        t.scales = w[f"{prefix}.scales"].ctypes.data if has_q else None
        t.biases = w[f"{prefix}.biases"].ctypes.data if has_q else None
        t.lin_bias = w[f"{prefix}.bias"].ctypes.data if f"{prefix}.bias" in w else None
        return t

    def _build_struct(self) -> orc_llama_t:
        m = orc_llama_t()
        m.dtype = DTYPES[self.dtype]
        m.hidden, m.n_layers, m.n_heads, m.n_kv_heads = self.hidden, self.n_layers, self.n_heads, self.n_kv_heads
        m.head_dim, m.inter, m.vocab = self.head_dim, self.inter, self.vocab
        m.group_size, m.bits, m.quantized, m.tie_word_embeddings = self.group_size, self.bits, int(self.quantized), int(self.tie)
        m.eps = float(self.config["rms_norm_eps"])
        m.rope_freqs = self.freqs.ctypes.data
        m.rope_traditional = int(bool(self.config.get("rope_traditional", False)))
        L = self.n_layers
        ptr_arr = C.c_void_p * L
        an = ptr_arr(*[self.weights[f"model.layers.{i}.input_layernorm.weight"].ctypes.data for i in range(L)])
        mn = ptr_arr(*[self.weights[f"model.layers.{i}.post_attention_layernorm.weight"].ctypes.data for i in range(L)])
        self._keep += [an, mn]
        m.attn_norm, m.mlp_norm = C.addressof(an), C.addressof(mn)
        for name in _LINEARS:
            arr = (orc_linear_t * L)(*[self._lin(f"model.layers.{i}.{_LIN_NAMES[name]}") for i in range(L)])
            self._keep.append(arr)
            setattr(m, name, C.addressof(arr))
        m.embed = self._lin("model.embed_tokens")
        m.final_norm = self.weights["model.norm.weight"].ctypes.data
        if not self.tie:
            m.lm_head = self._lin("lm_head")
        return m

    def forward(self, ids, cache: list["OracleKVCache"], last_only=False, sdpa_fused=True, want_hidden=False, inputs_embeds=None):
        """Model.__call__(inputs[None], cache=cache) for batch 1: returns logits [L, V] (or [V] if last_only).
        inputs_embeds [L, hidden] (ids then ignored): LanguageModel(None, inputs_embeds=...), models/intern/language.py:155-158."""
        emb = None
        if inputs_embeds is not None:
            emb = _f32(inputs_embeds).reshape(-1, self.hidden)
            ids = np.zeros(emb.shape[0], np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1)
        L = ids.size
        offset = cache[0].offset
        for c in cache:  # the capacity side of update_and_fetch (reusable.py:113-131) happens on the host
            c.ensure(L, self.n_kv_heads, self.head_dim)
        cap = cache[0].keys.shape[2]
        kp = (C.c_void_p * self.n_layers)(*[c.keys.ctypes.data for c in cache])
        vp = (C.c_void_p * self.n_layers)(*[c.values.ctypes.data for c in cache])
        logits = np.empty((self.vocab,) if last_only else (L, self.vocab), np.float32)
        hidden = np.empty((L, self.hidden), np.float32) if want_hidden else None
        rc = lib().orc_llama_forward_ex(C.byref(self._struct), _p(ids), _p(emb), L, kp, vp, cap, offset, int(last_only),
                                        int(sdpa_fused), _p(logits), _p(hidden))
        if rc != 0:
            raise RuntimeError("orc_llama_forward failed")
        for c in cache:
            c.offset += L  # reusable.py:139
        return (logits, hidden) if want_hidden else logits


# ------------------------------------------------------------------ cache/kv_cache/reusable.py
class OracleKVCache:
    """ReusableKVCache restated on numpy buffers [1, n_kv, cap, D] (values representable in T)."""

    def __init__(self, step: int = 256, growth_factor: float = 1.5, max_capacity: int | None = None):
        self.keys = None
        self.values = None
        self.offset = 0
        self.step, self.growth_factor, self.max_capacity = step, growth_factor, max_capacity

    def _round_up(self, n):
        return ((n + self.step - 1) // self.step) * self.step

    def _grow_to(self, new_capacity):
        if self.max_capacity is not None:
            new_capacity = min(new_capacity, self.max_capacity)
        B, n_kv, _, D = self.keys.shape
        nk = np.zeros((B, n_kv, new_capacity, D), np.float32)
        nv = np.zeros((B, n_kv, new_capacity, D), np.float32)
        nk[..., : self.offset, :] = self.keys[..., : self.offset, :]
        nv[..., : self.offset, :] = self.values[..., : self.offset, :]
        self.keys, self.values = nk, nv

    def reuse(self, new_prompt_length: int, common_prefix_length: int) -> None:  # reusable.py:44-94
        if self.keys is None:
            return
        self.offset = common_prefix_length
        current = self.keys.shape[2]
        if current < new_prompt_length:
            self._grow_to(self._round_up(max(int(current * self.growth_factor), new_prompt_length)))

    def ensure(self, needed: int, n_kv: int, D: int) -> None:  # reusable.py:113-131, 144-203
        if self.keys is None:
            cap = self._round_up(needed)
            if self.max_capacity is not None:
                cap = min(cap, self.max_capacity)
            self.keys = np.zeros((1, n_kv, cap, D), np.float32)
            self.values = np.zeros((1, n_kv, cap, D), np.float32)
            self.offset = 0
        elif self.offset + needed > self.keys.shape[2]:
            if self.offset % self.step != 0:  # "safety" trim, reusable.py:125-129: growth is then computed from the offset, not the old capacity
                self.keys = np.ascontiguousarray(self.keys[..., : self.offset, :])
                self.values = np.ascontiguousarray(self.values[..., : self.offset, :])
            current = self.keys.shape[2]
            self._grow_to(self._round_up(max(int(current * self.growth_factor), self.offset + needed)))

    def update_and_fetch(self, keys, values):  # reusable.py:96-142
        needed = keys.shape[2]
        self.ensure(needed, keys.shape[1], keys.shape[3])
        self.keys[..., self.offset : self.offset + needed, :] = keys
        self.values[..., self.offset : self.offset + needed, :] = values
        self.offset += needed
        return self.keys[..., : self.offset, :], self.values[..., : self.offset, :]

    @property
    def state(self):
        return self.keys, self.values

    def is_trimmable(self):
        return True

    def trim(self, n: int) -> int:  # reusable.py:235-248
        n = min(self.offset, n)
        self.offset -= n
        return n


# ------------------------------------------------------------------ cache/prompt_cache.py
class OraclePromptCache:
    def __init__(self):
        self.cache: list[OracleKVCache] = []
        self.computed_ids = np.zeros((0,), np.int64)

    def create_kv_cache(self, model) -> None:  # prompt_cache.py:34-41
        self.cache = [OracleKVCache() for _ in model.layers]

    def update(self, ids) -> None:  # prompt_cache.py:43-50
        ids = np.asarray(ids, np.int64).reshape(-1)
        self.computed_ids = ids if self.computed_ids.size == 0 else np.concatenate([self.computed_ids, ids])

    def __call__(self, prompt_ids):  # prompt_cache.py:52-76
        prompt_ids = np.asarray(prompt_ids, np.int64).reshape(-1)
        if not self.cache or self.computed_ids.size == 0:
            return prompt_ids
        common = 0
        for i, tok in enumerate(self.computed_ids):
            if i >= len(prompt_ids) - 1 or prompt_ids[i] != tok:
                break
            common += 1
        if common == 0:
            return prompt_ids
        for c in self.cache:
            c.reuse(len(prompt_ids), common)
        return prompt_ids[common:]


def generate_step(model: OracleLlama, prompt_cache: OraclePromptCache, prompt_ids, sdpa_fused=True):
    """InferenceEngine.generate_step with the greedy sampler (temp=0) and no logits processors
    (engine/inference_engine.py:228-297).  Yields (token_id, logprobs[V])."""

    def _inference(ids):
        logits = model.forward(ids, prompt_cache.cache, last_only=False, sdpa_fused=sdpa_fused)
        last = logits[-1]                       # logits[:, -1, :]            :254
        prompt_cache.update(ids)                #                              :255
        tok, logprobs = logprobs_argmax(last)   # f32 log-softmax + argmax     :268-271
        return tok, logprobs

    if len(prompt_cache.cache) == 0:
        prompt_cache.create_kv_cache(model)
    todo = prompt_cache(prompt_ids)
    tok, lp = _inference(todo)
    step = 0
    while True:
        if step > 0:
            tok, lp = _inference(np.array([tok]))
        yield tok, lp
        step += 1


# ------------------------------------------------------------------ synthetic checkpoints
def synth_checkpoint(config: dict, seed: int = 0, dtype: str = "bfloat16", lm_head_gain: float = 1.0) -> dict:
    """Random-weight checkpoint in the on-disk layout models/utils.py:51-125 consumes:
    HF names, MLX-quantised triplets {name}.weight/.scales/.biases when config["quantization"] is set
    (Linear and Embedding with K % 64 == 0, models/utils.py:99-109).  Linear W ~ N(0, 0.02^2),
    norm weights 1 + N(0, 0.02^2).  Returned arrays are storage bits (uint16 / float32 / uint32 codes)."""
    rng = np.random.default_rng(seed)
    H, I, V = config["hidden_size"], config["intermediate_size"], config["vocab_size"]
    nh = config["num_attention_heads"]
    nkv = config.get("num_key_value_heads") or nh
    D = config.get("head_dim") or H // nh
    q = config.get("quantization") or {}
    out: dict[str, np.ndarray] = {}

    def put_linear(name, N, K, gain=1.0, bias=False):
        if bias:  # nn.Linear(..., bias=True) for attention_bias / mlp_bias (language.py:42-53,117-126)
            out[f"{name}.bias"] = to_bits(round_T(rng.standard_normal(N, dtype=np.float32) * 0.1, dtype), dtype)
        w = round_T(rng.standard_normal((N, K), dtype=np.float32) * (0.02 * gain), dtype)
        if q and K % 64 == 0:
            wq, s, b = quantize(w, q["group_size"], q["bits"], dtype)
            out[f"{name}.weight"], out[f"{name}.scales"], out[f"{name}.biases"] = wq, s, b
        else:
            out[f"{name}.weight"] = to_bits(w, dtype)

    put_linear("model.embed_tokens", V, H)
    for i in range(config["num_hidden_layers"]):
        p = f"model.layers.{i}"
        out[f"{p}.input_layernorm.weight"] = to_bits(1.0 + 0.02 * rng.standard_normal(H, dtype=np.float32), dtype)
        out[f"{p}.post_attention_layernorm.weight"] = to_bits(1.0 + 0.02 * rng.standard_normal(H, dtype=np.float32), dtype)
        ab, mb = bool(config.get("attention_bias")), bool(config.get("mlp_bias"))
        put_linear(f"{p}.self_attn.q_proj", nh * D, H, bias=ab)
        put_linear(f"{p}.self_attn.k_proj", nkv * D, H, bias=ab)
        put_linear(f"{p}.self_attn.v_proj", nkv * D, H, bias=ab)
        put_linear(f"{p}.self_attn.o_proj", H, nh * D, bias=ab)
        put_linear(f"{p}.mlp.gate_proj", I, H, bias=mb)
        put_linear(f"{p}.mlp.up_proj", I, H, bias=mb)
        put_linear(f"{p}.mlp.down_proj", H, I, bias=mb)
    out["model.norm.weight"] = to_bits(1.0 + 0.02 * rng.standard_normal(H, dtype=np.float32), dtype)
    if not config.get("tie_word_embeddings", True):
        put_linear("lm_head", V, H, gain=lm_head_gain)
    return out


TINY_CONFIG = {
    "model_type": "llama", "hidden_size": 256, "num_hidden_layers": 2, "intermediate_size": 704,
    "num_attention_heads": 4, "num_key_value_heads": 2, "rms_norm_eps": 1e-5, "vocab_size": 512,
    "rope_theta": 10000.0, "max_position_embeddings": 2048, "tie_word_embeddings": False,
    "quantization": {"group_size": 64, "bits": 4},
}
