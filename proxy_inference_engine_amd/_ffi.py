"""ctypes binding of libpie_hip.so (include/pie_hip.h).

The product path has no CPU fallback: if the library is missing or the device is not a gfx950 the
import of anything that computes fails loudly here.  PyTorch is only the allocator / stream provider.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
PIE_BF16, PIE_F16 = 1, 2
PIE_STEP_LOGITS, PIE_STEP_GRAPH = 1, 2
PIE_OPT_KV_I8 = 2
# test / tuning switches (include/pie_hip.h: pie_set_knob); -1 restores a default
KNOBS = {"prefill_min": 0, "prefill_chunk": 1, "prefill_resident": 2, "small_m": 3, "w4l_slabs": 4, "prefill_attn_valu": 5,
         "prefill_qt": 6, "attn_merge_max_cap": 7, "attn_warm_max_mb": 8, "w4r": 9, "fuse_attn": 10}
PIE_I8 = 3  # KV page storage: int8 rows + per-head fp16 scales
KERNELS = {"embed": 0, "qkv": 1, "attn": 2, "o_proj": 3, "gate_up": 4, "down": 5, "lm_head": 6, "tail": 7}

EXPORTS = [
    "pie_hello", "pie_version", "pie_last_error", "pie_device_info", "pie_set_knob", "pie_get_knob",
    "pie_quantize_w4g64", "pie_dequantize_w4g64", "pie_w4s_bytes", "pie_repack_w4g64", "pie_w8s_bytes", "pie_repack_w8g64", "pie_qgemv_w8g64", "pie_quantize_g64", "pie_dequantize_g64", "pie_embedding_g64", "pie_w4s32_bytes", "pie_repack_w4g32", "pie_qgemv_w4g32", "pie_w8s32_bytes", "pie_repack_w8g32", "pie_qgemv_w8g32", "pie_embedding_g32",
    "pie_w2s_bytes", "pie_repack_w2g64", "pie_qgemv_w2g64", "pie_w6s_bytes", "pie_repack_w6g64", "pie_qgemv_w6g64",
    "pie_w16s_bytes", "pie_repack_dense", "pie_gemv_dense", "pie_embedding_dense", "pie_qgemv_w4g64", "pie_qgemv_w4g64_f32",
    "pie_embedding_w4g64", "pie_rms_norm", "pie_rope", "pie_rope_ex", "pie_sdpa_decode_workspace_bytes", "pie_sdpa_decode", "pie_sdpa_prefill",
    "pie_silu_mul", "pie_add", "pie_logprobs_argmax", "pie_stream_read", "pie_decoder_graph_launches", "pie_qkv_row_map", "pie_gateup_row_map",
    "pie_decoder_create", "pie_decoder_destroy", "pie_decoder_set_layer", "pie_decoder_set_globals",
    "pie_decoder_set_kv", "pie_decoder_set_paged_kv", "pie_decoder_step_batch", "pie_decoder_prefill_batch", "pie_decoder_step_mixed", "pie_decoder_set_state", "pie_decoder_step", "pie_decoder_prefill", "pie_decoder_prefill_embeds",
    "pie_decoder_bind_outputs", "pie_decoder_set_token_from", "pie_decoder_step_bytes",
    "pie_decoder_launch_kernel", "pie_decoder_kernel_bytes", "pie_decoder_configure", "pie_decoder_status",
    "pie_page_pool_slab_bytes", "pie_page_pool_create", "pie_page_pool_destroy", "pie_page_pool_size", "pie_page_pool_num_free",
    "pie_page_alloc", "pie_page_free", "pie_page_add_ref", "pie_page_ref_count", "pie_page_num_tokens", "pie_page_set_num_tokens",
    "pie_page_ptrs", "pie_paged_attn_workspace_bytes", "pie_paged_attn_decode", "pie_paged_kv_append",
    "pie_page_i8_bytes", "pie_page_scale_ptrs", "pie_page_i8_set_scales", "pie_paged_kv_append_i8", "pie_paged_attn_decode_i8",
    "pie_w16m_bytes", "pie_repack_w16m", "pie_linear_w16m_workspace", "pie_linear_w16m", "pie_gelu", "pie_vision_qkv_rope", "pie_sdpa_segments", "pie_bias_silu_mul", "pie_add_bias", "pie_add_bias_rms_norm",
    "pie_w4m_bytes", "pie_repack_w4s_to_w4m", "pie_qgemm_w4m",
    "pie_comm_create", "pie_comm_rccl_unique_id", "pie_comm_create_rccl", "pie_comm_export", "pie_comm_connect", "pie_allreduce_f32", "pie_comm_status", "pie_comm_destroy", "pie_decoder_set_comm", "pie_sample", "pie_sample_workspace_bytes",
]


class pie_decoder_config(C.Structure):
    _fields_ = [("dtype", C.c_int), ("hidden", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int),
                ("n_kv_heads", C.c_int), ("head_dim", C.c_int), ("inter", C.c_int), ("vocab", C.c_int),
                ("rms_eps", C.c_float), ("tie_word_embeddings", C.c_int), ("kv_splits", C.c_int), ("weight_format", C.c_int), ("rope_traditional", C.c_int),
                ("tp_rank", C.c_int), ("tp_world", C.c_int)]


class pie_layer_weights(C.Structure):
    _fields_ = [("attn_norm", C.c_void_p), ("mlp_norm", C.c_void_p), ("wqkv", C.c_void_p), ("wo", C.c_void_p),
                ("wgateup", C.c_void_p), ("wdown", C.c_void_p),
                ("bqkv", C.c_void_p), ("bo", C.c_void_p), ("bgateup", C.c_void_p), ("bdown", C.c_void_p),
                ("fmt_qkv", C.c_int), ("fmt_o", C.c_int), ("fmt_gateup", C.c_int), ("fmt_down", C.c_int)]


class pie_global_weights(C.Structure):
    _fields_ = [("embed_codes", C.c_void_p), ("embed_scales", C.c_void_p), ("embed_biases", C.c_void_p),
                ("final_norm", C.c_void_p), ("lm_head", C.c_void_p), ("rope_freqs", C.c_void_p),
                ("fmt_embed", C.c_int), ("fmt_lm_head", C.c_int)]


def set_knob(name: str, value: int | None) -> None:
    """One of the library's test / tuning switches (KNOBS); None restores its default."""
    check(load().pie_set_knob(KNOBS[name], -1 if value is None else int(value)))


def lib_path() -> Path:
    env = os.environ.get("PIE_HIP_LIB")
    return Path(env) if env else _PKG / "lib" / "libpie_hip.so"


_lib = None


def load() -> C.CDLL:
    """Loads libpie_hip.so; raises RuntimeError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: build it with `python -m proxy_inference_engine_amd.build` "
            "(the MI355X path has no CPU or PyTorch fallback)")
    lib = C.CDLL(str(path))
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError(f"{path} does not export {name}")
    for name in ("pie_hello", "pie_version", "pie_last_error"):
        getattr(lib, name).restype = C.c_char_p
    for name in ("pie_w4s_bytes", "pie_w16s_bytes", "pie_w8s_bytes", "pie_w4s32_bytes", "pie_w8s32_bytes", "pie_w2s_bytes", "pie_w6s_bytes", "pie_sdpa_decode_workspace_bytes", "pie_decoder_step_bytes", "pie_decoder_kernel_bytes"):
        getattr(lib, name).restype = C.c_size_t
    lib.pie_sdpa_decode.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pie_sdpa_prefill.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    lib.pie_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.pie_silu_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.pie_add.argtypes = lib.pie_silu_mul.argtypes
    lib.pie_decoder_step_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pie_decoder_kernel_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pie_decoder_configure.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pie_decoder_graph_launches.argtypes = [C.c_void_p, C.c_int]
    lib.pie_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.pie_decoder_status.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    lib.pie_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pie_sample_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.pie_sample_workspace_bytes.restype = C.c_size_t
    lib.pie_comm_create.argtypes = [C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.pie_comm_rccl_unique_id.argtypes = [C.c_void_p]
    lib.pie_comm_create_rccl.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.pie_comm_export.argtypes = [C.c_void_p, C.c_void_p]
    lib.pie_comm_connect.argtypes = [C.c_void_p, C.c_void_p]
    lib.pie_allreduce_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.pie_comm_status.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    lib.pie_comm_destroy.argtypes = [C.c_void_p]
    lib.pie_decoder_set_comm.argtypes = [C.c_void_p, C.c_void_p]
    for name in ("pie_page_pool_slab_bytes", "pie_page_pool_size", "pie_page_pool_num_free"):
        getattr(lib, name).restype = C.c_size_t
    lib.pie_page_pool_slab_bytes.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_int]
    lib.pie_page_pool_create.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.pie_page_pool_destroy.argtypes = lib.pie_page_pool_size.argtypes = lib.pie_page_pool_num_free.argtypes = [C.c_void_p]
    lib.pie_page_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.pie_page_free.argtypes = lib.pie_page_add_ref.argtypes = [C.c_void_p, C.c_uint32]
    lib.pie_page_ref_count.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.pie_page_num_tokens.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t)]
    lib.pie_page_set_num_tokens.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t]
    lib.pie_paged_attn_workspace_bytes.restype = C.c_size_t
    lib.pie_paged_attn_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 4 + [
        C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pie_paged_kv_append.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    lib.pie_page_i8_bytes.restype = C.c_size_t
    lib.pie_page_i8_bytes.argtypes = [C.c_int, C.c_int]
    lib.pie_page_scale_ptrs.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.pie_page_i8_set_scales.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pie_paged_attn_decode_i8.argtypes = lib.pie_paged_attn_decode.argtypes
    lib.pie_paged_kv_append_i8.argtypes = lib.pie_paged_kv_append.argtypes
    lib.pie_decoder_set_paged_kv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
    lib.pie_w16m_bytes.restype = C.c_size_t
    lib.pie_w16m_bytes.argtypes = [C.c_int, C.c_int]
    lib.pie_repack_w16m.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.pie_linear_w16m_workspace.restype = C.c_size_t
    lib.pie_linear_w16m_workspace.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.pie_linear_w16m.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.pie_gelu.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.pie_vision_qkv_rope.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p] * 4
    lib.pie_bias_silu_mul.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 2
    lib.pie_add_bias_rms_norm.argtypes = [C.c_void_p] * 4 + [C.c_float] + [C.c_int] * 3 + [C.c_void_p] * 3
    lib.pie_add_bias.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 2
    lib.pie_sdpa_segments.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    lib.pie_w4m_bytes.restype = C.c_size_t
    lib.pie_repack_w4s_to_w4m.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.pie_qgemm_w4m.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]
    lib.pie_decoder_step_batch.argtypes = [C.c_void_p] * 4 + [C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
    lib.pie_decoder_prefill_batch.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    lib.pie_decoder_step_mixed.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p]
    lib.pie_page_ptrs.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = load().pie_last_error().decode("utf-8", "replace")
    if rc in (-1, -2, -3):  # PIE_E_ARG / SHAPE / ALIGN: the caller's fault
        raise ValueError(f"pie_hip: {msg} (code {rc})")
    if rc == -7:  # PIE_E_RANGE: std::out_of_range in the reference
        raise IndexError(f"pie_hip: {msg} (code {rc})")
    raise RuntimeError(f"pie_hip: {msg} (code {rc})")


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("proxy_inference_engine_amd needs an MI355X (gfx950); no HIP device is visible")
    return torch.device("cuda", torch.cuda.current_device())


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return PIE_BF16
    if dt == torch.float16:
        return PIE_F16
    raise ValueError(f"activation dtype must be bfloat16 or float16, got {dt}")


def p(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(None) if t is None else C.c_void_p(t.data_ptr())


def stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def hello() -> str:
    """pie_core.hello() of the reference (src/pie_core/src/bindings.cpp:8)."""
    return load().pie_hello().decode("utf-8")
