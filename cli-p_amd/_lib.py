"""ctypes binding of libclipmi.so (the C ABI declared in include/clipmi.h).

There is NO fallback: if the HIP library is missing or a call fails, this raises. The product
path never routes through oracle/ or a CPU implementation.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CLIPMI_DEV_LIB=1 (tools/, two child-process tests): the development build, which alone reads the CLIPMI_* A/B knobs and
# holds the laboratory kernels (cli-p_amd/build.py --dev). Everything else loads the product library.
DEV_LIB = os.environ.get("CLIPMI_DEV_LIB", "") not in ("", "0")
LIB_PATH = os.path.join(HERE, "libclipmi_dev.so" if DEV_LIB else "libclipmi.so")

ABI_VERSION = 6
F32, BF16, U8 = 0, 1, 2

# every symbol include/clipmi.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "clipmi_encode_image_workspace_bytes", "clipmi_encode_image",
    "clipmi_encode_text_workspace_bytes", "clipmi_encode_text",
    "clipmi_topk_ip_workspace_bytes", "clipmi_topk_ip",
    "clipmi_topk_ip_coarse_workspace_bytes", "clipmi_topk_ip_coarse", "clipmi_dbg_topk_coarse_scan_ms",
    "clipmi_rows_stats", "clipmi_rows_absmax", "clipmi_rows_order_workspace_bytes", "clipmi_rows_order_by_absmax", "clipmi_rows_to_bf16", "clipmi_i8_copy_bytes", "clipmi_i8_meta_bytes", "clipmi_quantize_rows_i8", "clipmi_topk_ip_coarse_i8", "clipmi_dbg_topk_coarse_i8_scan_ms",
    "clipmi_dbg_quantize_rows_fp8", "clipmi_dbg_gemm_fp8",
    "clipmi_merge_topk_workspace_bytes", "clipmi_merge_topk", "clipmi_merge_topk_packed",
    "clipmi_l2_normalize_rows", "clipmi_resize_crop_rgb8", "clipmi_jpeg_workspace_bytes", "clipmi_jpeg_decode_rgb8", "clipmi_last_error", "clipmi_abi_version",
    "clipmi_dbg_gemm_bf16", "clipmi_dbg_layernorm", "clipmi_dbg_attention", "clipmi_dbg_topk_scan_ms",
    "clipmi_dbg_encode_image_probe_ms", "clipmi_dbg_encode_image_probe3_ms",
    "clipmi_dbg_split_stats", "clipmi_dbg_gemm_ln", "clipmi_dbg_gemm_resid_ln", "clipmi_dbg_gemm_resid_ln_leaf", "clipmi_dbg_gemm_ln_leaf", "clipmi_dbg_quantize_rows_fp8mx", "clipmi_dbg_gemm_fp8_bsa",
]


class Tower(C.Structure):
    """Mirror of `struct clipmi_tower` (include/clipmi.h) — keep field order identical."""
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "kind", "width", "layers", "heads", "mlp", "embed", "tokens",
        "patch", "res", "patch_k", "vocab")] + [(n, C.c_uint64) for n in (
        "blob_bytes",
        "off_patch_w", "off_cls", "off_pos", "off_ln_pre_w", "off_ln_pre_b",
        "off_tok_emb",
        "off_layers", "layer_stride",
        "lo_ln1_w", "lo_ln1_b", "lo_qkv_w", "lo_qkv_b", "lo_out_w", "lo_out_b",
        "lo_ln2_w", "lo_ln2_b", "lo_fc_w", "lo_fc_b", "lo_proj_w", "lo_proj_b",
        "off_ln_post_w", "off_ln_post_b", "off_out_proj")] + [
        ("weight_format", C.c_int32), ("ln_fold", C.c_int32)] + [(n, C.c_uint64) for n in (
        "lo_qkv_s", "lo_out_s", "lo_fc_s", "lo_proj_s", "lo_qkv_colsum", "lo_qkv_cb", "lo_fc_colsum", "lo_fc_cb")]


class ClipmiError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libclipmi.so (once). Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ClipmiError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    from . import build as _build
    if os.path.isdir(_build.CSRC) and not _build.is_current(_build.DEV if DEV_LIB else _build.PRODUCT):
        raise ClipmiError(
            f"{LIB_PATH} does not match the sources under csrc/ (content stamp mismatch): rebuild "
            "(python cli-p_amd/build.py). A stale HIP library is never loaded silently.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    TP = C.POINTER(Tower)
    L.clipmi_last_error.restype = C.c_char_p
    L.clipmi_abi_version.restype = i32
    L.clipmi_encode_image_workspace_bytes.restype = sz
    L.clipmi_encode_image_workspace_bytes.argtypes = [TP, i32]
    L.clipmi_encode_image.restype = i32
    L.clipmi_encode_image.argtypes = [TP, vp, vp, i32, i32, vp, i32, vp, sz, vp]
    L.clipmi_encode_text_workspace_bytes.restype = sz
    L.clipmi_encode_text_workspace_bytes.argtypes = [TP, i32]
    L.clipmi_encode_text.restype = i32
    L.clipmi_encode_text.argtypes = [TP, vp, vp, i32, vp, i32, vp, sz, vp]
    L.clipmi_topk_ip_workspace_bytes.restype = sz
    L.clipmi_topk_ip_workspace_bytes.argtypes = [i64, i32, i32, i32]
    L.clipmi_topk_ip.restype = i32
    L.clipmi_topk_ip.argtypes = [vp, i32, i64, i32, vp, i32, i32, i64, vp, vp, vp, sz, vp]
    L.clipmi_topk_ip_coarse_workspace_bytes.restype = sz
    L.clipmi_topk_ip_coarse_workspace_bytes.argtypes = [i64, i32, i32, i32]
    L.clipmi_topk_ip_coarse.restype = i32
    L.clipmi_topk_ip_coarse.argtypes = [vp, vp, i64, i32, C.c_float, vp, i32, i32, i64, vp, vp, vp, sz, vp]
    L.clipmi_dbg_quantize_rows_fp8.restype = i32
    L.clipmi_dbg_quantize_rows_fp8.argtypes = [vp, vp, vp, i32, i32, vp]
    L.clipmi_dbg_gemm_fp8.restype = i32
    L.clipmi_dbg_gemm_fp8.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.clipmi_rows_stats.restype = i32
    L.clipmi_rows_stats.argtypes = [vp, i64, i32, vp, vp, vp]
    L.clipmi_rows_to_bf16.restype = i32
    L.clipmi_rows_to_bf16.argtypes = [vp, i64, i32, vp, vp]
    L.clipmi_i8_copy_bytes.restype = sz
    L.clipmi_i8_copy_bytes.argtypes = [i64, i32]
    L.clipmi_i8_meta_bytes.restype = sz
    L.clipmi_i8_meta_bytes.argtypes = [i64]
    L.clipmi_quantize_rows_i8.restype = i32
    L.clipmi_quantize_rows_i8.argtypes = [vp, i64, i32, vp, vp, sz, vp, sz, vp]
    L.clipmi_rows_absmax.restype = i32
    L.clipmi_rows_absmax.argtypes = [vp, i64, i32, vp, vp]
    L.clipmi_rows_order_workspace_bytes.restype = sz
    L.clipmi_rows_order_workspace_bytes.argtypes = [i64]
    L.clipmi_rows_order_by_absmax.restype = i32
    L.clipmi_rows_order_by_absmax.argtypes = [vp, i64, i32, vp, vp, sz, vp]
    L.clipmi_topk_ip_coarse_i8.restype = i32
    L.clipmi_topk_ip_coarse_i8.argtypes = [vp, vp, vp, C.c_float, i64, i32, C.c_float, vp, i32, i32, i64, vp, vp, vp, sz, vp]
    L.clipmi_dbg_topk_coarse_i8_scan_ms.restype = i32
    L.clipmi_dbg_topk_coarse_i8_scan_ms.argtypes = [vp, vp, vp, C.c_float, i64, i32, C.c_float, vp, i32, i32, vp, vp, vp, sz, vp,
                                                    i32, C.POINTER(C.c_float), C.POINTER(C.c_longlong)]
    L.clipmi_dbg_topk_coarse_scan_ms.restype = i32
    L.clipmi_dbg_topk_coarse_scan_ms.argtypes = [vp, vp, i64, i32, C.c_float, vp, i32, i32, vp, vp, vp, sz, vp, i32,
                                                 C.POINTER(C.c_float), C.POINTER(C.c_longlong)]
    L.clipmi_merge_topk_workspace_bytes.restype = sz
    L.clipmi_merge_topk_workspace_bytes.argtypes = [i32, i32, i32]
    L.clipmi_merge_topk.restype = i32
    L.clipmi_merge_topk.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, sz, vp]
    L.clipmi_merge_topk_packed.restype = i32
    L.clipmi_merge_topk_packed.argtypes = [vp, sz, i32, i32, i32, vp, vp, vp]
    L.clipmi_l2_normalize_rows.restype = i32
    L.clipmi_l2_normalize_rows.argtypes = [vp, i64, i32, vp]
    L.clipmi_resize_crop_rgb8.restype = i32
    L.clipmi_resize_crop_rgb8.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, vp]
    L.clipmi_jpeg_workspace_bytes.restype = i64
    L.clipmi_jpeg_workspace_bytes.argtypes = [i64, i32]
    L.clipmi_jpeg_decode_rgb8.restype = i32
    L.clipmi_jpeg_decode_rgb8.argtypes = [vp, vp, i32, vp, i32, i64, i64, i64, vp, vp, vp, i64, vp]
    L.clipmi_dbg_gemm_bf16.restype = i32
    L.clipmi_dbg_gemm_bf16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.clipmi_dbg_split_stats.restype = i32
    L.clipmi_dbg_split_stats.argtypes = [vp, i32, vp, vp, i32, i32, vp]
    L.clipmi_dbg_gemm_ln.restype = i32
    L.clipmi_dbg_gemm_ln.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.clipmi_dbg_quantize_rows_fp8mx.restype = i32
    L.clipmi_dbg_quantize_rows_fp8mx.argtypes = [vp, vp, vp, i32, i32, vp]
    L.clipmi_dbg_gemm_fp8_bsa.restype = i32
    L.clipmi_dbg_gemm_fp8_bsa.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.clipmi_dbg_gemm_resid_ln_leaf.restype = i32
    L.clipmi_dbg_gemm_resid_ln_leaf.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.clipmi_dbg_gemm_ln_leaf.restype = i32
    L.clipmi_dbg_gemm_ln_leaf.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.clipmi_dbg_gemm_resid_ln.restype = i32
    L.clipmi_dbg_gemm_resid_ln.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.clipmi_dbg_layernorm.restype = i32
    L.clipmi_dbg_layernorm.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.clipmi_dbg_attention.restype = i32
    L.clipmi_dbg_attention.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.clipmi_dbg_topk_scan_ms.restype = i32
    L.clipmi_dbg_topk_scan_ms.argtypes = [vp, i64, i32, vp, i32, i32, vp, vp, vp, sz, vp, i32, C.POINTER(C.c_float)]
    L.clipmi_dbg_encode_image_probe_ms.restype = i32
    L.clipmi_dbg_encode_image_probe_ms.argtypes = [TP, vp, vp, i32, i32, vp, vp, sz, vp, i32, i32,
                                                   C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.clipmi_dbg_encode_image_probe3_ms.restype = i32
    L.clipmi_dbg_encode_image_probe3_ms.argtypes = [TP, vp, vp, i32, i32, vp, vp, sz, vp, i32, i32,
                                                    C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                    C.POINTER(C.c_int)]
    if L.clipmi_abi_version() != ABI_VERSION:
        raise ClipmiError(f"libclipmi.so ABI {L.clipmi_abi_version()} != binding {ABI_VERSION}: rebuild")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().clipmi_last_error().decode("utf-8", "replace")
        raise ClipmiError(f"{what} failed (code {rc}): {msg}")


def last_error():
    return lib().clipmi_last_error().decode("utf-8", "replace")


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream, so the library enqueues where torch does."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_SIDE_STREAMS = {}


_COPY_STREAMS = {}


def copy_stream(device):
    """THE stream for host -> device copies that run beside the kernels (pipeline.encode_files): one per device for the whole
    process, for the same reason as side_stream - a stream per call used up the hardware queues."""
    import torch
    key = torch.device(device).index
    s = _COPY_STREAMS.get(key)
    if s is None:
        s = _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


def side_stream(device):
    """THE internal HIP stream that work enqueued on torch's current stream of `device` may overlap with (CLIP.encode_image's
    second kernel sequence, IndexFlatIP's pipelined passes): ONE per (device, caller's stream) for the whole process, shared by
    every model and index - this ROCm gives only a process's first three streams a hardware queue of their own and maps every
    later one onto a shared fourth, where "two in flight" run back to back."""
    import torch
    cur = torch.cuda.current_stream(device)
    key = (torch.device(device).index, cur.cuda_stream)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return cur, s
