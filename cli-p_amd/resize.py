"""Row a2 on the device: Pillow-exact bicubic resize + centre crop of full-size 8-bit RGB images (csrc/resize.hip).

`resize_crop_device(images, n_px, device)` takes decoded images (uint8 [H,W,3] numpy arrays of any sizes) and returns the
uint8 [B,3,n_px,n_px] device tensor `transform(image)` would produce before its float tail (build-index.py:48) - the same
bytes `load_uint8` computes with Pillow on the host. One pinned buffer carries job records, coefficient tables and pixels
to the device in one copy.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .decode_worker import resize_plan

JOB = np.dtype([("src_off", "<i8"), ("w", "<i4"), ("h", "<i4"), ("r0", "<i4"), ("nrows", "<i4"), ("out_index", "<i4"),
                ("need_h", "<i4"), ("need_v", "<i4"), ("left", "<i4"), ("top", "<i4"), ("hk", "<i4"), ("vk", "<i4"),
                ("hcoef_off", "<i8"), ("vcoef_off", "<i8"), ("tmp_off", "<i8")], align=True)
assert JOB.itemsize == 80


def pack_jobs(shapes, n_px, plans=None):
    """Job records + one coefficient array for images of the given (h, w) shapes laid out back to back (HWC bytes).
    -> (jobs structured array, coef int32 array, raw bytes total, scratch bytes, max_rows)"""
    jobs = np.zeros(len(shapes), dtype=JOB)
    coefs, coff, soff, toff, max_rows = [], 0, 0, 0, 1
    for i, (h, w) in enumerate(shapes):
        p = plans[i] if plans is not None else resize_plan(w, h, n_px)
        j = jobs[i]
        j["src_off"], j["w"], j["h"], j["r0"], j["nrows"], j["out_index"] = soff, w, h, p["r0"], p["nrows"], i
        j["need_h"], j["need_v"], j["left"], j["top"], j["hk"], j["vk"] = p["need_h"], p["need_v"], p["left"], p["top"], p["hk"], p["vk"]
        j["hcoef_off"] = coff
        coefs.append(p["hcoef"]); coff += p["hcoef"].size
        j["vcoef_off"] = coff
        coefs.append(p["vcoef"]); coff += p["vcoef"].size
        j["tmp_off"] = toff
        toff += p["nrows"] * n_px * 3
        soff += h * w * 3
        max_rows = max(max_rows, p["nrows"])
    coef = np.concatenate(coefs).astype(np.int32) if coff else np.zeros(1, np.int32)
    return jobs, coef, soff, max(toff, 1), max_rows


def resize_crop_device(images, n_px, device, out=None):
    """images: list of uint8 [H,W,3] arrays -> uint8 [len, 3, n_px, n_px] on `device` (async on torch's current stream)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.ClipmiError("resize_crop_device needs the HIP path (no CPU fallback)")
    L = _lib.lib()
    B = len(images)
    if out is None:
        out = torch.empty((B, 3, n_px, n_px), dtype=torch.uint8, device=device)
    if B == 0:
        return out
    for a in images:
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("resize_crop_device: expected uint8 [H,W,3] arrays")
    jobs, coef, raw_bytes, scratch_bytes, max_rows = pack_jobs([a.shape[:2] for a in images], n_px)
    o_coef = (jobs.nbytes + 15) // 16 * 16
    o_raw = (o_coef + coef.nbytes + 15) // 16 * 16
    host = torch.empty(o_raw + raw_bytes, dtype=torch.uint8).pin_memory()
    hv = host.numpy()
    hv[:jobs.nbytes] = jobs.view(np.uint8).reshape(-1)
    hv[o_coef:o_coef + coef.nbytes] = coef.view(np.uint8)
    off = o_raw
    for a in images:
        n = a.size
        hv[off:off + n] = np.ascontiguousarray(a).reshape(-1)
        off += n
    dev = host.to(device, non_blocking=True)
    scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=device)
    base = dev.data_ptr()
    rc = L.clipmi_resize_crop_rgb8(base + o_raw, base, B, max_rows, base + o_coef, n_px, out.data_ptr(), scratch.data_ptr(),
                                   _lib.stream_ptr(device))
    _lib.check(rc, "clipmi_resize_crop_rgb8")
    dev.record_stream(torch.cuda.current_stream(device))
    return out
