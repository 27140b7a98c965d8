"""`Image.open(tfn)` for baseline JPEG files on the device (reference build-index.py:47).

The host walks the markers (and removes the 0xFF00 byte stuffing, unless it leaves that to the device too); csrc/jpeg.hip does the
rest (Huffman decode in self-synchronising subsequences, DC prediction, jpeg_idct_islow, fancy upsampling, YCbCr -> RGB) and leaves Pillow's
bytes in HBM: rows of width*3 RGB bytes per image, the layout `clipmi_resize_crop_rgb8` takes (resize.py). Files this
parser does not let through (progressive, CMYK / RGB-coded, 12-bit, odd sampling, anything that is not a
JPEG) raise `Unsupported` and stay with Pillow in the decode workers - that is a choice of decoder per file format, made
on the host from the file's own header; a file the device then reports as corrupt (status != 0) goes the same way, so
that Pillow's error handling stays the reference's.
"""
import numpy as np
import torch

from . import _lib
from .jpeg_parse import TABLE_BYTES, Parsed, Unsupported, parse  # noqa: F401

IMAGE = np.dtype([("stream_off", "<i8"), ("coef_off", "<i8"), ("out_off", "<i8"), ("intervals_off", "<i8"), ("stream_bytes", "<i4"),
                  ("width", "<i4"), ("height", "<i4"), ("ncomp", "<i4"), ("hs", "<i4"), ("vs", "<i4"), ("dc_tbl", "<i4", 3),
                  ("ac_tbl", "<i4", 3), ("restart_interval", "<i4"), ("n_intervals", "<i4"), ("stuffed", "<i4"), ("reserved", "<i4"),
                  ("quant", "u1", (3, 64))], align=True)
assert IMAGE.itemsize == 288


def pack(items):
    """Parsed records -> (IMAGE array, tables uint8 [nt][288], streams uint8, out_bytes, total_blocks, max_blocks, max_pixels)"""
    recs = np.zeros(len(items), dtype=IMAGE)
    pool = {}
    soff = coff = ooff = 0
    max_blocks = max_pixels = 1
    pieces = []
    for k, it in enumerate(items):
        r = recs[k]
        r["stream_off"], r["coef_off"], r["out_off"], r["stream_bytes"] = soff, coff, ooff, len(it.stream)
        r["width"], r["height"], r["ncomp"], r["hs"], r["vs"] = it.width, it.height, it.ncomp, it.hs, it.vs
        idx = [pool.setdefault(t, len(pool)) for t in it.tables]
        r["dc_tbl"], r["ac_tbl"] = idx[0::2], idx[1::2]
        r["quant"] = it.quant
        r["stuffed"] = it.stuffed
        pad = (-len(it.stream)) % 16 + 16
        pieces.append(it.stream)
        pieces.append(b"\0" * pad)
        soff += len(it.stream) + pad
        if it.ri:                                            # restart intervals: their byte offsets travel behind the segment
            r["restart_interval"], r["n_intervals"], r["intervals_off"] = it.ri, len(it.starts), soff
            raw = it.starts.astype("<u4").tobytes()
            raw += b"\0" * ((-len(raw)) % 16)
            pieces.append(raw)
            soff += len(raw)
        nb = it.blocks()
        coff += nb
        ooff += (it.width * it.height * 3 + 15) // 16 * 16
        max_blocks, max_pixels = max(max_blocks, nb), max(max_pixels, it.width * it.height)
    tables = np.frombuffer(b"".join(pool), np.uint8).reshape(-1, TABLE_BYTES) if pool else np.zeros((0, TABLE_BYTES), np.uint8)
    return recs, tables, np.frombuffer(b"".join(pieces), np.uint8), ooff, coff, max_blocks, max_pixels


def decode_device(items, device, stream=None):
    """Parsed records -> (out uint8 device tensor, records, status int32 device tensor): the RGB rows of image k start at
    records[k]["out_off"]. Asynchronous on torch's current stream of `device`; status is valid once that stream is."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.ClipmiError("jpeg.decode_device needs the HIP path (no CPU fallback)")
    L = _lib.lib()
    recs, tables, streams, out_bytes, total_blocks, max_blocks, max_pixels = pack(items)
    n = len(items)
    out = torch.empty(max(out_bytes, 16), dtype=torch.uint8, device=device)
    status = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    if n == 0:
        return out, recs, status
    o_tab = (recs.nbytes + 15) // 16 * 16
    o_str = (o_tab + tables.nbytes + 15) // 16 * 16
    host = torch.empty(o_str + streams.nbytes, dtype=torch.uint8).pin_memory()
    hv = host.numpy()
    hv[:recs.nbytes] = recs.view(np.uint8).reshape(-1)
    hv[o_tab:o_tab + tables.nbytes] = tables.reshape(-1)
    hv[o_str:] = streams
    dev = host.to(device, non_blocking=True)
    ws_bytes = int(L.clipmi_jpeg_workspace_bytes(total_blocks, len(tables)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    base = dev.data_ptr()
    rc = L.clipmi_jpeg_decode_rgb8(base + o_str, base, n, base + o_tab, len(tables), total_blocks, max_blocks, max_pixels,
                                   out.data_ptr(), status.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr(device))
    _lib.check(rc, "clipmi_jpeg_decode_rgb8")
    cur = torch.cuda.current_stream(device)
    dev.record_stream(cur)
    ws.record_stream(cur)
    return out, recs, status


def decode_files(blobs, device, keep_stuffing=False):
    """JPEG file contents -> list of uint8 [H,W,3] numpy arrays (None where the file is not for the device decoder or the
    device reported it corrupt). Synchronises; a convenience for tests and tools - the pipeline keeps the pixels in HBM.
    keep_stuffing: hand the segments over as they are in the file and let the device remove the byte stuffing (the pipeline's form)."""
    items, where = [], []
    for k, b in enumerate(blobs):
        try:
            items.append(parse(b, keep_stuffing=keep_stuffing))
            where.append(k)
        except Unsupported:
            pass
    res = [None] * len(blobs)
    if not items:
        return res
    out, recs, status = decode_device(items, device)
    st = status.cpu().numpy()
    host = out.cpu().numpy()
    for t, k in enumerate(where):
        if st[t] == 0:
            r = recs[t]
            h, w, o = int(r["height"]), int(r["width"]), int(r["out_off"])
            res[k] = host[o:o + h * w * 3].reshape(h, w, 3)
    return res
