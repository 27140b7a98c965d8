"""The Pillow part of the CLIP transform, and a worker PROCESS that runs it (host image pipeline).

`load_uint8` is what `transform(image)` does before its float tail (build-index.py:47-48): open, resize the shorter side
to n_px (bicubic), centre crop, RGB. It needs numpy and Pillow only, so this file can also run as a plain script

    python decode_worker.py

that serves decode requests over its stdin / stdout (see DecodePool in pipeline.py). Why processes: Pillow releases the
GIL inside the JPEG decoder, but for small images the Python around it dominates — eight decode THREADS gave 902 images/s
on 224 x 224 JPEGs where one gave 895; eight worker processes scale with the cores. The script is started by path, never
imported through the package, so a worker never imports torch or touches the GPU.

Protocol: see serve(). A file that does not decode is answered b"0" and reported per file like build-index.py:55-58.
"""
import sys

import numpy as np


def load_uint8(path, n_px, out=None):
    """Pillow part of the upstream transform: resize shorter side to n_px (bicubic), centre crop,
    RGB; returns uint8 [3, n_px, n_px] (written into `out` when given: the worker's slot of the shared segment, one
    strided copy instead of two). Identical pixels to `make_transform` before its float tail."""
    from PIL import Image
    img = Image.open(path)
    w, h = img.size
    if not (w <= h and w == n_px) and not (h <= w and h == n_px):
        if w <= h:
            nw, nh = n_px, int(n_px * h / w)
        else:
            nh, nw = n_px, int(n_px * w / h)
        img = img.resize((nw, nh), Image.BICUBIC)
        w, h = nw, nh
    left = int(round((w - n_px) / 2.0))
    top = int(round((h - n_px) / 2.0))
    img = img.crop((left, top, left + n_px, top + n_px)).convert("RGB")
    chw = np.asarray(img, dtype=np.uint8).transpose(2, 0, 1)
    if out is None:
        return np.ascontiguousarray(chw)
    np.copyto(out, chw)
    return out


# ---- the resize on the device: integer coefficient tables of Pillow's bicubic resampling -------------------------------
RESIZE_PREC = 22          # Pillow: PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5              # Pillow's bicubic_filter
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1,
                    np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def coeffs_window(in_size, out_size, o0, n_out):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (the published algorithm, same float64 operations in the same
    order) for outputs o0 .. o0 + n_out - 1 of an axis resampled from in_size to out_size.
    -> (first tap int32 [n_out], tap count int32 [n_out], coefficients int32 [n_out][ksize])"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(o0, o0 + n_out, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size)
    cnt = xmax - xmin
    k = np.arange(ksize, dtype=np.float64)[None, :]
    w = np.where(k < cnt[:, None], _bicubic((k + xmin[:, None] - center[:, None] + 0.5) * ss), 0.0)
    ww = np.cumsum(w, axis=1)[:, -1]                          # a sequential sum, as the C loop's
    w = np.where(ww[:, None] != 0.0, w / ww[:, None], w)
    kk = np.trunc(np.where(w < 0, -0.5 + w * (1 << RESIZE_PREC), 0.5 + w * (1 << RESIZE_PREC))).astype(np.int64)
    kk = np.where(k < cnt[:, None], kk, 0)
    return xmin.astype(np.int32), cnt.astype(np.int32), kk.astype(np.int32)


def resize_plan(w, h, n_px):
    """What clipmi_resize_crop_rgb8 needs to turn a w x h RGB image into the transform's n_px x n_px pixels: the sizes
    torchvision's Resize(n_px) + CenterCrop(n_px) arrive at (as load_uint8), the source rows the crop window needs, and the
    coefficient blocks ([n_px] first tap | [n_px] count | [n_px][k] coefficients) of the resampled axes."""
    if (w <= h and w == n_px) or (h <= w and h == n_px):
        nw, nh = w, h
    elif w <= h:
        nw, nh = n_px, int(n_px * h / w)
    else:
        nh, nw = n_px, int(n_px * w / h)
    left = int(round((nw - n_px) / 2.0))
    top = int(round((nh - n_px) / 2.0))
    plan = {"nw": nw, "nh": nh, "left": left, "top": top, "need_h": int(nw != w), "need_v": int(nh != h), "hk": 0, "vk": 0,
            "hcoef": np.zeros(0, np.int32), "vcoef": np.zeros(0, np.int32), "r0": top, "nrows": n_px}
    if plan["need_v"]:
        ymin, ycnt, kv = coeffs_window(h, nh, top, n_px)
        plan["r0"] = int(ymin.min())
        plan["nrows"] = int((ymin + ycnt).max()) - plan["r0"]
        plan["vk"] = kv.shape[1]
        plan["vcoef"] = np.concatenate([ymin, ycnt, kv.reshape(-1)])
    if plan["need_h"]:
        xmin, xcnt, kh = coeffs_window(w, nw, left, n_px)
        plan["hk"] = kh.shape[1]
        plan["hcoef"] = np.concatenate([xmin, xcnt, kh.reshape(-1)])
    return plan


PLAN_INTS = 16            # header of a full-size image's region: w h r0 nrows need_h need_v left top hk vk n_hcoef n_vcoef


def decode_full(path, n_px, region):
    """For the resize on the device: decode an RGB image at full size into `region` (a uint8 view of the parent's big
    segment) as [pixels HWC | pad to 16 | PLAN_INTS int32 | horizontal coefficient block | vertical block] and return
    (w, h, bytes used); None when the image should take the host path instead (not 8-bit RGB - the transform resamples in
    the image's own mode and converts afterwards -, nothing to resample, or it does not fit the region)."""
    from PIL import Image
    img = Image.open(path)
    w, h = img.size
    if img.mode != "RGB" or (w <= h and w == n_px) or (h <= w and h == n_px):
        return None
    npix = w * h * 3
    if npix + 4096 > region.size:
        return None
    plan = resize_plan(w, h, n_px)
    o_hdr = (npix + 15) // 16 * 16
    total = o_hdr + 4 * (PLAN_INTS + plan["hcoef"].size + plan["vcoef"].size)
    if total > region.size:
        return None
    region[:npix] = np.asarray(img, dtype=np.uint8).reshape(-1)
    ints = np.frombuffer(region, dtype=np.int32, count=(total - o_hdr) // 4, offset=o_hdr)
    ints[:PLAN_INTS] = [w, h, plan["r0"], plan["nrows"], plan["need_h"], plan["need_v"], plan["left"], plan["top"],
                        plan["hk"], plan["vk"], plan["hcoef"].size, plan["vcoef"].size, 0, 0, 0, 0]
    ints[PLAN_INTS:PLAN_INTS + plan["hcoef"].size] = plan["hcoef"]
    ints[PLAN_INTS + plan["hcoef"].size:] = plan["vcoef"]
    return w, h, total


JPEG_HDR_INTS = 32        # header of a JPEG region (stage_jpeg): 3 w h ncomp hs vs stream_bytes blocks | r0 nrows need_h need_v left top hk vk
                          # n_hcoef n_vcoef | stream offset, coefficient offset (bytes from the region's start) | restart interval,
                          # number of intervals, offset of their uint32 byte offsets into the segment | 1: the segment keeps its stuffing
JPEG_QUANT_OFF = 128      # 3 x 64 quantisation steps, natural order
JPEG_TABLES_OFF = 320     # six raw Huffman tables (jpeg_parse.TABLE_BYTES each): DC, AC per component
JPEG_COEF_OFF = 2048      # the resize plan's coefficient blocks (int32), then the entropy-coded segment (16-byte aligned)
_plans = {}


def stage_jpeg(path, n_px, region):
    """For the decode on the device (csrc/jpeg.hip): read the file, walk its markers (jpeg_parse.parse) and lay out in `region`
    [header | quantisation steps | Huffman tables | resize plan coefficients | entropy-coded segment without byte stuffing, followed
    by >= 16 zero bytes]. -> (w, h, bytes used); raises jpeg_parse.Unsupported for files Pillow has to decode, returns None when
    the file does not fit the region."""
    try:
        from . import jpeg_parse
    except ImportError:
        import jpeg_parse
    with open(path, "rb") as f:
        data = f.read()
    p = jpeg_parse.parse(data, keep_stuffing=True)     # (a plain slice where the file allows: the device removes the byte stuffing)
    key = (p.width, p.height, n_px)
    plan = _plans.get(key)
    if plan is None:
        if len(_plans) > 256:
            _plans.clear()
        plan = _plans[key] = resize_plan(p.width, p.height, n_px)
    nh, nv = plan["hcoef"].size, plan["vcoef"].size
    o_stream = (JPEG_COEF_OFF + 4 * (nh + nv) + 15) // 16 * 16
    o_int = (o_stream + len(p.stream) + 16 + 15) // 16 * 16
    n_int = len(p.starts) if p.ri else 0
    total = (o_int + 4 * n_int + 15) // 16 * 16
    if total > region.size:
        return p.width, p.height, -total
    ints = np.frombuffer(region, dtype=np.int32, count=JPEG_HDR_INTS)
    ints[:] = [3, p.width, p.height, p.ncomp, p.hs, p.vs, len(p.stream), p.blocks(), plan["r0"], plan["nrows"], plan["need_h"],
               plan["need_v"], plan["left"], plan["top"], plan["hk"], plan["vk"], nh, nv, o_stream, JPEG_COEF_OFF, p.ri, n_int, o_int, p.stuffed] + [0] * 8
    region[JPEG_QUANT_OFF:JPEG_QUANT_OFF + 192] = p.quant.reshape(-1)
    region[JPEG_TABLES_OFF:JPEG_TABLES_OFF + 6 * jpeg_parse.TABLE_BYTES] = np.frombuffer(b"".join(p.tables), np.uint8)
    if nh + nv:
        co = np.frombuffer(region, dtype=np.int32, count=nh + nv, offset=JPEG_COEF_OFF)
        co[:nh] = plan["hcoef"]
        co[nh:] = plan["vcoef"]
    region[o_stream:o_stream + len(p.stream)] = np.frombuffer(p.stream, np.uint8)
    region[o_stream + len(p.stream):total] = 0
    if n_int:
        np.frombuffer(region, dtype=np.uint32, count=n_int, offset=o_int)[:] = p.starts
    return p.width, p.height, total


def serve(fin, fout):
    """Answer requests until stdin closes. Request line (tab separated):
         n_px | small segment or - | byte offset of the slot | big segment or - | byte offset of the region | its size |
         what the region may take (1 full-size pixels, 2 a parsed JPEG file, 3 both) | path as hex (file names may contain
         newlines and tabs)
       Reply, 17 bytes when a segment was named (status + <iiq, zero where unused): b"0" failed | b"1" the transform's n_px x n_px
              pixels are in the slot (no segment named: b"1" + the pixels) |
              b"2" + <iiq (w, h, bytes)>: the image sits at full size, with its resize plan, in the region (decode_full) |
              b"3" + <iiq (w, h, bytes)>: a baseline JPEG file, parsed, with its resize plan, in the region (stage_jpeg) |
              b"5" + <iiq (w, h, bytes)>: as b"1", and the file would have been a b"3" with a region of that many bytes."""
    import mmap
    import os
    import struct
    segments = {}                                              # name -> mmap of /dev/shm/<name> (the parent owns it)

    def mapped(name):
        seg = segments.get(name)
        if seg is None:
            if len(segments) > 8:                              # the parent rotates a few segments; drop stale mappings
                _close_all(segments)
            # plain mmap of the POSIX segment's file: multiprocessing.shared_memory would start a resource-tracker
            # process per worker and try to unlink the parent's segment at exit (Python < 3.13)
            fd = os.open("/dev/shm/" + name.lstrip("/"), os.O_RDWR)
            try:
                seg = segments[name] = mmap.mmap(fd, 0)
            finally:
                os.close(fd)
        return seg

    while True:
        line = fin.readline()
        if not line:
            break
        try:
            n_px_s, shm_name, off_s, big_name, big_off_s, big_cap_s, mode_s, path = line.rstrip(b"\n").split(b"\t", 7)
            n_px, off, mode = int(n_px_s), int(off_s), int(mode_s)
            fname = bytes.fromhex(path.decode("ascii")).decode("utf-8", "surrogateescape")      # hex: see DecodePool._run
            reply = wanted = None
            if big_name != b"-":
                region = np.frombuffer(mapped(big_name.decode()), dtype=np.uint8, count=int(big_cap_s), offset=int(big_off_s))
                full = None
                if mode & 2:
                    try:
                        full = stage_jpeg(fname, n_px, region)
                    except Exception:                          # not a file for the device decoder (or unreadable: Pillow reports it)
                        full = None
                    if full is not None and full[2] < 0:
                        full, wanted = None, (full[0], full[1], -full[2])
                    if full is not None:
                        reply = b"3" + struct.pack("<iiq", *full)
                if full is None and mode & 1:
                    full = decode_full(fname, n_px, region)
                    if full is not None:
                        reply = b"2" + struct.pack("<iiq", *full)
                region = None
            if reply is None:
                if shm_name == b"-":
                    reply = b"1" + load_uint8(fname, n_px).tobytes()
                else:
                    slot = np.frombuffer(mapped(shm_name.decode()), dtype=np.uint8, count=3 * n_px * n_px, offset=off)
                    load_uint8(fname, n_px, out=slot.reshape(3, n_px, n_px))
                    slot = None                                # no view may outlive the request (close() refuses then)
                    reply = b"1" if wanted is None else b"5" + struct.pack("<iiq", *wanted)
        except KeyboardInterrupt:
            break
        except Exception:
            reply = b"0"
        if len(reply) == 1:
            reply += b"\0" * 16
        fout.write(reply)
        fout.flush()
    _close_all(segments)


def _close_all(segments):
    for s in segments.values():
        try:
            s.close()
        except BufferError:
            pass
    segments.clear()


if __name__ == "__main__":
    import signal
    signal.signal(signal.SIGINT, signal.SIG_IGN)      # Ctrl-C is the parent's business; a worker leaves at EOF on stdin
    try:
        serve(sys.stdin.buffer, sys.stdout.buffer)
    except (BrokenPipeError, KeyboardInterrupt):
        pass
