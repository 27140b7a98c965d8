"""Host image pipeline for the build side: threaded decode + the CLIP
transform's geometry on the host, pinned staging, asynchronous H2D, encode on the GPU in batches.

Reference per-image sequence (build-index.py:47-51): Image.open -> transform -> unsqueeze(0).to(device)
-> encode_image -> /norm -> .cpu(). Here the same images travel as uint8 [B,3,R,R] (resize / centre
crop / RGB done with Pillow exactly as `transform` does; the /255, -mean, /std tail is fused into the
device patch kernel), B at a time, decode of batch i+1 overlapping the GPU work of batch i.
Failures are per file (build-index.py:55-58): a file that does not decode is reported, not fatal.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def load_uint8(path, n_px):
    """Pillow part of the upstream transform: resize shorter side to n_px (bicubic), centre crop,
    RGB; returns uint8 [3, n_px, n_px]. Identical pixels to `make_transform` before its float tail."""
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
    return np.ascontiguousarray(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1))


def _load_safe(args):
    path, n_px = args
    try:
        return load_uint8(path, n_px)
    except KeyboardInterrupt:
        raise
    except Exception:
        return None


def encode_files(model, paths, batch=256, workers=8):
    """Generator over batches: yields (ok_paths, features f32 [n,E] numpy normalised, failed_paths).
    Decode runs on `workers` threads (Pillow releases the GIL while decoding)."""
    n_px = model.visual.input_resolution
    dev = model.device
    use_gpu = dev.type == "cuda"
    copy_stream = torch.cuda.Stream(device=dev) if use_gpu else None

    def stage(chunk):
        arrs = list(pool.map(_load_safe, [(p, n_px) for p in chunk]))
        ok = [p for p, a in zip(chunk, arrs) if a is not None]
        bad = [p for p, a in zip(chunk, arrs) if a is None]
        if not ok:
            return ok, bad, None, None
        host = torch.from_numpy(np.stack([a for a in arrs if a is not None]))
        if use_gpu:
            host = host.pin_memory()
            with torch.cuda.stream(copy_stream):
                devt = host.to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return ok, bad, devt, ev
        return ok, bad, host, None

    chunks = [paths[i:i + batch] for i in range(0, len(paths), batch)]
    with ThreadPoolExecutor(max_workers=workers) as pool, ThreadPoolExecutor(max_workers=1) as stager:
        nxt = stager.submit(stage, chunks[0]) if chunks else None
        for ci in range(len(chunks)):
            ok, bad, devt, ev = nxt.result()
            nxt = stager.submit(stage, chunks[ci + 1]) if ci + 1 < len(chunks) else None
            feats = None
            if devt is not None:
                if ev is not None:
                    torch.cuda.current_stream(dev).wait_event(ev)
                feats = model.encode_image(devt, normalize=True).cpu().numpy().astype("float32")
            yield ok, feats, bad
