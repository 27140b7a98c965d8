"""Differential test of the JPEG decode on the device against Pillow on many random files (sizes 1..400, qualities 1..100, every
sampling, optimised tables, restart intervals, grey; smooth / noise / flat / sparse content), both with the byte stuffing removed on
the host and on the device: tools/jpeg_fuzz.py [n_files] [seed] [largest side, default 400]"""
import io, sys
import numpy as np, torch
from PIL import Image
sys.path.insert(0, ".")
import clipmi
from clipmi import jpeg, jpeg_parse

MAXSIDE = 400


def make(rng):
    h, w = int(rng.integers(1, MAXSIDE)), int(rng.integers(1, MAXSIDE))
    kind = rng.integers(0, 5)
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.clip(np.stack([127 + 100 * np.sin(xx / rng.uniform(3, 40) + yy / rng.uniform(3, 40)), 127 + 100 * np.cos(xx / 13.0 - yy / 7.0), (xx * 3 + yy * 2) % 256], -1)
                    + rng.normal(0, rng.uniform(0, 30), (h, w, 3)), 0, 255).astype(np.uint8)
    elif kind == 2:
        a = np.full((h, w, 3), rng.integers(0, 256, 3), dtype=np.uint8)
    elif kind == 3:
        a = np.zeros((h, w, 3), np.uint8); a[rng.integers(0, h, 20) % h, rng.integers(0, w, 20) % w] = 255
    else:
        a = (rng.integers(0, 2, (h, w, 1)) * 255).astype(np.uint8).repeat(3, axis=2)
    kw = dict(quality=int(rng.integers(1, 101)))
    grey = rng.random() < 0.15
    if not grey:
        kw["subsampling"] = int(rng.integers(0, 3))
    if rng.random() < 0.3:
        kw["optimize"] = True
    r = rng.random()
    if r < 0.2:
        kw["restart_marker_blocks"] = int(rng.integers(1, 40))
    elif r < 0.3:
        kw["restart_marker_rows"] = int(rng.integers(1, 4))
    buf = io.BytesIO()
    try:
        Image.fromarray(a[..., 0] if grey else a).save(buf, format="JPEG", **kw)
    except OSError:                                        # (Pillow's encoder refuses some combinations)
        return make(rng)
    return buf.getvalue(), (h, w, kw, int(kind), grey)

def main():
    global MAXSIDE
    if len(sys.argv) > 3:
        MAXSIDE = int(sys.argv[3])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    bad = unsup = done = 0
    while done < n:
        files = [make(rng) for _ in range(min(500 if MAXSIDE <= 400 else 100, n - done))]
        blobs = [f[0] for f in files]
        for keep in (False, True):
            got = jpeg.decode_files(blobs, dev, keep_stuffing=keep)
            for (b, info), g in zip(files, got):
                if g is None:
                    try:
                        jpeg_parse.parse(b)
                        bad += 1; print("REPORTED CORRUPT", info, keep)
                    except jpeg_parse.Unsupported as e:
                        unsup += keep; 
                    continue
                ref = np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
                if not np.array_equal(ref, g):
                    bad += 1; print("MISMATCH", info, keep, int((ref != g).sum()))
        done += len(files)
    print(f"files {done}, not for the device decoder {unsup}, bad {bad}")

if __name__ == "__main__":
    main()
