"""Device JPEG decode against Pillow on seeded files + its rate on the bench's files (tools/jpeg_probe.py [n_images])."""
import io, sys, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, ".")
import clipmi
from clipmi import jpeg

rng = np.random.default_rng(1)
def smooth(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0 - yy / 7.0), (xx * 3 + yy * 2) % 256], -1)
    return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
def enc(a, **kw):
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format="JPEG", **kw); return buf.getvalue()
blobs = []
for (h, w) in [(224, 224), (37, 53), (8, 8), (17, 16), (100, 75), (64, 129), (5, 7), (480, 640), (1100, 1500)]:
    for sub in (0, 1, 2):
        for q in (95, 75, 30):
            blobs.append(enc(smooth(h, w), quality=q, subsampling=sub))
            if h * w < 100000:
                blobs.append(enc(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), quality=q, subsampling=sub))
blobs.append(enc(smooth(300, 200)[..., 0], quality=85, optimize=True))
blobs.append(enc(smooth(300, 200), quality=85, optimize=True))
dev = torch.device("cuda:0")
t0 = time.time(); got = jpeg.decode_files(blobs, dev); print("first call %.2f s" % (time.time() - t0))
bad = 0
for b, g in zip(blobs, got):
    ref = np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
    if g is None or not np.array_equal(ref, g):
        bad += 1
        print("MISMATCH", ref.shape, None if g is None else int((ref != g).sum()))
print("cases", len(blobs), "bad", bad)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 870
for name, mk in (("noise q95 (bench)", lambda: rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)), ("smooth q85", lambda: smooth(224, 224))):
    files = [enc(mk(), quality=95 if "noise" in name else 85) for _ in range(min(n, 64))]
    files = (files * (n // len(files) + 1))[:n]
    t0 = time.time(); items = [jpeg.parse(b) for b in files]; tp = time.time() - t0
    for _ in range(2):
        out, recs, status = jpeg.decode_device(items, dev); torch.cuda.synchronize()
    t0 = time.time(); reps = 5
    for _ in range(reps):
        out, recs, status = jpeg.decode_device(items, dev)
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    print(f"{name}: {n} files of ~{len(files[0]) >> 10} KB: parse {tp / n * 1e6:.0f} us/file, device decode {dt * 1e3:.2f} ms per batch = {n / dt / 1e3:.1f} k images/s, status sum {int(status.sum())}")
