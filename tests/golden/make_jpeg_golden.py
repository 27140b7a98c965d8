"""Writes tests/golden/jpeg_cases.npz: small JPEG files (seeded content, every sampling the device decoder takes, odd sizes,
grey, optimised Huffman tables) and the pixels PILLOW decodes them to - `Image.open(...).convert("RGB")`, the reference's own
decode (build-index.py:47) run in this container (Pillow 12.2.0, libjpeg-turbo). Run from the repository root:
    python tests/golden/make_jpeg_golden.py
"""
import io
import os

import numpy as np
from PIL import Image


def smooth(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0 - yy / 7.0), (xx * 3 + yy * 2) % 256], -1)
    return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


def cases(rng):
    out = []
    for (h, w) in [(37, 53), (8, 8), (17, 16), (64, 129), (5, 7), (96, 80)]:
        for sub in (0, 1, 2):
            q = (95, 75, 30)[(h + sub) % 3]
            a = smooth(rng, h, w) if (h + w + sub) % 2 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            out.append((a, dict(quality=q, subsampling=sub)))
    out.append((smooth(rng, 45, 61)[..., 0], dict(quality=85, optimize=True)))
    out.append((smooth(rng, 45, 61), dict(quality=85, optimize=True)))
    return out


def main():
    rng = np.random.default_rng(5)
    files, pixels = [], []
    for a, kw in cases(rng):
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, format="JPEG", **kw)
        files.append(np.frombuffer(buf.getvalue(), np.uint8))
        pixels.append(np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))
    d = {f"file_{i}": f for i, f in enumerate(files)}
    d.update({f"rgb_{i}": p for i, p in enumerate(pixels)})
    d["n"] = np.array(len(files))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_cases.npz"), **d)
    print(len(files), "cases")


if __name__ == "__main__":
    main()
