"""120 files of 4000 x 3000 (3-5 MB each; every third with restart intervals) through encode_files at its defaults against Pillow in
the workers: same vectors, and the rates (development aid)."""
import os, sys, shutil, tempfile, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import clipmi

def main():
    pool = clipmi.pipeline.DecodePool(16)
    from PIL import Image
    rng = np.random.default_rng(5)
    d = tempfile.mkdtemp(prefix="clipmi_p12_")
    try:
        yy, xx = np.mgrid[0:3000, 0:4000]
        paths = []
        for i in range(40):
            a = np.clip(np.stack([127 + 100 * np.sin(xx / (9.0 + i) + yy / 17.0), 127 + 100 * np.cos(xx / 13.0 - yy / (7.0 + i)), (xx * 3 + yy * 2 + 5 * i) % 256], -1)
                        + rng.normal(0, 10, (3000, 4000, 3)), 0, 255).astype(np.uint8)
            p = os.path.join(d, f"p{i:03d}.jpg")
            kw = dict(restart_marker_rows=1) if i % 3 == 0 else {}
            Image.fromarray(a).save(p, quality=90, subsampling=(2, 1, 0)[i % 3], **kw)
            paths.append(p)
        print("file MB", [os.path.getsize(p) >> 20 for p in paths[:3]], flush=True)
        paths = paths * 3
        model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device="cuda:0")
        res = {}
        for name, kb in (("device", None), ("pillow", 0)):
            list(clipmi.pipeline.encode_files(model, paths[:40], batch=870, pool=pool, device_jpeg_kb=kb))
            st = {}; t0 = time.perf_counter()
            res[name] = list(clipmi.pipeline.encode_files(model, paths, batch=870, pool=pool, device_jpeg_kb=kb, stats=st))
            dt = time.perf_counter() - t0
            print(f"{name}: {len(paths) / dt:.0f} images/s, device-decoded {st.get('jpeg_files', 0)} of {len(paths)}; stage s: {st}", flush=True)
        same = all(np.array_equal(a[1], b[1]) and a[0] == b[0] and a[2] == b[2] for a, b in zip(res["device"], res["pillow"]))
        print("same vectors:", same)
    finally:
        shutil.rmtree(d, ignore_errors=True); pool.close()

if __name__ == "__main__":
    main()
