"""build-index.py's loop end to end on JPEG files, by batch size and with / without the JPEG decode on the device
(tools/files_to_vectors.py [noise|photo] ...): bench.py's files_to_vectors leg with its knobs exposed."""
import io, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
import clipmi

def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
    pool = clipmi.pipeline.DecodePool(clipmi.indexer.default_workers())      # before the GPU is touched
    import torch
    from PIL import Image
    rng = np.random.default_rng(0)
    n = 2 * 870 if kind != "photo2k" else 435
    side = (224, 224) if kind != "photo2k" else (1500, 2000)
    d = tempfile.mkdtemp(prefix="clipmi_f2v_")
    try:
        yy, xx = np.mgrid[0:side[0], 0:side[1]]
        for i in range(n):
            if kind == "noise":
                a = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
            else:
                base = np.stack([127 + 100 * np.sin(xx / (5.0 + i % 7) + yy / 17.0), 127 + 100 * np.cos(xx / 13.0 - yy / (4.0 + i % 5)), (xx * 3 + yy * 2 + i) % 256], -1)
                a = np.clip(base + rng.normal(0, 12, side + (3,)), 0, 255).astype(np.uint8)
            Image.fromarray(a).save(os.path.join(d, f"img_{i:05d}.jpg"), quality=95 if kind == "noise" else 85)
        paths = sorted(os.path.join(d, f) for f in os.listdir(d)) * (5 if kind != "photo2k" else 10)
        print("shm free MB", pool.shm_room() >> 20, "workers", pool.n, "file KB", os.path.getsize(paths[0]) >> 10, flush=True)
        model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device="cuda:0")
        for batch in ((435, 870) if kind != "photo2k" else (435,)):
            for kb in ((0, 8192) if kind != "photo2k" else (8192,)):
                for _ in clipmi.pipeline.encode_files(model, paths[:batch], batch=batch, pool=pool, device_jpeg_kb=kb):
                    pass
                t0 = time.perf_counter(); got = 0; st = {}
                for ok, feats, bad in clipmi.pipeline.encode_files(model, paths, batch=batch, pool=pool, device_jpeg_kb=kb, stats=st):
                    got += len(ok)
                dt = time.perf_counter() - t0
                nb = len(paths) / batch
                print(f"{kind} batch {batch} device_jpeg_kb {kb}: {got / dt / 1e3:.1f} k images/s; per batch ms: wall {dt / nb * 1e3:.1f} decode {st.get('decode_s', 0) / nb * 1e3:.1f} "
                      f"copy {st.get('copy_s', 0) / nb * 1e3:.1f} encode {st.get('encode_s', 0) / nb * 1e3:.1f}; device-decoded {st.get('jpeg_files', 0)}", flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
        pool.close()

if __name__ == "__main__":
    main()
