"""encode_files over thousands of random JPEG files (tools/jpeg_fuzz.py's generator: every size / option, some too large for their
region at first, some the device decoder does not take) with the JPEG decode on the device against Pillow in the workers: the same
vectors and the same failed files (development aid; tools/attic/pipeline_jpeg_stress.py [n])."""
import os, sys, shutil, tempfile
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import clipmi
import jpeg_fuzz

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    pool = clipmi.pipeline.DecodePool(16)                      # before the GPU is touched
    rng = np.random.default_rng(3)
    d = tempfile.mkdtemp(prefix="clipmi_stress_")
    try:
        paths = []
        for i in range(n):
            blob, _ = jpeg_fuzz.make(rng)
            if i % 97 == 0:
                blob = blob[:len(blob) // 2]                   # truncated files: Pillow decides
            if i % 211 == 0:
                blob = b"not a jpeg"
            p = os.path.join(d, f"f{i:05d}.jpg")
            open(p, "wb").write(blob)
            paths.append(p)
        import warnings; warnings.simplefilter("ignore")
        model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device="cuda:0")
        st = {}
        host = list(clipmi.pipeline.encode_files(model, paths, batch=333, pool=pool, device_jpeg_kb=0))
        devj = list(clipmi.pipeline.encode_files(model, paths, batch=333, pool=pool, device_jpeg_kb=64, stats=st))
        bad = 0
        for h, g in zip(host, devj):
            if h[0] != g[0] or h[2] != g[2] or not ((h[1] is None and g[1] is None) or np.array_equal(h[1], g[1])):
                bad += 1
        print(f"files {n}, batches {len(host)}, device-decoded {st.get('jpeg_files', 0)}, failed {sum(len(h[2]) for h in host)}, batches that differ {bad}")
    finally:
        shutil.rmtree(d, ignore_errors=True); pool.close()

if __name__ == "__main__":
    main()
