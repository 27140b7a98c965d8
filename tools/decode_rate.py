"""Decode-only throughput of pipeline.DecodePool over worker counts (development aid; no GPU involved)."""
import sys, os, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
from PIL import Image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3480
d = tempfile.mkdtemp(); rng = np.random.default_rng(0)
for i in range(n):
    Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)).save(os.path.join(d, f"i{i:05d}.jpg"), quality=95)
paths = sorted(os.path.join(d, f) for f in os.listdir(d))
print("cpus:", len(os.sched_getaffinity(0)))
for w in (4, 8, 12, 16, 24):
    with clipmi.pipeline.DecodePool(w) as pool:
        pool.decode(paths[:435], 224)
        t0 = time.perf_counter()
        for lo in range(0, n, 435): pool.decode(paths[lo:lo + 435], 224)
        dt = time.perf_counter() - t0
    print(f"{w} processes: {n / dt:.0f} images/s decode only", flush=True)
