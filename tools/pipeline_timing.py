"""End-to-end build-side throughput: JPEG files -> Pillow decode threads -> pinned H2D -> HIP encode (development aid)."""
import sys, os, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
from PIL import Image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = sys.argv[3] if len(sys.argv) > 3 else "procs"
W_, H_ = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (224, 224)      # e.g. 1024 768: photo-sized files
pool = clipmi.pipeline.DecodePool(workers) if mode == "procs" else None       # before anything touches the GPU
dev = torch.device("cuda:0")
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
if (W_, H_) == (224, 224):
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)).save(os.path.join(d, f"img_{i:05d}.jpg"), quality=95)
else:                                       # smooth gradients + texture: JPEGs of photo-like size (a few hundred KB)
    yy, xx = np.mgrid[0:H_, 0:W_]
    for i in range(n):
        a = np.stack([(xx * (1 + i % 3) + yy) % 256, (yy * 2 + i) % 256, (xx + yy * (1 + i % 2)) % 256], axis=-1).astype(np.uint8)
        a[::3, ::5] = rng.integers(0, 256, 3)
        Image.fromarray(a).save(os.path.join(d, f"img_{i:05d}.jpg"), quality=90)
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
paths = sorted(os.path.join(d, f) for f in os.listdir(d)) * int(os.environ.get("PIPE_REPEAT", "1"))   # files decoded again
for _ in clipmi.pipeline.encode_files(model, paths[:512], batch=256, workers=workers, pool=pool): pass
t0 = time.perf_counter(); got = 0
for ok, feats, bad in clipmi.pipeline.encode_files(model, paths, batch=435, workers=workers, pool=pool):
    got += len(ok)
dt = time.perf_counter() - t0
print(f"{got} {W_}x{H_} JPEGs, {workers} decode {'processes' if pool else 'threads'}: {got/dt:.0f} images/s end to end (host decode bound; GPU encode alone ~100 k/s)", flush=True)
if pool: pool.close()
