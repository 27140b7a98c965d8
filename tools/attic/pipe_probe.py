"""Which part of the end-to-end loop slows every other batch's decode? (development aid)
usage: pipe_probe.py <level>   0 decode only | 1 + copy to pinned | 2 + H2D | 3 + encode | 4 + d2h of the result"""
import sys, os, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
from PIL import Image
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 4350
d = tempfile.mkdtemp(); rng = np.random.default_rng(0)
for i in range(n):
    Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)).save(os.path.join(d, f"i{i:05d}.jpg"), quality=95)
paths = sorted(os.path.join(d, f) for f in os.listdir(d))
pool = clipmi.pipeline.DecodePool(16)
import torch
model = None
if level >= 2:
    dev = torch.device("cuda:0")
    if level >= 3:
        model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
pinned = torch.empty((435, 3, 224, 224), dtype=torch.uint8)
if level >= 2: pinned = pinned.pin_memory()
pool.decode(paths[:435], 224, copy=False)
ts = []
t_all = time.perf_counter()
for lo in range(0, n, 435):
    t0 = time.perf_counter()
    (view, good), ok, bad = pool.decode(paths[lo:lo + 435], 224, copy=False)
    t1 = time.perf_counter()
    if level >= 1:
        if os.environ.get("PROBE_TORCH_COPY"): pinned[:len(ok)].copy_(torch.from_numpy(view))
        else: np.copyto(pinned.numpy()[:len(ok)], view)
    if level >= 2: devt = pinned[:len(ok)].to(dev, non_blocking=True)
    if level >= 3: f = model.encode_image(devt, normalize=True)
    if level >= 4: f = f.cpu().numpy()
    ts.append(f"{(t1 - t0) * 1e3:.0f}")
if level >= 2: torch.cuda.synchronize()
dt = time.perf_counter() - t_all
print(f"level {level}: {n / dt:.0f} images/s; decode ms per batch: {' '.join(ts)}", flush=True)
pool.close()
