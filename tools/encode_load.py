"""Sustained encode load for N seconds (development aid for tools/gpu_clocks_under_load.sh)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
x = torch.randint(0, 256, (870, 3, 224, 224), device=dev, dtype=torch.uint8)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
for _ in range(3): model.encode_image(x, normalize=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20): model.encode_image(x, normalize=True)
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
print(f"{n} steps in {dt:.2f} s: {870 * n / dt:.0f} img/s", flush=True)
