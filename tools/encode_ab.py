"""Encode step A/B of env knobs (development aid): ms per step of B = 870, bf16, min of 3 x 8 steps."""
import sys, os, time
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 870
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
x = torch.randint(0, 256, (B, 3, 224, 224), device=dev, dtype=torch.uint8)
for _ in range(4): model.encode_image(x, normalize=True)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(8): model.encode_image(x, normalize=True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 8)
print(f"{os.environ.get('CLIPMI_GEMM_ST', 'default')}: {best * 1e3:.3f} ms per step, {B / best:.0f} images/s", flush=True)
