"""encode images/s over batch sizes (development aid); CLIPMI_GEMM_SPLIT=0/1 A/B of the whole-rounds + remainder split."""
import sys, os, time
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
Bs = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 16, 64, 128, 256, 435, 512, 600, 700, 870, 1024]
for B in Bs:
    x = torch.randint(0, 256, (B, 3, 224, 224), generator=g, device=dev, dtype=torch.uint8)
    for _ in range(3): model.encode_image(x, normalize=True)
    torch.cuda.synchronize()
    reps = max(5, 6000 // B)
    t0 = time.perf_counter()
    for _ in range(reps): model.encode_image(x, normalize=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"B={B}: {ms:.3f} ms  {B / ms * 1e3:,.0f} images/s", flush=True)
