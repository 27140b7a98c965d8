"""Device-side timing of encode_image with the FP8 linear layers (BASELINE.json configs[4]; development aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
sd = clipmi.weights.random_state_dict("ViT-B/32", seed=0)
g = torch.Generator(device=dev); g.manual_seed(0)
for fmt in ("bf16", "fp8"):
    model = clipmi.CLIP(sd, device=dev, vision_weights=fmt)
    for B in [int(x) for x in sys.argv[1:]] or [435, 870]:
        x = torch.randint(0, 256, (B, 3, 224, 224), generator=g, device=dev, dtype=torch.uint8)
        for _ in range(2): model.encode_image(x, normalize=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(2, 2048 // B)
        e0.record()
        for _ in range(reps): model.encode_image(x, normalize=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{fmt} B={B}: {ms:.3f} ms/batch  {B/ms*1e3:.0f} img/s", flush=True)
