"""Device-side timing of ViT-L/14@336px encode_image (BASELINE.json configs[3]; development aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-L/14@336px", seed=0), device=dev)
FLOP = 381.92e9
for B in [int(x) for x in sys.argv[1:]] or [16, 64, 128]:
    x = torch.randint(0, 256, (B, 3, 336, 336), device=dev, dtype=torch.uint8)
    for _ in range(2): model.encode_image(x, normalize=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps): model.encode_image(x, normalize=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"ViT-L/14@336 B={B}: {ms:.2f} ms/batch  {B/ms*1e3:.0f} img/s  {FLOP*B/ms/1e9:.0f} TFLOP/s", flush=True)
