"""Latency of encode_text (+ normalise) at interactive batch sizes (development aid): host ids (the tokenizer's output:
the tower runs on EOT + 1 positions) and device-resident ids (all 77 positions)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
for Q in (1, 16, 64, 256):
    ids = torch.zeros(Q, 77, dtype=torch.int64)
    ids[:, 0] = 49406; ids[:, 1:9] = torch.randint(1, 40000, (Q, 8)); ids[:, 9] = 49407
    for where, t in (("host ids, 10 tokens", ids), ("device ids, 77 positions", ids.to(dev))):
        for _ in range(3): model.encode_text(t, normalize=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): model.encode_text(t, normalize=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"encode_text Q={Q} ({where}): {ms:.3f} ms  {Q/ms*1e3:.0f} q/s", flush=True)
imgs = torch.randint(0, 256, (1, 3, 224, 224), dtype=torch.uint8, device=dev)
for _ in range(3): model.encode_image(imgs, normalize=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): model.encode_image(imgs, normalize=True)
e1.record(); torch.cuda.synchronize()
print(f"encode_image B=1: {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
