"""Quick device-side timing of clipmi_topk_ip (development aid; bench.py is the contract)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
db = torch.randn((N, 512), generator=g, device=dev, dtype=torch.float32)
db /= db.norm(dim=1, keepdim=True)
coarse = len(sys.argv) > 2 and sys.argv[2] == "coarse"
idx = clipmi.IndexFlatIP(512, device=dev, coarse="bf16" if coarse else None); idx.add(db)
print("coarse bf16 path" if coarse else "exact f32 path", flush=True)
for Q in ((1, 16, 64, 128, 256) if coarse else (1, 16, 32, 64)):
    for K in (11, 51, 101):
        q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
        for _ in range(3): idx.search_device(q, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): idx.search_device(q, K)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        passes = (Q + 63) // 64 if coarse else (Q + 31) // 32
        print(f"N={N} Q={Q} K={K}: {ms:.3f} ms/batch  {Q/ms*1e3:.0f} q/s  {passes*N*(1024 if coarse else 2048)/ms/1e6:.1f} GB/s algorithmic", flush=True)
