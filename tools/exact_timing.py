"""Exact f32 scan (clipmi_topk_ip, the path without a coarse copy): time per call at Q = 32 (development aid). usage: exact_timing.py [N]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
idx = clipmi.IndexFlatIP(512, device=dev)
for lo in range(0, N, 1 << 20):
    x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    idx.add(x)
for Q in (1, 32):
    q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
    for _ in range(2): idx.search_device(q, 51)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): idx.search_device(q, 51)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"exact N={N} Q={Q}: {dt * 1e3:.3f} ms per call = {N * 2048 / dt / 1e12:.2f} TB/s", flush=True)
