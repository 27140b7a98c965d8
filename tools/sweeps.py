"""The sweeps SURVEY.md 8(d) cfg-2 names: encode images/s over B, search queries/s over Q at 1 M and 10 M rows (int8 coarse
path, K = 51). One process, synthetic data (development aid; bench.py is the contract)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
def t_ms(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("| B | ms/batch | images/s |"); print("|---|---|---|")
for B in (1, 16, 64, 128, 256, 435, 512, 870, 1740):
    x = torch.randint(0, 256, (B, 3, 224, 224), generator=g, device=dev, dtype=torch.uint8)
    ms = t_ms(lambda: model.encode_image(x, normalize=True), max(5, 4000 // B))
    print(f"| {B} | {ms:.3f} | {B / ms * 1e3:,.0f} |", flush=True)
del model
for N in (1_000_000, 10_000_000):
    idx = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
    for lo in range(0, N, 1 << 20):
        x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
        idx.add(x)
    print(f"\nN = {N:,}"); print("| Q | ms/call | queries/s |"); print("|---|---|---|")
    for Q in (1, 16, 64, 256, 1024):
        q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
        ms = t_ms(lambda: idx.search_device(q, 51), 20 if Q <= 64 else 5)
        print(f"| {Q} | {ms:.3f} | {Q / ms * 1e3:,.0f} |", flush=True)
    del idx
    torch.cuda.empty_cache()
