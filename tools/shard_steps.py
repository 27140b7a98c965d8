"""Is the 12.5 M-row search faster with two calls in flight than with one, and does the answer depend on how many calls are
timed? (in a fresh process it is, at every length; bench.py's old shard leg lost because its streams shared a
hardware queue.) usage: python tools/shard_steps.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, clipmi
dev = torch.device("cuda:0")
N, K, Q = 12_500_000, 51, 64
g = torch.Generator(device=dev); g.manual_seed(5000)
idx = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
for lo in range(0, N, 1 << 20):
    x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    idx.add(x)
idx.matrix_i8()
q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def run(nfl, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        if nfl == 1: idx.search_device(q, K)
        else:
            with torch.cuda.stream(streams[i % 2]): idx.search_device(q, K)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for nfl in (1, 2):
    run(nfl, 4)
    for steps in (10, 20, 40, 80):
        print(f"in flight {nfl}, {steps} steps: {run(nfl, steps):.3f} ms per call", flush=True)
# host cost of one enqueue
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): idx.search_device(q, K)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host enqueue time per call: {(t1 - t0) / 50 * 1e3:.3f} ms")
