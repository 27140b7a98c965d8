"""Throughput of ONE search of many queries (Q = 1024 by default): passes pipelined on two internal streams vs one stream
(development aid)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator(device=dev); g.manual_seed(1)
idx = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
for lo in range(0, N, 1 << 20):
    x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    idx.add(x)
q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
if os.environ.get("LQ_PRE"):          # what bench.py does first: 64-query batches on two streams of its own
    q64 = q[:64].contiguous()
    st = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for i in range(8):
        with torch.cuda.stream(st[i % 2]):
            idx.search_device(q64, 51)
    torch.cuda.synchronize()
ref = None
for rep in range(2):
    for nfl in (1, 2):
        idx.batches_in_flight = nfl
        idx.search_device(q, 51); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): s, i = idx.search_device(q, 51)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        if ref is None: ref = i.clone()
        print(f"N={N} Q={Q} passes in flight {nfl}: {dt * 1e3:.2f} ms per search, {Q / dt:.0f} q/s, same ids: {torch.equal(i, ref)}", flush=True)
