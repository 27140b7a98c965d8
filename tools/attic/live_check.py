"""Live-threshold scan (development aid): bit-exactness vs the exact f32 scan + timing at several N / Q."""
import sys, os, time
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
os.environ.setdefault("CLIPMI_LIVE", "1")      # the live scan is off by default (topk.hip)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Qs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 16, 64]
K = 51
g = torch.Generator(device=dev); g.manual_seed(1)
idx = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
ex = clipmi.IndexFlatIP(512, device=dev)
for lo in range(0, N, 1 << 20):
    x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    idx.add(x)
ex._chunks = [idx.matrix()]
idx.matrix_i8()
for Q in Qs:
    q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
    s, i = idx.search_device(q, K); torch.cuda.synchronize()
    se, ie = ex.search_device(q, K); torch.cuda.synchronize()
    ok = bool(torch.equal(i, ie) and torch.equal(s.view(torch.int32), se.view(torch.int32)))
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(10): idx.search_device(q, K)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    print(f"N={N} Q={Q}: {dt * 1e3:.3f} ms per search, {Q / dt:.0f} q/s, exact: {ok}", flush=True)
