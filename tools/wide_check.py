"""Wide coarse pass (development aid): bit-exactness against the exact f32 scan and timing, several Q.
usage: wide_check.py [N] [Q,Q,...]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Qs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [128, 256, 1024]
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
    ok = "unchecked"
    if os.environ.get("WC_CHECK", "1") == "1":
        se, ie = ex.search_device(q, K); torch.cuda.synchronize()
        ok = bool(torch.equal(i, ie) and torch.equal(s.view(torch.int32), se.view(torch.int32)))
        if not ok:
            bad = (i != ie).any(dim=1).nonzero().flatten()
            print("  mismatching queries:", bad[:20].tolist(), "of", len(bad), flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(3): idx.search_device(q, K)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    print(f"N={N} Q={Q}: {dt * 1e3:.3f} ms per search, {Q / dt:.0f} q/s, exact: {ok}", flush=True)
