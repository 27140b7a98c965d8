"""A/B of the search leg only (development aid): prints one-in-flight / two-in-flight ms at Q = 1, 16, 32, 64 over N rows.
usage: [CLIPMI_LIVE=1 CLIPMI_LIVE_NSCAN=6] python tools/search_ab.py [N]"""
import sys, os, time
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = 51
g = torch.Generator(device=dev); g.manual_seed(1)
idx = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
for lo in range(0, N, 1 << 20):
    x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    idx.add(x)
idx.matrix_i8()
s2 = torch.cuda.Stream()
idx2 = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
idx2._chunks = idx._chunks; idx2._i8 = getattr(idx, "_i8", None)
for Q in (1, 16, 32, 64):
    q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
    for _ in range(3): idx.search_device(q, K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): idx.search_device(q, K)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / 20
    print(f"N={N} Q={Q}: one in flight {one * 1e3:.3f} ms ({5.2e9 * N / 1e7 / one / 8e12:.3f} of 8 TB/s)", flush=True)
