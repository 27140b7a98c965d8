"""Search leg, one and two calls in flight (development A/B of env knobs). usage: python tools/search_ab2.py [N] [Q]"""
import sys, os, time
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = 51
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = torch.Generator(device=dev); g.manual_seed(1)
idx = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
for lo in range(0, N, 1 << 20):
    x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    idx.add(x)
db = idx.matrix(); db8, meta, amax, rmax = idx.matrix_i8()
L = clipmi._lib.lib()
qs = []
for _ in range(2):
    q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True); qs.append(q)
wsb = L.clipmi_topk_ip_coarse_workspace_bytes(N, 512, Q, K)
ws = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(2)]
D = [torch.empty((Q, K), dtype=torch.float32, device=dev) for _ in range(2)]
I = [torch.empty((Q, K), dtype=torch.int64, device=dev) for _ in range(2)]
st = [torch.cuda.Stream() for _ in range(2)]
def call(i):
    clipmi._lib.check(L.clipmi_topk_ip_coarse_i8(db.data_ptr(), db8.data_ptr(), meta.data_ptr(), amax, N, 512, rmax, qs[i].data_ptr(), Q, K, 0,
                                                 D[i].data_ptr(), I[i].data_ptr(), ws[i].data_ptr(), wsb, st[i].cuda_stream), "topk")
for nfl in (1, 2):
    for _ in range(4):
        for i in range(nfl): call(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        for i in range(nfl): call(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (20 * nfl)
    print(f"N={N} Q={Q} in flight {nfl}: {dt * 1e3:.3f} ms per call ({5.2e9 * N / 1e7 / dt / 8e12:.3f} of 8 TB/s; {Q / dt / 1e3:.1f} k q/s)", flush=True)
