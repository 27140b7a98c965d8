"""Statistics of the live-threshold scan (development aid): CLIPMI_LIVE_STATS=1 python tools/live_stats.py [N] [Q]"""
import sys, os, ctypes as C
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
os.environ["CLIPMI_LIVE_STATS"] = "1"
os.environ.setdefault("CLIPMI_LIVE", "1")      # the live scan is off by default (topk.hip)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = 51
g = torch.Generator(device=dev); g.manual_seed(1)
idx = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
for lo in range(0, N, 1 << 20):
    x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    idx.add(x)
q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
L = clipmi._lib.lib()
db = idx.matrix(); db8, meta, amax, rmax = idx.matrix_i8()
ws = torch.empty(L.clipmi_topk_ip_coarse_workspace_bytes(N, 512, Q, K), dtype=torch.uint8, device=dev)
os_ = torch.empty((Q, K), dtype=torch.float32, device=dev); oi_ = torch.empty((Q, K), dtype=torch.int64, device=dev)
ms, sv = C.c_float(0), C.c_longlong(0)
clipmi._lib.check(L.clipmi_dbg_topk_coarse_i8_scan_ms(db.data_ptr(), db8.data_ptr(), meta.data_ptr(), amax, N, 512, rmax, q.data_ptr(), Q, K,
                                                      os_.data_ptr(), oi_.data_ptr(), ws.data_ptr(), ws.numel(), None, 3, C.byref(ms), C.byref(sv)), "x")
print(f"N={N} Q={Q}: scan {ms.value:.3f} ms, list entries per query {sv.value / Q:.0f}")
