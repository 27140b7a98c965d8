"""bf16 vs int8 coarse path at full size: scan-kernel time, survivors, whole-call queries/s (development aid)."""
import sys, os, time, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = 51
g = torch.Generator(device=dev); g.manual_seed(1)
db = torch.empty((N, 512), dtype=torch.float32, device=dev)
for lo in range(0, N, 1 << 20):
    blk = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev)
    db[lo:lo + blk.shape[0]] = blk / blk.norm(dim=1, keepdim=True)
res = {}
for kind in ("bf16", "int8"):
    idx = clipmi.IndexFlatIP(512, device=dev, coarse=kind)
    idx.add(db)
    for Q in (1, 16, 64):
        q = torch.randn((Q, 512), generator=g, device=dev); q = q / q.norm(dim=1, keepdim=True)
        for _ in range(2): s, i = idx.search_device(q, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): s, i = idx.search_device(q, K)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res[(kind, Q)] = (s.clone(), i.clone())
        ws = torch.empty(L.clipmi_topk_ip_coarse_workspace_bytes(N, 512, Q, K), dtype=torch.uint8, device=dev)
        sm, sv = C.c_float(0), C.c_longlong(0)
        os_, oi_ = torch.empty_like(s), torch.empty_like(i)
        if kind == "bf16":
            dbh, rmax = idx.matrix_bf16()
            clipmi._lib.check(L.clipmi_dbg_topk_coarse_scan_ms(db.data_ptr(), dbh.data_ptr(), N, 512, rmax, q.data_ptr(), Q, K, os_.data_ptr(), oi_.data_ptr(), ws.data_ptr(), ws.numel(), None, 5, C.byref(sm), C.byref(sv)), "x")
            byt = N * 1024
        else:
            d8, meta, amax, rmax = idx.matrix_i8()
            clipmi._lib.check(L.clipmi_dbg_topk_coarse_i8_scan_ms(db.data_ptr(), d8.data_ptr(), meta.data_ptr(), amax, N, 512, rmax, q.data_ptr(), Q, K, os_.data_ptr(), oi_.data_ptr(), ws.data_ptr(), ws.numel(), None, 5, C.byref(sm), C.byref(sv)), "x")
            byt = N * 520
        print(f"{kind} N={N} Q={Q}: call {ms:.3f} ms = {Q/ms*1e3:.0f} q/s; scan {sm.value:.3f} ms = {byt/sm.value/1e6:.0f} GB/s; survivors/query {sv.value/Q:.0f}", flush=True)
    del idx
for Q in (1, 16, 64):
    a, b = res[("bf16", Q)], res[("int8", Q)]
    print("Q", Q, "identical:", torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
