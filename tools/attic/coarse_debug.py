import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
for N in (1_250_000, 10_000_000):
    Q, K = 64, 51
    g = torch.Generator(device=dev); g.manual_seed(1)
    db = torch.randn((N, 512), generator=g, device=dev); db /= db.norm(dim=1, keepdim=True)
    q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
    idx = clipmi.IndexFlatIP(512, device=dev, coarse="bf16"); idx.add(db)
    dbh, rmax = idx.matrix_bf16()
    ws = torch.empty(L.clipmi_topk_ip_coarse_workspace_bytes(N, 512, Q, K), dtype=torch.uint8, device=dev)
    os_ = torch.empty((64, K), dtype=torch.float32, device=dev); oi_ = torch.empty((64, K), dtype=torch.int64, device=dev)
    ms, sv = C.c_float(0), C.c_longlong(-1)
    clipmi._lib.check(L.clipmi_dbg_topk_coarse_scan_ms(db.data_ptr(), dbh.data_ptr(), N, 512, rmax, q.data_ptr(), Q, K, os_.data_ptr(), oi_.data_ptr(), ws.data_ptr(), ws.numel(), None, 3, C.byref(ms), C.byref(sv)), "x")
    print(f"N={N}: scan {ms.value:.3f} ms, survivors total {sv.value} = {sv.value/Q:.0f} per query, rmax {rmax}", flush=True)
    del db, idx, dbh, ws
