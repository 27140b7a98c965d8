"""Attention kernel time vs batch: where does the extra round of workgroups start? (development aid)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
for B in ([int(x) for x in sys.argv[1:]] or (256, 320, 340, 341, 342, 360, 400, 420, 426, 427, 430, 435, 512)):
    qkv = (torch.randn(B * 50, 2304, device=dev) * 1.5).to(torch.bfloat16)
    out = torch.empty(B * 50, 768, dtype=torch.bfloat16, device=dev)
    def run():
        clipmi._lib.check(L.clipmi_dbg_attention(qkv.data_ptr(), out.data_ptr(), B, 50, 12, 0, None), "attn")
    for _ in range(3): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"B={B}: {best*1e3:.1f} us  ({B*12} waves, {B*12/256:.1f} per CU)", flush=True)
