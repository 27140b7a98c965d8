import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
N, Q, K = 10_000_000, 64, 51
g = torch.Generator(device=dev); g.manual_seed(1)
db = torch.randn((N, 512), generator=g, device=dev); db /= db.norm(dim=1, keepdim=True)
q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
idx = clipmi.IndexFlatIP(512, device=dev, coarse="bf16"); idx.add(db)
s, i = idx.search_device(q, K)
torch.cuda.synchronize()
ws = idx._ws
off = 32 * 81920 * 8 + 64 * 262144 * 8
gc_e = ws[off:off + 128].view(torch.int32).cpu().numpy()
gc_c = ws[off + 256:off + 512].view(torch.int32).cpu().numpy()
thr0 = ws[off + 512:off + 768].view(torch.float32).cpu().numpy()
tauc = ws[off + 768:off + 1024].view(torch.float32).cpu().numpy()
print("gcnt_c", gc_c[:8], gc_c.sum(), "thr0", thr0[:4], "tauc", tauc[:4], "kth", s[:4, -1].cpu().numpy())
