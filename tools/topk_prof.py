"""One configuration of clipmi_topk_ip for rocprofv3 kernel traces (development aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
N, Q, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device=dev); g.manual_seed(1)
db = torch.randn((N, 512), generator=g, device=dev); db /= db.norm(dim=1, keepdim=True)
q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
coarse = len(sys.argv) > 4 and sys.argv[4] == 'coarse'
idx = clipmi.IndexFlatIP(512, device=dev, coarse='bf16' if coarse else None); idx.add(db)
for _ in range(10): idx.search_device(q, K)
torch.cuda.synchronize()
