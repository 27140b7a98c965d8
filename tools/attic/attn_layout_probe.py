"""Does attention want a head-major qkv? Times the ViT-B/32 attention kernel on the token-major layout ([B*L][3W], a head's
rows 4608 B apart) and on the same bytes arranged per (image, head) ([B*H][L][3][64]: heads = 1 for the kernel's
addressing, 19.2 KB contiguous per item). Development aid."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
L_ = clipmi._lib.lib()
dev = torch.device("cuda:0")
B, L, H = 870, 50, 12
for (b, h) in ((B, H), (B * H, 1)):
    W = h * 64
    qkv = (torch.randn(b * L, 3 * W, device=dev) * 1.5).to(torch.bfloat16)
    out = torch.empty(b * L, W, dtype=torch.bfloat16, device=dev)
    big = torch.empty(300 * 1024 * 1024 // 2, dtype=torch.bfloat16, device=dev)      # flushes the 256 MB Infinity Cache
    for flags in (0, 4):
        ts = []
        for _ in range(12):
            big.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            clipmi._lib.check(L_.clipmi_dbg_attention(qkv.data_ptr(), out.data_ptr(), b, L, h, flags, None), "attn")
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"B={b} heads={h} kernel={'x4' if flags == 0 else 'one-wave'}: median {ts[len(ts)//2]:.1f} us, min {ts[0]:.1f} us", flush=True)
