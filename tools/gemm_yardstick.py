"""Yardstick only (not a product path): what the vendor library's bf16 GEMM reaches on the tower's shapes, to judge the
hand-written kernels' distance from what the machine sustains. python tools/gemm_yardstick.py"""
import torch, time
dev = torch.device("cuda:0")
M = 43500
for (N, K, name) in [(2304, 768, "qkv"), (768, 768, "out_proj"), (3072, 768, "c_fc"), (768, 3072, "c_proj"), (8192, 8192, "square 8192 (M=8192)")]:
    m = 8192 if N == 8192 else M
    a = torch.randn((m, K), device=dev, dtype=torch.bfloat16)
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    for _ in range(5): c = a @ w.t()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: M={m} N={N} K={K}: {ms * 1e3:.1f} us, {2.0 * m * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
