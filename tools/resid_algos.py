"""Residual GEMM (f32 in-place epilogue) on the three kernels for the ViT-B/32 out_proj / c_proj shapes (development aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
for (M, N, K) in ((43500, 768, 768), (43500, 768, 3072)):
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    x = torch.randn(M, N, generator=g, device=dev)
    res = {}
    for rnd in range(3):
        for algo in (1, 2, 3):
            def run():
                clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), M, N, K, 2 | (algo << 8), None), "gemm")
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(algo, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            x.zero_()
    print(f"M={M} N={N} K={K}: gemm128 {min(res[1]):.1f} us, gemm256 {min(res[2]):.1f} us, gemm256p {min(res[3]):.1f} us", flush=True)
