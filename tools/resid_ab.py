"""Plain f32 residual epilogue vs the split-residual producer (hi / lo + statistics) of the persistent GEMM, same process,
interleaved rounds (development aid)."""
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
    x3 = torch.zeros(M, 3 * N, dtype=torch.uint8, device=dev)        # split rows: N bf16 hi | N int8 lo
    part = torch.zeros(M, N // 256, 2, device=dev)
    tmp = torch.zeros(M, N, device=dev)
    def plain():
        clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), M, N, K, 2 | (3 << 8), None), "gemm")
    def rln():
        clipmi._lib.check(L.clipmi_dbg_gemm_resid_ln(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x3.data_ptr(),
                                                     part.data_ptr(), tmp.data_ptr(), M, N, K, 3, None), "rln")
    res = {"plain": [], "rln": []}
    for rnd in range(5):
        for name, fn in (("plain", plain), ("rln", rln)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
            x.zero_(); x3.zero_()
    print(f"M={M} N={N} K={K}: plain {min(res['plain']):.1f} us   split-residual + stats {min(res['rln']):.1f} us", flush=True)
