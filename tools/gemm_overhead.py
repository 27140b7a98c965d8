"""Per-tile overhead of gemm256: time vs K at fixed M, N (development aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
M, N = 21750, 3072          # 1020 tiles = 4 rounds
g = torch.Generator(device=dev); g.manual_seed(0)
for epi in (0, 1, 2, 3):
    for K in (128, 256, 768, 1536, 3072):
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=dev)
        out = torch.zeros(M, N, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16, device=dev)
        def run():
            clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi | (2 << 8), None), "gemm")
        for _ in range(3): run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print(f"epi={epi} K={K}: {best*1e3:.1f} us total, {best*1e3/4:.2f} us/round, {2.0*M*N*K/best/1e9:.0f} TF", flush=True)
