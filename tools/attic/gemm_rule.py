"""Where does the 256x256 kernel stop beating the 128x128 one as its last round empties? (development aid)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
for (M, N, K, epi) in [(36928, 1024, 1024, 2), (36928, 1024, 4096, 2), (36928, 3072, 1024, 0), (36928, 4096, 1024, 1),
                       (25600, 768, 768, 2), (25600, 768, 3072, 2), (12800, 2304, 768, 0), (12800, 3072, 768, 1),
                       (16000, 768, 3072, 2), (19200, 2304, 768, 0)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    out = torch.zeros(M, N, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16, device=dev)
    res = {}
    for rnd in range(3):
        for algo in (1, 2):
            def run():
                clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi | (algo << 8), None), "gemm")
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(algo, []).append(e0.elapsed_time(e1) / 10)
    tiles = (N // 256) * ((M + 255) // 256); rounds = (tiles + 255) // 256
    print(f"M={M} N={N} K={K} epi={epi}: 256-tiles={tiles} fill={tiles/(rounds*256):.2f}  v128 {min(res[1])*1e3:.0f} us  v256 {min(res[2])*1e3:.0f} us", flush=True)
