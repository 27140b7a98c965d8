"""gemm256 (algo 2) vs the persistent role-split gemm256p (algo 3) on bf16-store shapes (development aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
shapes = [(21750, 3072, 768, 1), (21750, 2304, 768, 0), (43500, 3072, 768, 1), (43500, 2304, 768, 0),
          (36928, 4096, 1024, 1), (36928, 3072, 1024, 0), (12800, 3072, 768, 1), (6400, 3072, 768, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]]
for (M, N, K, epi) in shapes:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    out = torch.zeros(M, N, dtype=torch.float32 if epi in (2, 3) else torch.bfloat16, device=dev)
    res = {}
    for rnd in range(3):
        for algo in (2, 3):
            def run():
                clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi | (algo << 8), None), "gemm")
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(algo, []).append(e0.elapsed_time(e1) / 20)
    tiles = (N // 256) * ((M + 255) // 256)
    fl = 2.0 * M * N * K
    t2, t3 = min(res[2]), min(res[3])
    print(f"M={M} N={N} K={K} epi={epi}: tiles={tiles} ({tiles/256:.2f} rounds)  gemm256 {t2*1e3:.1f} us {fl/t2*1e-9:.0f} TF   gemm256p {t3*1e3:.1f} us {fl/t3*1e-9:.0f} TF", flush=True)
