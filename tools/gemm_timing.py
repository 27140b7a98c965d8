"""A/B timing of the two GEMM kernels on the encode shapes, interleaved in one process."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shapes = [("qkv", B*50, 2304, 768, 0), ("out", B*50, 768, 768, 2), ("fc", B*50, 3072, 768, 1), ("proj", B*50, 768, 3072, 2), ("patch", B*49, 768, 3072, 3)]
g = torch.Generator(device=dev); g.manual_seed(0)
for name, M, N, K, epi in shapes:
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
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(algo, []).append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * K
    for algo in (1, 2):
        ms = min(res[algo])
        print(f"{name} M={M} N={N} K={K} algo={algo}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s", flush=True)
