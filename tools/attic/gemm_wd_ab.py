"""W fragments straight from L2 (gemm256 WD) against the LDS-DMA form of the SAME kernel (development aid):
bit-identity of the outputs, then interleaved timing rounds in one process on the encode step's GEMM shapes.
algo 2 = gemm256 (both operands by LDS-DMA), 5 = gemm256 WD, 3 = the persistent role-split kernel (for scale)."""
import sys, os
os.environ.setdefault("CLIPMI_DEV_LIB", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
shapes = [(43500, 768, 768), (43500, 2304, 768), (43500, 3072, 768), (43500, 768, 3072)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
g = torch.Generator(device="cpu"); g.manual_seed(0)
for (M, N, K) in shapes:
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    outs = {}
    def run(algo, out):
        clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, 0 | (algo << 8), None), f"algo {algo}")
    for algo in (2, 5, 3):
        outs[algo] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        run(algo, outs[algo])
    torch.cuda.synchronize()
    same = torch.equal(outs[2], outs[5]) and torch.equal(outs[2], outs[3])
    ref = (a[:512].float() @ w.float().t() + bias)
    err = (outs[5][:512].float() - ref).abs().max().item()
    best = {2: 1e9, 5: 1e9, 3: 1e9}
    for rnd in range(5):
        for algo in (2, 5, 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(algo, outs[algo])
            e0.record()
            for _ in range(10): run(algo, outs[algo])
            e1.record(); torch.cuda.synchronize()
            best[algo] = min(best[algo], e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: identical={same} err={err:.3g} | gemm256 {best[2]*1e3:.1f} us ({fl/best[2]/1e9:.0f} TF) | "
          f"WD {best[5]*1e3:.1f} us ({fl/best[5]/1e9:.0f} TF) | persistent {best[3]*1e3:.1f} us ({fl/best[3]/1e9:.0f} TF)", flush=True)
