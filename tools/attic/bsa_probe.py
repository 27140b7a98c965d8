"""Development probe of the block-scaled FP8 GEMM (gemm256f8 BSA): which lanes / blocks a scale byte reaches."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(0)
M, N, K = 256, 256, 256
a = torch.randn(M, K, generator=g)
w = torch.randn(N, K, generator=g) * 0.1
a8 = a.to(torch.float8_e4m3fn); w8 = w.to(torch.float8_e4m3fn)
ad, wd = a8.float(), w8.float()
sw = torch.ones(N)
a8d, w8d, swd = a8.view(torch.uint8).to(dev), w8.view(torch.uint8).to(dev), sw.to(dev)
def run(sb):
    out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    sbd = sb.to(dev)
    clipmi._lib.check(L.clipmi_dbg_gemm_fp8_bsa(a8d.data_ptr(), w8d.data_ptr(), sbd.data_ptr(), swd.data_ptr(), None, out.data_ptr(), M, N, K, 3, None), "bsa")
    torch.cuda.synchronize()
    return out.cpu()
def ref(sb):
    s = torch.ldexp(torch.ones(M, K // 32), sb.to(torch.int32) - 127).repeat_interleave(32, dim=1)
    return (ad * s).double() @ wd.double().t()
for name, sb in [("unit", torch.full((M, K // 32), 127, dtype=torch.uint8)),
                 ("all x2", torch.full((M, K // 32), 128, dtype=torch.uint8)),
                 ("block 0 x4", torch.full((M, K // 32), 127, dtype=torch.uint8).index_fill_(1, torch.tensor([0]), 129)),
                 ("block 5 x4", torch.full((M, K // 32), 127, dtype=torch.uint8).index_fill_(1, torch.tensor([5]), 129)),
                 ("row 3 x4", torch.full((M, K // 32), 127, dtype=torch.uint8).index_fill_(0, torch.tensor([3]), 129)),
                 ("row 200 x4", torch.full((M, K // 32), 127, dtype=torch.uint8).index_fill_(0, torch.tensor([200]), 129))]:
    o = run(sb); r = ref(sb).float()
    err = (o - r).abs().max().item()
    print(f"{name}: max err {err:.4g} of {r.abs().max().item():.4g}; rows wrong: {((o - r).abs().amax(1) > 1e-2).nonzero().flatten()[:12].tolist()}", flush=True)
    if name.startswith("block"):
        # which single-block hypothesis explains the output: try every block index
        for b in range(K // 32):
            sb2 = torch.full((M, K // 32), 127, dtype=torch.uint8); sb2[:, b] = 129
            if (o - ref(sb2).float()).abs().max().item() < 1e-2: print(f"   output matches 'block {b} x4'")
sb = torch.full((M, K // 32), 127, dtype=torch.uint8)
o = run(sb)
nan = torch.isnan(o)
print("NaNs:", nan.sum().item(), "rows with NaN:", nan.any(1).nonzero().flatten()[:20].tolist(), "cols with NaN:", nan.any(0).nonzero().flatten()[:20].tolist())
out = torch.zeros(M, N, dtype=torch.float32, device=dev)
ones = torch.ones(M, device=dev)
clipmi._lib.check(L.clipmi_dbg_gemm_fp8(a8d.data_ptr(), w8d.data_ptr(), ones.data_ptr(), swd.data_ptr(), None,
                                        out.data_ptr(), M, N, K, 3, None), "plain")
torch.cuda.synchronize()
print("row-scaled kernel err:", (out.cpu() - ref(sb).float()).abs().max().item())
good = ~nan
print("BSA err on finite entries:", ((o - ref(sb).float()).abs() * good).max().item())
# which 16-wide k chunks does a block's scale byte reach?
o_unit = run(torch.full((M, K // 32), 127, dtype=torch.uint8)).double()
C = torch.stack([(ad[:, 16 * j:16 * j + 16].double() @ wd[:, 16 * j:16 * j + 16].double().t()).flatten() for j in range(K // 16)], dim=1)
for b in range(K // 32):
    sb = torch.full((M, K // 32), 127, dtype=torch.uint8); sb[:, b] = 129
    d = (run(sb).double() - o_unit).flatten() / 3.0
    x = torch.linalg.lstsq(C, d[:, None]).solution.flatten()
    print(f"scale byte of block {b} reaches 16-chunks {[j for j in range(K // 16) if x[j] > 0.5]} (fit {[round(v, 2) for v in x.tolist()]})", flush=True)
