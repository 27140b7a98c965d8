"""Is the residual producer's store pass bound by HBM (all 256 CUs at once) or by each CU's own memory pipe? The persistent
kernel on 256 / 192 / 128 / 64 workgroups (CLIPMI_GEMM_GRID, read once per process: one child per value), bias-only epilogue
beside it for scale (development aid)."""
import sys, os, subprocess
os.environ.setdefault("CLIPMI_DEV_LIB", "1")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import clipmi
    L = clipmi._lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0)
    for (M, N, K) in ((43500, 768, 768), (43500, 768, 3072)):
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=dev)
        x3 = torch.zeros(M, 3 * N, dtype=torch.uint8, device=dev)      # split rows: N bf16 hi | N u8 lo (round 5)
        part = torch.zeros(M, N // 256, 2, device=dev)
        tmp = torch.zeros(M, N, device=dev)
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        def rln():
            clipmi._lib.check(L.clipmi_dbg_gemm_resid_ln(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x3.data_ptr(),
                                                         part.data_ptr(), tmp.data_ptr(), M, N, K, 3, None), "rln")
        def plain():
            clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, 0 | (3 << 8), None), "gemm")
        best = {}
        for name, fn in (("resid", rln), ("bias", plain)):
            ts = []
            for rnd in range(4):
                for _ in range(3): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
                x3.zero_()
            best[name] = min(ts)
        print(f"grid {os.environ.get('CLIPMI_GEMM_GRID', '256'):>3} K={K}: split-residual producer {best['resid']:.1f} us, bias -> bf16 {best['bias']:.1f} us, difference {best['resid'] - best['bias']:.1f} us", flush=True)
else:
    for grid in ("256", "192", "128", "64"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CLIPMI_GEMM_GRID=grid))
