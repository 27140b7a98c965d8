"""Supertile shape (GM row panels x GN column panels run together inside one XCD) of the persistent GEMM's tile order:
time per launch for the ViT-B/32 shapes at B = 870 (development aid; CLIPMI_GEMM_DBG carries GM << 8 | GN << 16)."""
import sys, os, subprocess
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import clipmi
    L = clipmi._lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0)
    for (M, N, K, epi) in ((43500, 3072, 768, 1), (43500, 2304, 768, 0), (43500, 768, 3072, 2), (43500, 768, 768, 2)):
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=dev)
        out = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.bfloat16, device=dev)
        def run():
            clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi | (3 << 8), None), "gemm")
        for _ in range(5): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"   N={N} K={K} epi={epi}: {min(ts):.1f} us", flush=True)
    sys.exit(0)
for gm, gn in ((8, 4), (4, 4), (16, 4), (8, 6), (4, 6), (6, 6), (8, 3), (16, 2), (8, 12), (2, 12), (32, 3)):
    env = dict(os.environ, CLIPMI_GEMM_DBG=str((gm << 8) | (gn << 16)))
    print(f"GM={gm} GN={gn}", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
