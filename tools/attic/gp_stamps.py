"""In-kernel wall-clock stamps of the persistent GEMM (development aid; CLIPMI_GEMM_DBG=4[+1/2])."""
import sys, os
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
os.environ["CLIPMI_GEMM_DBG"] = os.environ.get("CLIPMI_GEMM_DBG", "4")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
L = clipmi._lib.lib()
dev = torch.device("cuda:0")
M, N, K, epi = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "21750,3072,768,1").split(",")]
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
st = torch.zeros(1024, dtype=torch.int64, device=dev)
out = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.bfloat16, device=dev)
for _ in range(5):
    clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), st.data_ptr(), out.data_ptr(), M, N, K, epi | (3 << 8), None), "gemm")
torch.cuda.synchronize()
s = st.cpu()[:512].view(2, 8, 4, 8)
names = ["top", "kloop", "kdone", "p0w", "p0s", "p1w", "p1s", "end"]
for wg in range(2):
    base = s[wg, 0, 0, 0].item()
    for wave in (0, 4):
        print(f"wg {'0' if wg == 0 else '100'} wave {wave} (us from first stamp; {names})")
        for t in range(4):
            print("   tile", t, " ".join(f"{(s[wg, wave, t, k].item() - base) / 100.0:7.2f}" for k in range(8)))
if int(os.environ["CLIPMI_GEMM_DBG"]) & 8:
    raw = st.cpu()
    for wg in range(2):
        for wave in range(8):
            k = raw[wg * 256 + 128 + wave * 9: wg * 256 + 128 + wave * 9 + 9].tolist()
            print(f"wg {'0' if wg == 0 else '100'} wave {wave} K-tile 5 slot cycles (load,mfma x4):", [k[i + 1] - k[i] for i in range(8)], "total", k[8] - k[0])
