"""Kernel timeline of ONE clipmi_topk_ip_coarse call (development aid).
run:    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/search_timeline.py run N Q K
report: python3 tools/search_timeline.py report gpurun_out/tl"""
import sys, os, glob, csv
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import clipmi
    if os.environ.get("AB_LIB"):            # same-box A/B against another build of the library (development only)
        import importlib
        _b = importlib.import_module("cli-p_amd.build")
        clipmi._lib.LIB_PATH = os.environ["AB_LIB"]
        _b.is_current = lambda: True
    dev = torch.device("cuda:0")
    N, Q, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    g = torch.Generator(device=dev); g.manual_seed(1)
    idx = clipmi.IndexFlatIP(512, device=dev, coarse="int8")
    for lo in range(0, N, 1 << 20):
        x = torch.randn((min(1 << 20, N - lo), 512), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
        idx.add(x)
    q = torch.randn((Q, 512), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
    import time
    nfl = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % nfl]):
                idx.search_device(q, K)
    run(4); torch.cuda.synchronize()
    best = []
    for rep in range(int(os.environ.get("AB_REPS", "1"))):
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 40 * 1e3)
    print(f"in_flight={nfl}: {min(best):.4f} ms per batch of {Q} (min of {len(best)}; median {sorted(best)[len(best) // 2]:.4f})", flush=True)
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "clipmi" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # calls start at coarse_prep_kernel
    starts = [i for i, r in enumerate(rows) if "coarse_prep" in r["Kernel_Name"]]
    i0, i1 = starts[-3], starts[-1]
    t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("clipmi::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3} lds {int(r.get('LDS_Block_Size', 0)) // 1024:3d}K {name}")
        prev_end = e
    print(f"two calls: {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us start-to-start")
