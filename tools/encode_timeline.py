"""Kernel timeline of encode_image steps: busy time, idle gaps between kernels (development aid).
run:    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/etl -- python3 tools/encode_timeline.py run [B] [fp8]
report: python3 tools/encode_timeline.py report gpurun_out/etl"""
import sys, os, glob, csv
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import clipmi
    dev = torch.device("cuda:0")
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 870
    fp8 = len(sys.argv) > 3 and sys.argv[3] == "fp8"
    model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev, **({"vision_weights": "fp8"} if fp8 else {}))
    x = torch.randint(0, 256, (B, 3, 224, 224), device=dev, dtype=torch.uint8)
    for _ in range(12):
        model.encode_image(x, normalize=True)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "clipmi" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
    for a, b in zip(starts[-4:-1], starts[-3:]):
        step = rows[a:b]
        t0 = int(step[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
        gaps = [int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"]) for i in range(len(step) - 1)]
        print(f"step: {len(step)} kernels, start-to-start {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, "
              f"gaps sum {sum(gaps) / 1e3:.1f} us (max {max(gaps) / 1e3:.1f}, mean {sum(gaps) / len(gaps) / 1e3:.2f})")
    step = rows[starts[-2]:starts[-1]]
    agg = {}
    for r in step:
        n = r["Kernel_Name"].replace("clipmi::", "").replace("void ", "").split("(")[0][:50]
        d = agg.setdefault(n, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:52s} x{c:3d}  {t / 1e3:8.1f} us  avg {t / c / 1e3:7.1f}")
