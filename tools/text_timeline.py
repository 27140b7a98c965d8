"""Kernel timeline of ONE encode_text call for one prompt (development aid).
run:    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -- python3 tools/text_timeline.py run
report: python3 tools/text_timeline.py report gpurun_out/tt"""
import sys, os, glob, csv
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import clipmi
    dev = torch.device("cuda:0")
    model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
    ids = torch.zeros(1, 77, dtype=torch.int64)
    ids[:, 0] = 49406; ids[:, 1:9] = torch.randint(1, 40000, (1, 8)); ids[:, 9] = 49407
    for _ in range(5): model.encode_text(ids, normalize=True)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "text_embed" in r["Kernel_Name"]]
    i0, i1 = starts[-2], starts[-1]
    t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
    tot = {}
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("clipmi::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
        if len(sys.argv) > 3: print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  grid {r.get('Grid_Size_X','?'):>7} {name}")
        d = tot.setdefault(name, [0, 0.0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3; d[2] += (s - prev_end) / 1e3
        prev_end = e
    for name, (n, dur, gap) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:4d} x {name:52s} dur {dur:8.1f} us  (avg {dur / n:6.1f})  gaps in front {gap:7.1f} us")
    print(f"one call: {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us start-to-start, {i1 - i0} kernels")
