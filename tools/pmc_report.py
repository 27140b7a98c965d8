"""Per-kernel averages of raw rocprofv3 --pmc counters (+ duration from the kernel trace of the same run).
usage: python tools/pmc_report.py <dir> <kernel-name regex> [min_duration_us]
For every counter: average over the matching dispatches; with GRBM_GUI_ACTIVE and the trace: effective clock =
GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back)."""
import csv, glob, os, re, sys
d, rx = sys.argv[1], re.compile(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
vals, durs = {}, []
seen = set()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if not rx.search(r["Kernel_Name"]):
            continue
        t = dur.get(r["Dispatch_Id"])
        if t is not None and t < min_us:
            continue
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen and t is not None:
            seen.add(r["Dispatch_Id"]); durs.append(t)
avg = {k: sum(v) / len(v) for k, v in vals.items()}
t = sum(durs) / len(durs) if durs else None
print(f"kernel /{sys.argv[2]}/: {len(durs)} dispatches, avg duration {t} us")
for k in sorted(avg):
    print(f"  {k:32s} {avg[k]:.6g}")
if t and "GRBM_GUI_ACTIVE" in avg:
    clk = avg["GRBM_GUI_ACTIVE"] / 8 / (t * 1e-6) / 1e9
    print(f"  effective clock {clk:.3f} GHz")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        print(f"  mfma busy frac {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * avg['GRBM_GUI_ACTIVE'] / 8):.3f}")
if "SQ_WAVE_CYCLES" in avg:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM"):
        if k in avg:
            print(f"  {k} / SQ_WAVE_CYCLES = {avg[k] / avg['SQ_WAVE_CYCLES']:.3f}")
if "FETCH_SIZE" in avg:
    print(f"  FETCH_SIZE x2 (gfx950 wide-read correction) = {avg['FETCH_SIZE'] * 2 * 1024 / 1e9:.3f} GB (if unit KB)")
if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
    print(f"  L2 hit rate {avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.3f}")
