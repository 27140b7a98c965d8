"""Turns rocprofv3 --pmc CSVs (one pass with FETCH_SIZE, one with WRITE_SIZE) into per-launch HBM
bytes for the dominant kernels, with the gfx950 corrections of MI355X_MICROARCH.md §HBM:
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced streaming read on gfx950 -> doubled; WRITE_SIZE is exact for 16-B-per-lane
stores (narrower stores are uncalibrated and flagged).
Kernel names are matched by REGEX on the demangled symbol (template-argument spelling changes between
builds); a label whose pattern matches nothing is listed under "missing" and makes the script exit 3 unless
--allow-missing is given. The output records the source digest of the profiled libclipmi.so ("lib_digest"):
bench.py reports `traffic` only from a summary whose digest equals the library it runs.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [--allow-missing label,label]"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LABELS = (
    ("scan", r"scan_topk_f32_kernel<512, false"),
    ("scan_coarse", r"scan_coarse_kernel<512, 4, false, false, false>"),
    ("scan_coarse_i8", r"scan_coarse_kernel<512, 4, false, true, false>"),          # the last segment (75 % of the rows)
    ("scan_coarse_i8_pre", r"scan_coarse_kernel<512, 4, true, true, false>"),       # the two earlier segments of the same pass
    ("scan_wide", r"scan_coarse_wide2_kernel<4, 2>"),                        # wide pass (Q = 1024: the second form): all its segments averaged
    ("rescore", r"rescore_pairs_kernel"),
    ("gemm_c_fc", r"gemm256p_bf16_nt_kernel<(1|6), ?false>"),
    ("gemm_qkv", r"gemm256p_bf16_nt_kernel<(0|5), ?false>"),
    ("gemm_resid", r"gemm256p_bf16_nt_kernel<(2|7), ?false>"),
    ("gemm128_c_fc", r"gemm_bf16_nt_kernel<(1|6)>"),
    ("gemm128_resid", r"gemm_bf16_nt_kernel<(2|7)>"),
    ("attention", r"attention"),
    ("layernorm", r"layernorm_kernel"),
    ("split_stats", r"split_stats_kernel"),
)
OPTIONAL = {"scan", "scan_coarse", "scan_coarse_i8_pre", "scan_wide", "gemm128_c_fc", "gemm128_resid", "layernorm", "split_stats"}   # not on every bench path


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def lib_digest():
    try:
        return json.load(open(os.path.join(ROOT, "cli-p_amd", "libclipmi.so.stamp")))["digest"]
    except Exception:
        return None


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    allow = set(OPTIONAL)
    for i, a in enumerate(sys.argv):
        if a == "--allow-missing" and i + 1 < len(sys.argv):
            allow |= set(sys.argv[i + 1].split(","))
            args.remove(sys.argv[i + 1])
    fetch = per_kernel(args[0], "FETCH_SIZE")
    write = per_kernel(args[1], "WRITE_SIZE")
    res = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = KiB*1024; "
                      "FETCH_SIZE doubled (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section)",
           "lib_digest": lib_digest(), "missing": [], "step_total": {}}

    def avg(d, pat):
        rx = re.compile(pat)
        xs = [v for k, vs in d.items() if rx.search(k) for v in vs]
        return (sum(xs) / len(xs), len(xs)) if xs else (None, 0)
    for label, pat in LABELS:
        (fk, nf), (wk, nw) = avg(fetch, pat), avg(write, pat)
        if fk is None and wk is None:
            res["missing"].append(label)
            continue
        rd = (fk or 0.0) * 1024 * 2
        wr = (wk or 0.0) * 1024
        res[f"{label}_read_bytes_per_launch"] = rd
        res[f"{label}_write_bytes_per_launch"] = wr
        res[f"{label}_bytes_per_launch"] = rd + wr
        res[f"{label}_launches_seen"] = max(nf, nw)
    # one coarse PASS streams the copy once in three launches: the last segment + two earlier ones
    if "scan_coarse_i8_bytes_per_launch" in res and "scan_coarse_i8_pre_bytes_per_launch" in res:
        res["scan_coarse_i8_bytes_per_pass"] = res["scan_coarse_i8_bytes_per_launch"] + 2 * res["scan_coarse_i8_pre_bytes_per_launch"]
    # all clipmi encode kernels summed: counter bytes per launch x launches, for the per-step traffic figure
    tot_r = sum(sum(vs) for k, vs in fetch.items() if "clipmi::" in k and "scan" not in k and "rescore" not in k
                and "select" not in k and "quantize_rows_i8" not in k and "coarse" not in k) * 1024 * 2
    tot_w = sum(sum(vs) for k, vs in write.items() if "clipmi::" in k and "scan" not in k and "rescore" not in k
                and "select" not in k and "quantize_rows_i8" not in k and "coarse" not in k) * 1024
    res["step_total"] = {"encode_kernels_read_bytes_all_launches": tot_r, "encode_kernels_write_bytes_all_launches": tot_w,
                         "note": "divide by the number of encode steps the profiled command ran (warmup + steps + probe reps)"}
    json.dump(res, open(args[2], "w"), indent=1)
    print(json.dumps(res, indent=1))
    hard = [m for m in res["missing"] if m not in allow]
    if hard:
        print("pmc_traffic: no kernel matched: " + ", ".join(hard), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
