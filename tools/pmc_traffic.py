"""Turns rocprofv3 --pmc CSVs (one pass with FETCH_SIZE, one with WRITE_SIZE) into per-launch HBM
bytes for the two dominant kernels, with the gfx950 corrections of MI355X_MICROARCH.md §HBM:
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced streaming read on gfx950 -> doubled; WRITE_SIZE is exact for 16-B-per-lane
stores (narrower stores are uncalibrated and flagged).
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = KiB*1024; "
                      "FETCH_SIZE doubled (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section)"}

    def avg(d, key):
        xs = [v for k, vs in d.items() if key in k for v in vs]
        return sum(xs) / len(xs) if xs else None
    for label, key in (("scan", "scan_topk_f32_kernel<512, false"), ("scan_coarse", "scan_coarse_kernel<512, 4, false, false>"),
                       ("scan_coarse_i8", "scan_coarse_kernel<512, 4, false, true>"),
                       ("rescore", "rescore_pairs_kernel"), ("gemm_c_fc", "gemm256p_bf16_nt_kernel<1>"),
                       ("gemm_qkv", "gemm256p_bf16_nt_kernel<0>"), ("gemm_resid", "gemm256p_bf16_nt_kernel<2>"),
                       ("gemm128_c_fc", "gemm_bf16_nt_kernel<1>"), ("gemm128_resid", "gemm_bf16_nt_kernel<2>"),
                       ("attention", "attention"), ("layernorm", "layernorm_kernel")):
        fk, wk = avg(fetch, key), avg(write, key)
        if fk is None and wk is None:
            continue
        rd = (fk or 0.0) * 1024 * 2
        wr = (wk or 0.0) * 1024
        res[f"{label}_read_bytes_per_launch"] = rd
        res[f"{label}_write_bytes_per_launch"] = wr
        res[f"{label}_bytes_per_launch"] = rd + wr
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
