"""Per-kernel MFMA-busy fraction from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
SQ_WAVE_CYCLES (SQ block) and GRBM_GUI_ACTIVE (GRBM block) — the MfmaUtil evidence SURVEY.md §8(d) names.

gfx950 has no derived-metric section in ROCm 7.2 (MI355X_MICROARCH.md, rocprofv3 PMC slots), so the fraction is
built from raw counters and CALIBRATED against the one thing known exactly: the number of MFMA instructions a GEMM
launch issues. For gemm256p c_fc (M x 3072 x 768): M_pad/16 * 3072/16 * 768/32 instructions of
v_mfma_f32_16x16x32_bf16, 16 cycles each on one SIMD (cycle table). If the counter is summed over SIMDs in cycles,
SQ_VALU_MFMA_BUSY_CYCLES ~= 16 * N_mfma; the script prints the measured ratio so a different unit shows up at once.
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
(GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.)
usage: python tools/pmc_mfma.py <pmc_dir> <out.json> [rows_M_of_c_fc]"""
import csv
import glob
import json
import os
import re
import sys

LABELS = (
    ("gemm_c_fc", r"gemm256p_bf16_nt_kernel<(1|6), ?false>"),
    ("gemm_qkv", r"gemm256p_bf16_nt_kernel<(0|5), ?false>"),
    ("gemm_resid", r"gemm256p_bf16_nt_kernel<(2|7), ?false>"),
    ("gemm_patch", r"gemm256_bf16_nt_kernel<4>"),
    ("scan_coarse_i8", r"scan_coarse_kernel<512, 4, false, true, false>"),
    ("scan_wide_i8", r"scan_coarse_wide2_kernel<4, 2>"),
    ("scan_f32", r"scan_topk_f32_kernel<512, false"),
    ("attention", r"attention"),
)


def main():
    d, out = sys.argv[1], sys.argv[2]
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 43500
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            vals.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    res = {"_method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE (own pass, "
                      "--kernel-trace only); mfma_busy_frac = MFMA_BUSY / (1024 * GRBM_GUI_ACTIVE / 8)"}
    for label, pat in LABELS:
        rx = re.compile(pat)

        def avg(counter):
            xs = [v for (k, c), vs in vals.items() if c == counter and rx.search(k) for v in vs]
            return sum(xs) / len(xs) if xs else None
        mf, gui, busy, wc = avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("GRBM_GUI_ACTIVE"), avg("SQ_BUSY_CYCLES"), avg("SQ_WAVE_CYCLES")
        if mf is None or not gui:
            continue
        res[label] = {"SQ_VALU_MFMA_BUSY_CYCLES": mf, "GRBM_GUI_ACTIVE": gui, "SQ_BUSY_CYCLES": busy, "SQ_WAVE_CYCLES": wc,
                      "mfma_busy_frac": mf / (1024.0 * gui / 8.0)}
    if "gemm_c_fc" in res:
        mp = (M + 255) // 256 * 256
        n_mfma = (mp // 16) * (3072 // 16) * (768 // 32)
        res["calibration"] = {"kernel": "gemm_c_fc", "rows": M, "mfma_instructions": n_mfma,
                              "expected_busy_cycles_at_16_per_instruction": 16 * n_mfma,
                              "measured_over_expected": res["gemm_c_fc"]["SQ_VALU_MFMA_BUSY_CYCLES"] / (16.0 * n_mfma)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
