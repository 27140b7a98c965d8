#!/bin/bash
# The JPEG decode's kernels under rocprofv3 (kernel trace) on 870 files of the bench's kind and of photo-like content, on 435 files of
# 2000 x 1500, and the files -> vectors pipeline with and without the device decoder: writes gpurun_out/jpeg_kernels.txt
# and the traces tools/jpeg_profile_report.py turns into profiles/r05_jpeg_kernels.txt. usage: bash tools/gpu_jpeg_profile.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; cd "$ROOT"; mkdir -p gpurun_out
out=gpurun_out/jpeg_kernels.txt; : > $out
summarise() {   # $1 = rocprof directory, $2 = label
  python3 - "$1" "$2" >> $out <<'PY'
import csv, glob, os, re, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
by = {}
for r in csv.DictReader(open(f)):
    m = re.search(r"(jpeg_\w+_kernel|resize_\w_kernel)", r["Kernel_Name"])
    if m:
        by.setdefault(m.group(1), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"# {sys.argv[2]}: kernel, launches, median us, min us, max us")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k:28s} {len(v):4d} {v[len(v) // 2]:10.1f} {v[0]:10.1f} {v[-1]:10.1f}")
PY
}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/jpeg_prof_870 -- python3 tools/jpeg_probe.py 870 > gpurun_out/jpeg_probe_870.log 2>&1 || exit 1
grep -v "^[EW]2026" gpurun_out/jpeg_probe_870.log | grep -E "cases|files of" | sed 's/^/# tools\/jpeg_probe.py 870: /' >> $out
summarise gpurun_out/jpeg_prof_870 "870 files of 224 x 224 per launch: the first 8 launches of each kernel are the noise files (quality 95, ~58 KB), the last 7 the photo-like files (quality 85, ~19 KB); the first launch of all is the 146-file parity batch"
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/jpeg_prof_2k -- python3 tools/files_to_vectors.py photo2k > gpurun_out/f2v_photo2k.log 2>&1 || exit 1
grep "images/s" gpurun_out/f2v_photo2k.log | sed 's/^/# tools\/files_to_vectors.py photo2k (under rocprofv3): /' >> $out
summarise gpurun_out/jpeg_prof_2k "435 files of 2000 x 1500 (1.16 MB each) per launch, in the pipeline"
for k in noise photo; do
  timeout -k 10 500 python3 tools/files_to_vectors.py $k > gpurun_out/f2v_$k.log 2>&1 || exit 1
  grep "images/s" gpurun_out/f2v_$k.log | sed 's/^/# tools\/files_to_vectors.py: /' >> $out
done
cat $out
