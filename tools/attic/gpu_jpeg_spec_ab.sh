ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
for spec in 3 6; do
  sed -i "s/^constexpr int JP_SPEC = [0-9]*;/constexpr int JP_SPEC = $spec;/" cli-p_amd/csrc/jpeg.hip
  python3 cli-p_amd/build.py > gpurun_out/spec_build_$spec.log 2>&1 || { tail -3 gpurun_out/spec_build_$spec.log; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/spec_prof_$spec -- python3 tools/files_to_vectors.py photo2k > gpurun_out/spec_f2v_$spec.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/spec_prof870_$spec -- python3 tools/jpeg_probe.py 870 > gpurun_out/spec_probe_$spec.log 2>&1
  python3 - $spec <<'PY'
import csv, glob, os, sys
for tag, sl in (("photo2k", slice(None)), ("870 x 224", None)):
    d = f"gpurun_out/spec_prof_{sys.argv[1]}" if tag == "photo2k" else f"gpurun_out/spec_prof870_{sys.argv[1]}"
    f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "jpeg_huffman" in r["Kernel_Name"]]
    if tag == "photo2k":
        print("JP_SPEC", sys.argv[1], tag, "huffman median us", sorted(v)[len(v) // 2])
    else:
        a, b = sorted(v[1:8]), sorted(v[8:15]); print("JP_SPEC", sys.argv[1], tag, "noise", a[3], "photo-like", b[3])
PY
done
sed -i "s/^constexpr int JP_SPEC = [0-9]*;/constexpr int JP_SPEC = 3;/" cli-p_amd/csrc/jpeg.hip
