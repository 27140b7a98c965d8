"""profiles/r05_jpeg_kernels.txt out of what tools/gpu_jpeg_profile.sh left under gpurun_out/ (kernel traces of tools/jpeg_probe.py and
tools/files_to_vectors.py photo2k, the files -> vectors lines): python3 tools/jpeg_profile_report.py [out]"""
import csv, glob, os, re, sys


def rows(d):
    f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
    by = {}
    for r in csv.DictReader(open(f)):
        m = re.search(r"(jpeg_\w+_kernel|resize_\w_kernel)", r["Kernel_Name"])
        if m:
            by.setdefault(m.group(1), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return by


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_jpeg_kernels.txt"
    lines = [l.rstrip("\n") for l in open("gpurun_out/jpeg_kernels.txt") if l.startswith("# tools/")]
    txt = ["# Round 5: csrc/jpeg.hip under rocprofv3 --kernel-trace (tools/gpu_jpeg_profile.sh), one MI355X; microseconds per launch", ""]
    txt += [l for l in lines if "jpeg_probe" in l]
    by = rows("gpurun_out/jpeg_prof_870")
    txt.append("# 870 files of 224 x 224 per launch. Launch 0 is the 146-file parity batch, launches 1-7 the noise files (quality 95, ~58 KB each: the")
    txt.append("# bench's kind - no end-of-block symbols, the serial case), launches 8-14 the photo-like files (quality 85, ~19 KB): median us")
    txt.append("# (tools/jpeg_probe.py removes the byte stuffing on the host: jpeg_unstuff_kernel returns at once here)")
    txt.append(f"{'kernel':28s} {'noise':>10s} {'photo-like':>12s}")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        a, b = sorted(v[1:8]), sorted(v[8:15])
        txt.append(f"{k:28s} {a[len(a) // 2]:10.1f} {b[len(b) // 2]:12.1f}")
    txt.append("")
    txt += [l for l in lines if "photo2k" in l]
    by = rows("gpurun_out/jpeg_prof_2k")
    txt.append("# 435 files of 2000 x 1500 (1.16 MB each) per launch, in the pipeline (byte stuffing removed on the device): median us")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        txt.append(f"{k:28s} {v[len(v) // 2]:10.1f}")
    txt.append("")
    txt.append("# files -> vectors (build-index.py's loop, 16 decode processes), Pillow in the workers (device_jpeg_kb 0) against the decode on the device;")
    txt.append("# per batch: wall, and the busy time of the three pipelined stages (worker processes | shared memory -> device | kernels incl. encode)")
    txt += [l for l in lines if "files_to_vectors.py:" in l]
    open(out, "w").write("\n".join(txt) + "\n")


if __name__ == "__main__":
    main()
