"""Build recipe for libclipmi.so: every HIP translation unit under csrc/ for gfx950, in-tree.

`python cli-p_amd/build.py` or `__graft_entry__.build()`. hipcc cross-compiles without a GPU.
The .so is git-ignored but travels with the tree to the GPU box (it is not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libclipmi.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "clipmi.h"))
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
