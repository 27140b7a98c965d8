"""Build recipe for libclipmi.so: every HIP translation unit under csrc/ for gfx950, in-tree.

`python cli-p_amd/build.py` or `__graft_entry__.build()`. hipcc cross-compiles without a GPU.
The .so is git-ignored but travels with the tree to the GPU box (it is not gpurun-ignored).

Staleness is decided by CONTENT, not mtimes: each object file has a stamp holding the SHA-256 of its
source, every header and the compiler flags; the library has one over the object stamps. A stamp that
does not match (edited source, changed flags, a .so copied in from elsewhere) recompiles.
`CLIPMI_FORCE_BUILD=1` (or force=True / `--force`) recompiles everything. build() returns a report
dict — how many translation units were compiled vs reused — that `__graft_entry__.build()` prints.
"""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]


class Variant:
    """One library built from csrc/: the product (libclipmi.so) or the development build (libclipmi_dev.so, -DCLIPMI_DEV:
    the CLIPMI_* A/B environment knobs and the laboratory kernels that no product path selects - tools/ and two
    child-process tests load it by setting CLIPMI_DEV_LIB=1; the product library reads no environment variable)."""

    def __init__(self, dev):
        self.dev = dev
        self.obj = os.path.join(CSRC, "_obj_dev" if dev else "_obj")
        self.lib = os.path.join(HERE, "libclipmi_dev.so" if dev else "libclipmi.so")
        self.stamp = self.lib + ".stamp"
        # CLIPMI_EXTRA_CXXFLAGS (development library only): e.g. -DCLIPMI_GEMM_STAMPS=1 for tools/gp_stamps.py; the flags
        # are part of the content stamp, so the same variable must be set when that build is loaded
        self.flags = BASE_FLAGS + (["-DCLIPMI_DEV"] + os.environ.get("CLIPMI_EXTRA_CXXFLAGS", "").split() if dev else [])


PRODUCT, DEV = Variant(False), Variant(True)
OBJ, LIB, STAMP, FLAGS = PRODUCT.obj, PRODUCT.lib, PRODUCT.stamp, PRODUCT.flags


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _sha(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def source_digest(variant=PRODUCT):
    """Digest of everything the library is made from (sources, headers, flags)."""
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    headers.append(os.path.join(HERE, "..", "include", "clipmi.h"))
    srcs = [os.path.join(CSRC, s) for s in _sources()]
    return _sha(srcs + headers, variant.flags + LINK_FLAGS)


def is_current(variant=PRODUCT):
    """True when the library exists and its stamp matches the sources in the tree."""
    st = _read(variant.stamp)
    if st is None or not os.path.exists(variant.lib):
        return False
    try:
        return json.loads(st).get("digest") == source_digest(variant)
    except ValueError:
        return False


def build(force=False, verbose=True, dev=False):
    variant = DEV if dev else PRODUCT
    OBJ, LIB, STAMP, FLAGS = variant.obj, variant.lib, variant.stamp, variant.flags
    force = force or os.environ.get("CLIPMI_FORCE_BUILD", "") not in ("", "0")
    os.makedirs(OBJ, exist_ok=True)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    headers.append(os.path.join(HERE, "..", "include", "clipmi.h"))
    jobs, objs, stamps = [], [], {}
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        digest = _sha([s] + headers, FLAGS)
        objs.append(o)
        stamps[o] = digest
        if force or not os.path.exists(o) or _read(o + ".stamp") != digest:
            jobs.append((src, [HIPCC] + FLAGS + ["-c", s, "-o", o], o, digest))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    def compile_one(job):
        _, cmd, o, digest = job
        if os.path.exists(o + ".stamp"):
            os.remove(o + ".stamp")
        run(cmd)
        with open(o + ".stamp", "w") as f:
            f.write(digest + "\n")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    digest = source_digest(variant)
    linked = False
    if force or jobs or not is_current(variant):
        if os.path.exists(STAMP):
            os.remove(STAMP)
        run([HIPCC] + LINK_FLAGS + objs + ["-o", LIB])
        # a shared library links with undefined symbols: load it once (hipcc can drop a kernel template's host stub without a
        # diagnostic - DESIGN 4.1i - and the first sign would be an OSError at import time on the GPU box)
        import ctypes
        try:
            ctypes.CDLL(LIB)
        except OSError as e:
            raise RuntimeError(f"{LIB} does not load: {e}") from None
        with open(STAMP, "w") as f:
            json.dump({"digest": digest, "objects": {os.path.basename(o): d for o, d in stamps.items()}}, f)
        linked = True
    return {"lib": LIB, "units": len(objs), "compiled": [j[0] for j in jobs], "reused": len(objs) - len(jobs),
            "linked": linked, "forced": bool(force), "digest": digest}


if __name__ == "__main__":
    rep = build(force="--force" in sys.argv, dev="--dev" in sys.argv)
    print(f"{rep['lib']}: compiled {len(rep['compiled'])} of {rep['units']} translation units "
          f"({', '.join(rep['compiled']) or 'none'}), linked={rep['linked']}, digest {rep['digest'][:16]}")
