"""CPU checks of the drop-in boundary: libclipmi.so loads without a GPU and exports every symbol
include/clipmi.h declares; the ctypes Tower mirrors the C struct; argument validation answers
without touching the device."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "clipmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clipmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(clipmi):
    L = clipmi._lib.lib()
    names = _header_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), f"libclipmi.so does not export {n}"
    assert sorted(clipmi._lib.SYMBOLS) == names
    assert L.clipmi_abi_version() == clipmi._lib.ABI_VERSION


def test_product_library_reads_no_environment_and_ships_no_lab_kernels(clipmi):
    """The product library imports no getenv (the CLIPMI_* A/B knobs exist in the -DCLIPMI_DEV build only) and does not
    contain the laboratory kernels that no product path selects (live-threshold scan, gemm2w); the development library,
    which tools/ and two child-process tests load with CLIPMI_DEV_LIB=1, has both."""
    lib = os.path.join(ROOT, "cli-p_amd", "libclipmi.so")
    dev = os.path.join(ROOT, "cli-p_amd", "libclipmi_dev.so")
    assert clipmi._lib.LIB_PATH == lib
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", lib], text=True)
    assert "getenv" not in undefined
    blob = open(lib, "rb").read()
    assert b"scan_coarse_live_kernel" not in blob and b"gemm2w_resid_ln_kernel" not in blob
    two_digit_scan = b"scan_coarse_kernelILi512ELi4ELb0ELb1ELb1E"        # <512, 4, false, int8, two query digits>: DESIGN 4.1h
    assert two_digit_scan not in blob
    assert os.path.exists(dev), "build() also builds libclipmi_dev.so"
    assert "getenv" in subprocess.check_output(["nm", "-D", "--undefined-only", dev], text=True)
    dblob = open(dev, "rb").read()
    assert b"scan_coarse_live_kernel" in dblob and two_digit_scan in dblob


def test_tower_struct_matches_header(clipmi, tmp_path):
    """sizeof/offsetof of the ctypes mirror equal the C compiler's view of the header."""
    src = tmp_path / "t.c"
    fields = [f[0] for f in clipmi._lib.Tower._fields_]
    body = "\n".join(f'printf("{f} %zu\\n", offsetof(clipmi_tower, {f}));' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "clipmi.h"\nint main(){'
                   'printf("sizeof %zu\\n", sizeof(clipmi_tower));' + body + 'return 0;}')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(out["sizeof"]) == C.sizeof(clipmi._lib.Tower)
    for f in fields:
        assert int(out[f]) == getattr(clipmi._lib.Tower, f).offset, f


def test_topk_argument_validation_without_gpu(clipmi):
    L = clipmi._lib.lib()
    assert L.clipmi_topk_ip_workspace_bytes(1000000, 512, 16, 51) > 0
    assert L.clipmi_topk_ip_workspace_bytes(1000, 500, 1, 10) == 0          # E not supported
    assert "unsupported" in clipmi._lib.last_error()
    assert L.clipmi_topk_ip_workspace_bytes(1000, 512, 1, 100000) == 0        # K too large
    rc = L.clipmi_topk_ip(None, clipmi._lib.BF16, 10, 512, None, 1, 5, 0, None, None, None, 0, None)
    assert rc == 4 and "db_dtype" in clipmi._lib.last_error()
    rc = L.clipmi_merge_topk(None, None, 2, 1, 5, None, None, None, 0, None)
    assert rc == 1
    # the int8 copy's buffers are sized by the library and CHECKED (no launch happens: argument validation only) -
    # a caller that sizes them as the first version-3 header described ((N32 + 32) x 2 floats, an [N][E] copy) is refused
    N = 1000
    N32 = (N + 31) // 32 * 32
    assert L.clipmi_i8_copy_bytes(N, 512) == N32 * 512 and L.clipmi_i8_meta_bytes(N) == ((N32 + 32) + (N32 // 32 + 1)) * 8 + (N32 + 32) * 4
    fake = C.c_void_p(256)
    rc = L.clipmi_quantize_rows_i8(fake, N, 512, None, fake, N * 512, fake, (N32 + 32) * 8, None)
    assert rc == 1 and "clipmi_i8_meta_bytes" in clipmi._lib.last_error()


def test_missing_library_fails_loudly(clipmi, monkeypatch):
    monkeypatch.setattr(clipmi._lib, "_lib", None)
    monkeypatch.setattr(clipmi._lib, "LIB_PATH", "/nonexistent/libclipmi.so")
    with pytest.raises(clipmi.ClipmiError, match="no CPU fallback"):
        clipmi._lib.lib()


def test_search_on_cpu_device_refuses(clipmi):
    import numpy as np
    idx = clipmi.IndexFlatIP(512, device="cpu")
    idx.add(np.zeros((3, 512), np.float32))
    with pytest.raises(clipmi.ClipmiError, match="no CPU fallback"):
        idx.search(np.zeros((1, 512), np.float32), 2)
