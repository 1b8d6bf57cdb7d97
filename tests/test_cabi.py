"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/qrgpu.h declares.
No compute call is made here (there is no GPU); creating a context must fail loudly, not fall back."""
import ctypes as C
import os
import re

import pytest


def _declared_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "qrgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qrgpu_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_header(pkg):
    so = pkg._build.build()
    assert os.path.exists(so)
    lib = C.CDLL(so)
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libqrgpu.so does not export %s" % s
    for s in pkg.qrgpu.EXPORTS:
        assert s in syms


def test_kernels_are_gfx950_code_objects(pkg):
    so = pkg._build.build()
    data = open(so, "rb").read()
    assert b"gfx950" in data and b"qr_mpc_kernel" in data and b"qr_wbc_kernel" in data


def test_no_cpu_fallback(pkg, monkeypatch):
    """Without the library the package raises; without a gfx950 device qrgpu_create reports NO_DEVICE."""
    monkeypatch.setattr(pkg.qrgpu, "_LIB", None)
    monkeypatch.setattr(pkg.qrgpu, "lib_path", lambda: "/nonexistent/libqrgpu.so")
    with pytest.raises(pkg.MissingExtension):
        pkg.qrgpu.load_library()
    monkeypatch.undo()
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(pkg.QrgpuError):
            pkg.Context(device_id=0, max_batch=4)


def test_model_desc_defaults(pkg):
    lib = pkg.load_library()
    d = pkg.model_desc_struct()
    lib.qrgpu_model_desc_default(C.byref(d))
    assert abs(d.hip_l - 0.08505) < 1e-7 and d.kp_foot == 500.0 and abs(d.mu - 0.4) < 1e-7 and abs(d.weight_fb - 0.1) < 1e-7
