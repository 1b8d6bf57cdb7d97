"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/qrgpu.h declares.
No compute call is made here (there is no GPU); creating a context must fail loudly, not fall back."""
import ctypes as C
import os
import re

import numpy as np

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


def test_all_kernels_present(pkg):
    """Every kernel of the path and of the SURVEY 8f rows is in the gfx950 code object."""
    data = open(pkg._build.build(), "rb").read()
    for k in (b"qr_mpc_kernel", b"qr_wbc_kernel", b"qr_vmc_kernel", b"qr_frontend_kernel", b"qr_estimator_kernel", b"qr_pack_state_kernel",
              b"qr_swing_kernel", b"qr_gait_kernel", b"qr_foothold_kernel", b"qr_lpt_order_kernel", b"qr_ground_kernel", b"qr_walk_gait_kernel", b"qr_swing_velocity_kernel"):
        assert k in data, k


def test_desc_defaults(pkg):
    lib = pkg.load_library()
    v = pkg.qrgpu.vmc_desc_struct(); lib.qrgpu_vmc_desc_default(C.byref(v))
    assert v.mass == 13.0 and abs(v.reg_weight - 1e-4) < 1e-10 and v.fmax_ratio == 10.0 and list(v.acc_weight) == [1, 1, 1, 10, 10, 1]
    e = pkg.qrgpu.estimator_desc_struct(); lib.qrgpu_estimator_desc_default(C.byref(e))
    assert e.window == 120 and abs(e.time_step - 0.002) < 1e-9 and abs(e.hip_offset[0] - 0.1805) < 1e-7
    assert lib.qrgpu_estimator_state_doubles(120) == 96 + 360 and lib.qrgpu_estimator_state_doubles(0) == 0
    f = pkg.qrgpu.foothold_desc_struct(); lib.qrgpu_foothold_desc_default(C.byref(f))
    assert abs(f.swing_kp[1] - 0.16) < 1e-7 and abs(f.foot_clearance - 0.01) < 1e-9 and abs(f.default_hip_position[1] + 0.135) < 1e-7
    assert np.allclose(np.array(f.hip_offset[:] + f.default_hip_position[:] + [f.hip_l] + f.swing_kp[:] + [f.foot_clearance], np.float32), pkg.workload.foothold_cfg("a1"))
    w = pkg.qrgpu.walk_gait_desc_struct(); lib.qrgpu_walk_gait_desc_default(C.byref(w))
    assert np.allclose(np.array(w.stance_duration[:] + w.duty_factor[:] + w.initial_leg_phase[:] + [float(x) for x in w.initial_leg_state[:]]
                                + [w.contact_detection_phase_threshold, float(w.n_states)] + [float(x) for x in w.state_switch[:]] + w.state_ratio[:], np.float32),
                       pkg.workload.walk_cfg())


@pytest.mark.gpu
def test_error_returns(gpu_ctx, pkg):
    """The reference has no error channel on this path; the C ABI returns a code from every call (error behaviour)."""
    import numpy as np
    ctx = pkg.Context(device_id=0, max_batch=8, horizon_max=16)
    try:
        a = ctx.alloc((28, 8)); t = ctx.alloc((120, 8)); g = ctx.alloc((40, 8)); f = ctx.alloc((12, 8))
        with pytest.raises(pkg.QrgpuError, match="NOT_SETUP"):
            ctx.mpc_solve_batch(8, a, t, g, None, f, None, None)
        ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), 10)
        with pytest.raises(pkg.QrgpuError, match="BAD_ARG"):
            ctx.mpc_solve_batch(9, a, t, g, None, f, None, None)              # n > max_batch
        with pytest.raises(pkg.QrgpuError, match="BAD_ARG"):
            ctx.mpc_solve_batch(8, a, t, g, None, None, None, None)           # no output array
        with pytest.raises(pkg.QrgpuError, match="BAD_ARG"):
            ctx.mpc_solve_batch(8, a, t, g, None, f, f, None)                 # torques requested without joint angles
        with pytest.raises(pkg.QrgpuError, match="NOT_SETUP"):
            ctx.vmc_force_batch(8, a, None, f)
        with pytest.raises(pkg.QrgpuError, match="BAD_ARG"):
            ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), 17)                    # horizon beyond K_MAX_GAIT_SEGMENTS
        # the compute stream: a non-blocking stream of the context's own, a different one per context; set_stream(None) = the default stream
        other = pkg.Context(device_id=0, max_batch=8, horizon_max=16)
        try:
            assert ctx.get_stream() != 0 and other.get_stream() != 0 and ctx.get_stream() != other.get_stream()
            own = other.get_stream()
            other.set_stream(None); assert other.get_stream() == 0
            other.set_stream(own); assert other.get_stream() == own
        finally:
            other.close()
        for v in (a, t, g, f):
            v.free()
    finally:
        ctx.close()


def test_supported_environment_switches_are_listed_in_the_header():
    """include/qrgpu.h carries the table of supported environment switches; csrc/qrgpu_ctx.h the list the library checks the environment against
    (anything else named QRGPU_* is reported once by qrgpu_create).  Every getenv of the host code names a supported switch -- laboratory
    switches go through lab_env and answer only under QRGPU_LAB=1 -- and every supported switch is in the header's table."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctxh = open(os.path.join(root, "quadruped-robot_amd", "csrc", "qrgpu_ctx.h")).read()
    header = open(os.path.join(root, "include", "qrgpu.h")).read()
    sup = re.search(r"#define QRGPU_SUPPORTED_ENV (.*?)\n#define QRGPU_LAB_ENV (.*?)\n", ctxh, re.S)
    supported = set(re.findall(r'"(QRGPU_\w+)"', sup.group(1)))
    lab = set(re.findall(r'"(QRGPU_\w+)"', sup.group(2)))
    assert supported and lab and not (supported & lab)
    for name in supported:
        assert re.search(r"^ \*   %s\b" % name, header, re.M), name
    for f in ("qrgpu_api.hip", "qrgpu_comm.hip"):
        src = open(os.path.join(root, "quadruped-robot_amd", "csrc", f)).read()
        assert set(re.findall(r'[^_]getenv\("(QRGPU_\w+)"\)', src)) <= supported, f
        assert set(re.findall(r'lab_env\("(QRGPU_\w+)"\)', src)) <= lab | {"QRGPU_LAB"}, f
