"""The header-only C++ adapters (include/qrgpu_adapters.hpp): they must compile against stand-ins of the
reference types (CPU), and, on a GPU box, drive libqrgpu.so to the oracle's numbers through the exact call
sequence of SolveDenseMPC / qrWbcLocomotionController::Run."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "stubs", "adapter_demo")


def _compile(pkg):
    so = pkg._build.build()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "stubs"),
           os.path.join(ROOT, "tests", "stubs", "adapter_demo.cpp"), "-o", EXE, so, "-Wl,-rpath," + os.path.dirname(so)]
    subprocess.check_call(cmd)
    return EXE


def test_adapters_compile_and_link(pkg):
    exe = _compile(pkg)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_adapters_drive_the_gpu_path(pkg, oracle):
    exe = _compile(pkg)
    b = pkg.make_batch(3, 10, "a1", seed=17)
    cfg = pkg.mpc_cfg("a1"); md = pkg.model_desc("a1")
    vin, vq = pkg.workload.make_vmc_batch(3, seed=23)
    _, _, wratio = pkg.workload.make_vmc_world_batch(3, seed=29)
    for i in range(3):
        vals = [10] + list(cfg) + list(b["mpc_state"][i]) + list(b["traj"][i]) + list(b["gait"][i]) + list(b["fb_state"][i]) + list(b["wbc_cmd"][i])
        vals += list(vin[i]) + list(wratio[i])
        inp = " ".join(repr(float(v)) if not isinstance(v, int) else str(v) for v in vals)
        out = subprocess.run([exe], input=inp.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert out.returncode == 0, out.stderr.decode()
        lines = {l.split()[0]: l.split()[1:] for l in out.stdout.decode().splitlines() if l and l.split()[0] in ("before", "force", "tau", "status", "vmcforce", "vmcstatus", "vmcwforce", "vmcwstatus")}
        assert float(lines["before"][0]) == 0.0 and int(lines["status"][0]) == 0
        f = np.array([float(x) for x in lines["force"]]); tau = np.array([float(x) for x in lines["tau"]])
        u, st, rc = oracle.mpc_solve(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert np.abs(f - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max())
        cmd = b["wbc_cmd"][i].copy(); cmd[51:63] = u[:12].astype(np.float32)
        w = oracle.wbc_run(md, b["fb_state"][i].astype(np.float64), cmd.astype(np.float64), dtype=np.float64)
        assert np.all(np.abs(tau - w["tau"]) <= 1e-5 * np.maximum(1.0, np.abs(w["tau"])))
        # force-balance adapter: ComputeContactForce through qrgpu_vmc_force1
        fv = np.array([float(x) for x in lines["vmcforce"]])
        fo, _, _, _, rc = oracle.vmc_solve(pkg.workload.vmc_cfg("a1"), md[:3], vin[i], vq[i])
        assert np.abs(fv - fo).max() <= 1e-5 * max(1.0, np.abs(fo).max())
        assert bool(int(lines["vmcstatus"][0]) & 0x80) == (rc == 1)
        # the world-frame adapter was called with the same Rcb slot: compare with the oracle on exactly that input
        xin = vin[i].copy(); xin[31:34] = (0, 0, 9.8); xin[34:37] = (0, 0, 1)
        fw = np.array([float(x) for x in lines["vmcwforce"]])
        fo, _, _, _, rc = oracle.vmc_solve(pkg.workload.vmc_cfg("a1"), md[:3], xin, vq[i], wratio[i])
        assert np.abs(fw - fo).max() <= 1e-5 * max(1.0, np.abs(fo).max())
        assert bool(int(lines["vmcwstatus"][0]) & 0x80) == (rc == 1)
