"""-m gpu: the HIP path against the committed golden fixtures (reference solver outputs + oracle outputs)."""
import numpy as np
import pytest

import golden_io
import gpu_helpers as G

pytestmark = pytest.mark.gpu


def test_mpc_golden(gpu_ctx, pkg):
    rows = golden_io.load("mpc_golden.npz")
    groups = {}
    for r in rows:
        groups.setdefault((str(r["robot"]), int(r["h"])), []).append(r)
    for (robot, h), rs in groups.items():
        gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), h)
        gpu_ctx.wbc_setup_packed(0, pkg.model_desc(robot))
        n = len(rs)
        fb = np.zeros((n, 37), np.float32)
        fb[:, 0:4] = [r["quat"] for r in rs]; fb[:, 13:25] = [r["q"] for r in rs]
        b = dict(n=n, horizon=h, mpc_state=np.stack([r["mpc_state"] for r in rs]), traj=np.stack([r["traj"] for r in rs]),
                 gait=np.stack([r["gait"] for r in rs]), fb_state=fb)
        out = G.run_mpc(gpu_ctx, pkg, b)
        assert np.all((out["status"] & 0xff) == 0), (robot, h, out["status"] & 0xff)
        for i, r in enumerate(rs):
            scale = max(1.0, np.abs(r["f_qpoases_sym"]).max())
            # reference qpOASES on the symmetric data, converged: same unique optimum
            assert np.abs(out["force"][i] - r["f_qpoases_sym"]).max() <= 2e-6 * scale, (robot, h, i)
            assert np.abs(out["force"][i] - r["f_oracle"]).max() <= 2e-6 * scale
            assert np.all(np.abs(out["tau"][i] - r["tau_oracle"]) <= G.tau_tol(r["tau_oracle"], 1e-5))


def test_wbc_golden(gpu_ctx, pkg):
    rows = golden_io.load("wbc_golden.npz")
    for robot in ("a1", "lite3"):
        rs = [r for r in rows if str(r["robot"]) == robot]
        gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg(robot), 10)
        gpu_ctx.wbc_setup_packed(0, pkg.model_desc(robot))
        n = len(rs)
        b = dict(n=n, horizon=10, fb_state=np.stack([r["fb_state"] for r in rs]), wbc_cmd=np.stack([r["wbc_cmd"] for r in rs]),
                 prev_ori_vel=np.stack([r["prev"] for r in rs]))
        out = G.run_wbc(gpu_ctx, pkg, b)
        assert np.all(out["status"] == 0)
        dbg = G.run_fb_debug(gpu_ctx, pkg, b)
        for i, r in enumerate(rs):
            assert np.all(np.abs(out["tau"][i] - r["tau64"]) <= G.tau_tol(r["tau64"], 1e-6)), (robot, i, np.abs(out["tau"][i] - r["tau64"]).max())
            assert np.all(np.abs(out["tau"][i] - r["tau32"]) <= G.tau_tol(r["tau32"], 1e-4))
            assert np.abs(out["qdes"][i] - r["qdes64"]).max() <= 1e-5
            assert np.abs(dbg["H"][i] - r["H"]).max() <= 2e-6 and np.abs(dbg["Jc"][i] - r["Jc"]).max() <= 1e-6
    G.setup_a1(gpu_ctx, pkg, 10)
