"""GPU: the committed golden tick log (inputs + CPU-oracle outputs) replayed through the batched tick, stateless and stateful;
a log recorded from the library replays onto itself bit for bit; the CLI."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ticklog_a1_h10.qrtl")


def test_replay_golden_log(pkg):
    r = pkg.ticklog.TickLogReader(GOLDEN)
    ctx = pkg.Context(device_id=0, max_batch=r.n_robots, horizon_max=16)
    try:
        pkg.replay.setup_from_log(ctx, r)
        for stateful in (False, True):
            res = pkg.replay.replay(ctx, r, stateful=stateful)
            assert res["robot_ticks"] == 24 and res["flagged"] == 0 and res["recorded_flagged"] == 0
            assert res["worst_force"] <= 1e-5 and res["worst_tau"] <= 1e-4, (stateful, res["worst_force"], res["worst_tau"])    # north_star tolerance
    finally:
        ctx.close()


def test_record_then_replay_is_bit_identical(pkg, tmp_path):
    n, ticks, h = 16, 5, 10
    ctx = pkg.Context(device_id=0, max_batch=n, horizon_max=16)
    try:
        ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
        stream = pkg.make_batch(n * ticks, h, "a1", seed=77)
        keys = ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel")
        batches = [dict({k: stream[k][t * n:(t + 1) * n] for k in keys}, n=n, horizon=h) for t in range(ticks)]
        p = str(tmp_path / "rec.qrtl")
        pkg.replay.record(ctx, p, batches, pkg.mpc_cfg("a1"), pkg.model_desc("a1"), "a1")
        r = pkg.ticklog.TickLogReader(p)
        assert (r.ticks, r.n_robots) == (ticks, n)
        assert np.abs(r.tick(ticks - 1)["prev_ori_vel"]).max() > 0
        for stateful in (False, True):
            res = pkg.replay.replay(ctx, r, stateful=stateful)
            assert res["worst_force"] == 0.0 and res["worst_tau"] == 0.0 and res["flagged"] == res["recorded_flagged"]
    finally:
        ctx.close()


def test_cli_replay():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "qr_replay.py"), "replay", GOLDEN, "--stateful"], capture_output=True, text=True)
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr
