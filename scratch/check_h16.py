import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import oracle_py as O
import gpu_helpers as G
pkg = load_pkg()
ctx = pkg.Context(0, 8192, 16)
h = 16
ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h); ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
for excite in (0.3, 1.0):
    n = 256
    ba = pkg.make_batch(n // 2, h, "a1", seed=501, excite=excite); bl = pkg.make_batch(n // 2, h, "lite3", seed=502, excite=excite)
    b = dict(ba)
    for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
        b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
    b["n"] = n
    tid = pkg.shard.interleave_types(n, 2)
    t0 = time.time(); out = G.run_tick(ctx, pkg, b, type_id=tid); dt = time.time() - t0
    flags = out["status"] & 0xff
    print("excite", excite, "h16 mixed tick: status flags", {int(k): int((flags == k).sum()) for k in np.unique(flags)}, "iters mean %.1f max %d" % ((out["status"] >> 8).mean(), (out["status"] >> 8).max()))
    worst_f = 0; worst_t = 0; nchk = 0
    for i in range(0, n, 4):
        robot = "a1" if tid[i] == 0 else "lite3"
        u, st, rc = O.mpc_solve(pkg.mpc_cfg(robot), h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        if flags[i]: continue
        worst_f = max(worst_f, np.abs(out["force"][i] - u[:12]).max() / max(1, np.abs(u[:12]).max())); nchk += 1
        cmd = b["wbc_cmd"][i].copy(); cmd[51:63] = u[:12].astype(np.float32)
        w = O.wbc_run(pkg.model_desc(robot), b["fb_state"][i].astype(np.float64), cmd.astype(np.float64), dtype=np.float64)
        tau = O.mpc_force_to_torque(pkg.model_desc(robot)[:3], b["fb_state"][i, :4], b["fb_state"][i, 13:25], u[:12]).astype(np.float64)
        for l in range(4):
            if cmd[63 + l]: tau[3*l:3*l+3] = w["tau"][3*l:3*l+3]
        worst_t = max(worst_t, (np.abs(out["tau"][i] - tau) / np.maximum(1, np.abs(tau))).max())
    print("   checked", nchk, "robots: worst rel force err %.2e, worst tau err/max(1,|tau|) %.2e" % (worst_f, worst_t))
