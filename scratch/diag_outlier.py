import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py as O
O.build(); pkg = load_pkg()
n, h, robot = 4096, 10, "lite3"
ctx = pkg.Context(0, n, 16)
cfg, md = pkg.mpc_cfg(robot), pkg.model_desc(robot)
ctx.mpc_setup_packed(0, cfg, h); ctx.wbc_setup_packed(0, md)
b = pkg.make_batch(n, h, robot, seed=1310, excite=1.0)
o = G.run_tick(ctx, pkg, b)
f, tau, st, sec, prev = O.tick_batch(1, cfg, h, md[:3], md, b["mpc_state"], b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=32)
et = (np.abs(o["tau"] - tau) / np.maximum(1.0, np.abs(tau))).max(1)
for i in np.argsort(-et)[:4]:
    print("robot %d: torque err %.2e, force err %.2e abs (|f|max %.1f), iters %d, nls %d" % (i, et[i], np.abs(o["force"][i] - f[i]).max(), np.abs(f[i]).max(), o["status"][i] >> 8, int(b["gait"][i].sum())))
    # WBC on the GPU's forces through the fp64 oracle: isolates the WBC kernel from the propagation of the MPC difference
    cmd = b["wbc_cmd"][i].copy(); cmd[51:63] = o["force"][i]
    w = O.wbc_run(md, b["fb_state"][i].astype(np.float64), cmd.astype(np.float64), prev_ori_vel=b["prev_ori_vel"][i].astype(np.float64), dtype=np.float64)
    stance = b["wbc_cmd"][i, 63:67] != 0
    m = np.repeat(stance, 3)
    print("   WBC(fp64 oracle, GPU forces) vs GPU tau on stance legs: %.2e ; vs oracle-tick tau: %.2e" % (np.abs(w["tau"][m] - o["tau"][i][m]).max(), np.abs(w["tau"][m] - tau[i][m]).max()))
    u, s2, rc = O.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
    print("   oracle mpc: iters %d n_active %d ; first-step force diff GPU-oracle:" % (s2["iters"], s2["n_active"]), np.round(o["force"][i] - u[:12], 6))
