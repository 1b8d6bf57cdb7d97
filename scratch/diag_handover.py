import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py as O
O.build()
pkg = load_pkg()
ctx = pkg.Context(0, 4096, 16)
h = 10
G.setup_a1(ctx, pkg, h)
b = pkg.make_batch(96, h, "a1", seed=0xBEEF, excite=3.0, frac_all_stance=0.0, frac_three_leg=1.0)
ctx.set_rescue_pass(False)
out = G.run_mpc(ctx, pkg, b)
st = out["status"]
cfg = pkg.mpc_cfg("a1")
for i in range(96):
    u, s, rc = O.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
    err = np.abs(out["force"][i] - u[:12]).max() / max(1.0, np.abs(u[:12]).max())
    if (st[i] & 0xff) or err > 1e-5:
        print("robot %d: status %d iters %d | oracle iters %d n_active %d | force err %.2e" % (i, st[i] & 0xff, st[i] >> 8, s["iters"], s["n_active"], err))
print("flagged", int(((st & 0xff) != 0).sum()), "max iters", (st >> 8).max())
