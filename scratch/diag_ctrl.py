import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import gpu_helpers as G
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py as O
O.build(); pkg = load_pkg()
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, 10)
b = pkg.make_batch(256, 10, "a1", seed=0xA2)
out = G.run_mpc(ctx, pkg, b)
st = out["status"]; fl = st & 0xff; it = st >> 8
print("flags:", np.unique(fl, return_counts=True), "iters max", it.max())
cfg = pkg.mpc_cfg("a1")
bad = np.where(fl != 0)[0]
print("bad robots", bad[:10], "nls", b["gait"][bad[:10]].sum(1), "iters", it[bad[:10]])
good = np.where(fl == 0)[0]
err = []
for i in good[:64]:
    u, s2, rc = O.mpc_solve(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
    err.append(np.abs(out["force"][i] - u[:12]).max() / max(1, np.abs(u[:12]).max()))
print("good robots: max rel force err %.2e" % max(err), "iters of oracle vs gpu on robot", good[0], it[good[0]])
for i in bad[:3]:
    u, s2, rc = O.mpc_solve(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
    print("bad", i, "oracle iters", s2, "gpu force", out["force"][i][:6], "oracle", u[:6])
