import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg
import oracle_py as O
import gpu_helpers as G
pkg = load_pkg()
ctx = pkg.Context(0, 4096, 16)
G.setup_a1(ctx, pkg, 10)
cfg = pkg.mpc_cfg("a1")
b = pkg.make_batch(4, 10, "a1", seed=3, frac_all_stance=1.0, frac_three_leg=0.0)
Hg, gg = G.run_assemble(ctx, pkg, b)
for i in range(2):
    Ho, go, ub = O.mpc_assemble(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
    d = Hg[i].view(np.uint32) != Ho.view(np.uint32)
    print("robot", i, "H entries differing:", d.sum(), "of", d.size, "max rel", np.abs(Hg[i]-Ho).max()/np.abs(Ho).max(), "g diff", (gg[i].view(np.uint32) != go.view(np.uint32)).sum())
    idx = np.argwhere(d)[:10]
    for a, c in idx: print("   ", a, c, Hg[i][a, c], Ho[a, c], "blockrow", a//12, "blockcol", c//12, "ta", a%12, "tb", c%12)
