"""The pathological robot of bench draw 5 at 128 robots (all stance, 80-130 changes a tick despite the warm start): how different are the optimal
working sets of consecutive ticks?  CPU, the oracle's MPC."""
import sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from conftest import load_pkg
import oracle_py as O
O.build(); pkg = load_pkg()
n, h, r = 128, 10, 97
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2 + 1000 * 5, steps=8)
cfg = pkg.mpc_cfg("a1")
A = O.mpc_constraint_matrix(h, 0.45)
prev = None
for k, b in enumerate(seq):
    H, g = O.mpc_assemble(cfg, h, b["mpc_state"][r], b["traj"][r], b["gait"][r])[:2]
    res = O.mpc_solve(cfg, h, b["mpc_state"][r], b["traj"][r], b["gait"][r])
    u = np.asarray(res[0] if isinstance(res, tuple) else res, np.float64).reshape(-1)[:12 * h]
    gait = b["gait"][r].reshape(h, 4)
    fmax = float(cfg[2])
    # rows of a stance leg-step: fz >= 0, fz <= fmax, |fx| <= mu fz, |fy| <= mu fz
    act = set()
    for s in range(h):
        for l in range(4):
            if gait[s, l] == 0: continue
            fx, fy, fz = u[12 * s + 3 * l: 12 * s + 3 * l + 3]
            tol = 1e-6 * max(1.0, fmax)
            if fz < tol: act.add((s, l, 'z0'))
            if fz > fmax - tol: act.add((s, l, 'zmax'))
            for nm, v in (('x+', 0.45 * fz - fx), ('x-', 0.45 * fz + fx), ('y+', 0.45 * fz - fy), ('y-', 0.45 * fz + fy)):
                if v < tol: act.add((s, l, nm))
    msg = "tick %d: active rows %d" % (k, len(act))
    if prev is not None:
        # the same set shifted by one horizon step (the gait table scrolls)
        same = len(act & prev); shifted = len(act & {(s - 1, l, t) for (s, l, t) in prev})
        msg += "; in common with the previous tick's set %d (as is) / %d (shifted one step)" % (same, shifted)
    print(msg, " iterations of the oracle (cold):", res[2] if isinstance(res, tuple) and len(res) > 2 else "?")
    prev = act
