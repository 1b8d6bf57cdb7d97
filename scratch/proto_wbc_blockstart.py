"""How good a starting working set for the WBC relaxation QP are "the six equalities + the inequality rows violated at z = 0" (Fr_des from the
MPC on the mu = 0.45 cone, the WBC's cone at mu = 0.4)?  CPU, the oracle's QP laid open; forces from the oracle's MPC."""
import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from conftest import load_pkg
import oracle_py as O
O.build()
pkg = load_pkg()
n, h = 256, 10
b = pkg.make_batch(n, h, "a1", seed=0xA1 + 2)
f, tau, st, sec, prev = O.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=8)
same = sub = sup = other = 0; ng = []; nf = []; extra_changes = []
for i in range(n):
    cmd = b["wbc_cmd"][i].copy(); cmd[51:63] = f[i]
    qp = O.wbc_qp(pkg.model_desc("a1"), b["fb_state"][i], cmd, b["prev_ori_vel"][i], dtype=np.float64)
    s0 = qp["ci0"]                                  # slack at z = 0
    sz = qp["CI"].T @ qp["z"] + qp["ci0"]           # slack at the solution
    guess = set(np.where(s0 < -1e-9)[0].tolist()); final = set(np.where(np.abs(sz) < 1e-7)[0].tolist())
    ng.append(len(guess)); nf.append(len(final))
    if guess == final: same += 1
    elif guess < final: sub += 1
    elif guess > final: sup += 1
    else: other += 1
    extra_changes.append(len(final - guess) + len(guess - final))
print("robots %d: guess == final %d, guess a subset %d, a superset %d, neither %d" % (n, same, sub, sup, other))
print("rows: guess mean %.2f max %d, final mean %.2f max %d; changes needed from the guess: mean %.2f max %d (from the equalities alone: mean %.2f)" % (
    np.mean(ng), max(ng), np.mean(nf), max(nf), np.mean(extra_changes), max(extra_changes), np.mean(nf)))
