"""Emulate the device active set with the degenerate-row exclusion on one robot; report what is violated at termination."""
import sys
sys.path.insert(0, '/root/repo/scratch')
import numpy as np
from proto_pdas import reduced, rows, pkg, O

def gi_excl(M, x0, N, c0, verbose=False):
    m = N.shape[0]; act = []; u = []; excl = set(); it = 0; x = x0.copy()
    Si = np.zeros((0, 0))
    while True:
        s = N @ x + c0
        cand = [i for i in range(m) if i not in act and i not in excl and s[i] < -1e-9]
        if not cand:
            viol = [(i, s[i]) for i in range(m) if s[i] < -1e-9]
            return x, it, act, viol
        ip = min(cand, key=lambda i: s[i]); unew = 0.0
        while True:
            it += 1
            NA = N[act] if act else np.zeros((0, N.shape[1]))
            w = M @ N[ip]; d = NA @ w
            r = Si @ d if act else np.zeros(0)
            z = w - M @ NA.T @ r if act else w
            delta = N[ip] @ w; zc = delta - d @ r if act else delta
            tt = [u[k] / r[k] if r[k] > 0 else np.inf for k in range(len(act))]
            t1 = min(tt) if tt else np.inf
            have_z = zc > 1e-13 * delta
            t2 = -(N[ip] @ x + c0[ip]) / zc if have_z else np.inf
            t = min(t1, t2)
            if verbose: print(it, "ip", ip, divmod(ip, 6), "q", len(act), "zc/delta %.2e t1 %.2e t2 %.2e" % (zc / delta, t1, t2), "r range", (min(r) if len(r) else 0, max(r) if len(r) else 0))
            if not t < np.inf:
                excl.add(ip)
                if verbose: print("   EXCLUDED", ip, divmod(ip, 6), "slack", s[ip], "active on same leg-step:", [a % 6 for a in act if a // 6 == ip // 6])
                break
            if have_z: x = x + t * z
            u = [u[k] - t * r[k] for k in range(len(act))]; unew += t
            if have_z and t == t2:
                act.append(ip); u.append(unew); excl.clear()
                NA = N[act]; Si = np.linalg.inv(NA @ M @ NA.T); break
            l = int(np.argmin(tt)); act.pop(l); u.pop(l); excl.clear()
            NA = N[act] if act else np.zeros((0, N.shape[1])); Si = np.linalg.inv(NA @ M @ NA.T) if act else np.zeros((0, 0))

if __name__ == "__main__":
    robot, seed, idx = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    b = pkg.make_batch(4096, 10, robot, seed=seed, excite=1.0)
    cfg = pkg.mpc_cfg(robot)
    H, g = reduced(cfg, 10, b, idx)
    nls = H.shape[0] // 3
    M = np.linalg.inv(H); x0 = -M @ g
    N, c0 = rows(nls, float(np.float32(1) / np.float32(cfg[1]))); c0[5::6] = float(cfg[2])
    x, it, act, viol = gi_excl(M, x0, N, c0, verbose="-v" in sys.argv)
    u, st, rc = O.mpc_solve(cfg, 10, b["mpc_state"][idx], b["traj"][idx], b["gait"][idx])
    free = np.repeat(b["gait"][idx].reshape(-1) != 0, 3)
    print("iters", it, "q", len(act), "violations at exit:", [(divmod(i, 6), "%.2e" % v) for i, v in viol])
    print("max |x - oracle|: %.2e" % np.abs(x - u[free]).max())
