"""Prototype: primal-dual active set (block pivoting) on the reduced MPC QP; how often and how fast does it converge?"""
import sys, importlib.util, collections
sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_py as O
from conftest import load_pkg
pkg = load_pkg()

def reduced(cfg, h, b, i):
    H, g = O.mpc_assemble(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])[:2]
    H = np.asarray(H, np.float64).reshape(12 * h, 12 * h); g = np.asarray(g, np.float64)
    H = 0.5 * (H + H.T)
    free = np.repeat(b["gait"][i].reshape(-1) != 0, 3)
    return H[np.ix_(free, free)], g[free]

def rows(nls, im):
    n = 3 * nls
    N = np.zeros((6 * nls, n)); c0 = np.zeros(6 * nls)
    for k in range(nls):
        b = 3 * k
        N[6*k+0, [b, b+2]] = [im, 1]; N[6*k+1, [b, b+2]] = [-im, 1]
        N[6*k+2, [b+1, b+2]] = [im, 1]; N[6*k+3, [b+1, b+2]] = [-im, 1]
        N[6*k+4, b+2] = 1; N[6*k+5, b+2] = -1
    return N, c0

def sanitize(act, s, nls):
    """at most 3 rows per leg-step and never both rows of an opposing pair: keep the most violated"""
    act = act.copy()
    for k in range(nls):
        r = [6*k+t for t in range(6) if act[6*k+t]]
        for a, b_ in ((0, 1), (2, 3), (4, 5)):
            if act[6*k+a] and act[6*k+b_]:
                drop = 6*k+a if s[6*k+a] > s[6*k+b_] else 6*k+b_
                act[drop] = False
        r = [6*k+t for t in range(6) if act[6*k+t]]
        if len(r) > 3:
            r.sort(key=lambda j: s[j])
            for j in r[3:]: act[j] = False
    return act

def pdas(H, g, mu, fmax, maxit=20):
    n = H.shape[0]; nls = n // 3
    M = np.linalg.inv(H); x0 = -M @ g
    N, c0 = rows(nls, 1.0 / mu); c0[5::6] = fmax
    S = N @ M @ N.T
    s0 = N @ x0 + c0
    act = sanitize(s0 < -1e-9, s0, nls)
    seen = set()
    for it in range(1, maxit + 1):
        idx = np.where(act)[0]
        u = np.zeros(6 * nls)
        if len(idx):
            try:
                u[idx] = -np.linalg.solve(S[np.ix_(idx, idx)], s0[idx])
            except np.linalg.LinAlgError:
                return None, it, "singular"
        s = s0 + S @ u
        new = (act & (u > 0)) | (~act & (s < -1e-9))
        new = sanitize(new, np.where(act, -u, s), nls)
        if np.array_equal(new, act):
            return x0 + M @ N.T @ u, it, "ok"
        key = new.tobytes()
        if key in seen: return None, it, "cycle"
        seen.add(key); act = new
    return None, maxit, "cap"

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    h = 10
    b = pkg.make_batch(1024, h, "a1", seed=0xA1 + 2, excite=1.0)
    cfg = pkg.mpc_cfg("a1")
    res = collections.Counter(); its = []; err = []
    for i in range(n):
        H, g = reduced(cfg, h, b, i)
        x, it, st = pdas(H, g, float(cfg[1]), float(cfg[2]))
        res[st] += 1
        if st == "ok":
            its.append(it)
            u_or = O.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])[0]
            free = np.repeat(b["gait"][i].reshape(-1) != 0, 3)
            err.append(np.abs(x - u_or[free]).max())
    print(res, "iters mean %.1f max %d" % (np.mean(its), np.max(its)), "max err vs oracle %.2e" % np.max(err))
    print(np.bincount(its))
