"""Prototype: crash start for the dual active set.  A0 = rows violated at the unconstrained optimum (made independent), one
block solve, drop negative multipliers until dual feasible, then the ordinary Goldfarb-Idnani iterations.  How many iterations
does that save on the bench batch?"""
import sys
sys.path.insert(0, '/root/repo/scratch')
import numpy as np
from proto_pdas import reduced, rows, sanitize, pkg, O

def gi(M, x0, N, c0, A_init=None, maxit=2000):
    """Dense GI from a dual-feasible start.  Returns x, iterations (adds + drops), setup drops."""
    m = N.shape[0]
    act = list(A_init) if A_init is not None else []
    setup = 0
    def solve(act):
        if not act: return x0.copy(), np.zeros(0), np.zeros((0, 0))
        NA = N[act]; S = NA @ M @ NA.T
        Si = np.linalg.inv(S)
        u = -Si @ (NA @ x0 + c0[act])
        return x0 + M @ NA.T @ u, u, Si
    x, u, Si = solve(act)
    while len(u) and u.min() < 0:               # make the start dual feasible: drop the most negative multiplier
        act.pop(int(np.argmin(u))); setup += 1
        x, u, Si = solve(act)
    u = list(u); it = 0
    excl = set()
    while True:
        s = N @ x + c0
        cand = [i for i in range(m) if i not in act and i not in excl and s[i] < -1e-9]
        if not cand: return x, it, setup, len(act)
        ip = min(cand, key=lambda i: s[i]); unew = 0.0
        while True:
            it += 1
            if it > maxit: return x, it, setup, len(act)
            NA = N[act] if act else np.zeros((0, N.shape[1]))
            w = M @ N[ip]
            d = NA @ w
            r = Si @ d if act else np.zeros(0)
            z = w - M @ NA.T @ r if act else w
            zc = N[ip] @ z; delta = N[ip] @ w
            tt = [u[k] / r[k] if r[k] > 0 else np.inf for k in range(len(act))]
            t1 = min(tt) if tt else np.inf
            t2 = -(N[ip] @ x + c0[ip]) / zc if zc > 1e-13 * delta else np.inf
            t = min(t1, t2)
            if not t < np.inf: excl.add(ip); break
            if t2 < np.inf: x = x + t * z
            u = [u[k] - t * r[k] for k in range(len(act))]; unew += t
            if t2 < np.inf and t == t2:
                act.append(ip); u.append(unew); excl.clear()
                NA = N[act]; Si = np.linalg.inv(NA @ M @ NA.T)
                break
            l = int(np.argmin(tt)); act.pop(l); u.pop(l)
            NA = N[act] if act else np.zeros((0, N.shape[1])); Si = np.linalg.inv(NA @ M @ NA.T) if act else np.zeros((0, 0))

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    h = 10
    b = pkg.make_batch(1024, h, "a1", seed=0xA1 + 2, excite=1.0)
    cfg = pkg.mpc_cfg("a1")
    base, crash, setups, q0s, qf = [], [], [], [], []
    for i in range(n):
        H, g = reduced(cfg, h, b, i)
        nls = H.shape[0] // 3
        M = np.linalg.inv(H); x0 = -M @ g
        N, c0 = rows(nls, 1.0 / float(cfg[1])); c0[5::6] = float(cfg[2])
        xb, itb, _, qb = gi(M, x0, N, c0)
        s0 = N @ x0 + c0
        A0 = list(np.where(sanitize(s0 < -1e-9, s0, nls))[0])
        xc, itc, st, qc = gi(M, x0, N, c0, A_init=A0)
        assert np.abs(xb - xc).max() < 1e-6 * max(1, np.abs(xb).max()), (i, np.abs(xb - xc).max())
        base.append(itb); crash.append(itc); setups.append(st); q0s.append(len(A0)); qf.append(qb)
    base, crash, setups, q0s = map(np.array, (base, crash, setups, q0s))
    print("baseline iterations: mean %.1f max %d" % (base.mean(), base.max()))
    print("crash start: |A0| mean %.1f; setup drops mean %.1f max %d; remaining iterations mean %.1f max %d" % (q0s.mean(), setups.mean(), setups.max(), crash.mean(), crash.max()))
    print("total (setup drops + remaining): mean %.1f max %d" % ((setups + crash).mean(), (setups + crash).max()))
    k = np.argsort(-base)[:8]
    print("worst baseline robots:", [(int(base[j]), int(q0s[j]), int(setups[j]), int(crash[j])) for j in k])
