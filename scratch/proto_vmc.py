"""numpy emulation of qr_vmc_kernel's Schur-form active set, to debug decisions against the oracle."""
import sys
sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_py as O
from conftest import load_pkg
pkg = load_pkg(); W = pkg.workload

def emu(G, a, CI, b, verbose=False):
    Gd = np.tril(G.astype(np.float64)); Gd = Gd + np.tril(Gd, -1).T
    L = np.linalg.cholesky(Gd)
    c1 = np.trace(Gd); c2 = np.sum(1.0 / np.diag(L))
    M = np.linalg.inv(Gd)
    x = M @ a.astype(np.float64)
    N = CI.astype(np.float64)            # 12 x 24
    ci0 = -b.astype(np.float64)
    eps = 2.220446049250313e-16
    term = 24 * eps * c1 * c2 * 100
    active = np.zeros(24, bool); excl = np.zeros(24, bool)
    act = []; u = []; Sinv = np.zeros((0, 0)); it = 0
    while True:
        it += 1
        if it > 1900: return x, "maxit", it
        s = ci0 + N.T @ x
        psi = np.minimum(s, 0).sum()
        cand = (~active) & (~excl) & (s < 0)
        if not cand.any() or abs(psi) <= term: return x, "ok", it
        ip = int(np.argmin(np.where(cand, s, np.inf)))
        sip = s[ip]; unew = 0.0
        while True:
            it += 1
            if it > 1900: return x, "maxit", it
            q = len(act)
            w = M @ N[:, ip]
            d = np.array([N[:, c] @ w for c in act]); r = Sinv @ d if q else np.zeros(0)
            z = w - (M @ N[:, act]) @ r if q else w
            zz = z @ z; znp = z @ N[:, ip]
            tt = np.array([u[k] / r[k] if r[k] > 0 else np.inf for k in range(q)])
            t1 = tt.min() if q else np.inf; l = int(np.argmin(tt)) if q and t1 < np.inf else -1
            delta = N[:, ip] @ w
            t2 = -sip / znp if znp > 1e-8 * delta else np.inf
            t = min(t1, t2)
            if verbose: print(it, "ip", ip, "q", q, "zz %.3e znp %.3e t1 %.3e t2 %.3e sip %.3e" % (zz, znp, t1, t2, sip), "act", act)
            if not t < np.inf: return x, "inf", it
            dual_only = not t2 < np.inf
            if not dual_only: x = x + t * z
            u = [u[k] - t * r[k] for k in range(q)]; unew += t
            if not dual_only and t == t2:
                isg = 1.0 / znp
                S2 = np.zeros((q + 1, q + 1)); S2[:q, :q] = Sinv + np.outer(r, r) * isg; S2[q, :q] = -r * isg; S2[:q, q] = -r * isg; S2[q, q] = isg
                Sinv = S2; act.append(ip); u.append(unew); active[ip] = True; excl[:] = False
                break
            cl = act[l]
            col = Sinv[:, l].copy()
            Sinv = Sinv - np.outer(col, col) / col[l]
            keep = [k for k in range(q) if k != l]
            Sinv = Sinv[np.ix_(keep, keep)]; act = [act[k] for k in keep]; u = [u[k] for k in keep]
            active[cl] = False
            if not dual_only: sip = ci0[ip] + N[:, ip] @ x

if __name__ == "__main__":
    cfg = W.vmc_cfg("a1"); geom = pkg.model_desc("a1")[:3]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    vin, q = W.make_vmc_batch(n, sloped=float(sys.argv[2]) if len(sys.argv) > 2 else 0.0, excite=float(sys.argv[3]) if len(sys.argv) > 3 else 1.0, seed=n)
    bad = 0
    for i in range(n):
        G, a, CI, b = O.vmc_assemble(cfg, vin[i])
        x, st, it = emu(G, a, CI, b)
        force, tau, xo, sto, rc = O.vmc_solve(cfg, geom, vin[i], q[i])
        d = np.abs(x - xo).max()
        if st == "maxit" or d > 1e-6 or (st == "inf") != (rc == 1):
            bad += 1
            if bad <= 3:
                print("robot", i, st, it, "oracle rc", rc, sto, "diff %.2e" % d, "contacts", vin[i, 18:22])
                emu(G, a, CI, b, verbose=True) if it < 200 else None
    print("bad", bad)
