"""Prototype: does the choice of the violated row (the pivot rule of the dual active set) change the number of working-set changes?
Rules: most violated slack (what the kernel does); slack / sqrt(c' M c) (distance in the M-norm: 'steepest edge'); slack / (c' M c)."""
import sys
sys.path.insert(0, '/root/repo/scratch')
import numpy as np
from proto_pdas import reduced, rows, pkg

def gi(M, x0, N, c0, rule, maxit=3000):
    m = N.shape[0]
    dlt = np.einsum('ij,jk,ik->i', N, M, N)
    act, u, it = [], [], 0
    Si = np.zeros((0, 0))
    x = x0.copy()
    excl = set()
    while True:
        s = N @ x + c0
        cand = [i for i in range(m) if i not in act and i not in excl and s[i] < -1e-9]
        if not cand: return x, it, len(act)
        if rule == 0: ip = min(cand, key=lambda i: s[i])
        elif rule == 1: ip = min(cand, key=lambda i: s[i] / np.sqrt(dlt[i]))
        elif rule == 2: ip = min(cand, key=lambda i: s[i] / dlt[i])
        elif rule == 3: ip = min(cand, key=lambda i: (0 if i % 6 >= 4 else 1, s[i]))          # normal-force rows first, then friction
        elif rule == 4: ip = min(cand, key=lambda i: (0 if i % 6 < 4 else 1, s[i]))           # friction rows first
        elif rule == 5: ip = min(cand, key=lambda i: (i // 6, s[i]))                           # lowest leg-step (earliest horizon step) first
        else: ip = min(cand, key=lambda i: (-(i // 6), s[i]))                                  # latest first
        unew = 0.0
        while True:
            it += 1
            if it > maxit: return x, it, len(act)
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    h = 10
    b = pkg.make_batch(1024, h, "a1", seed=0xA1 + 2, excite=1.0)
    cfg = pkg.mpc_cfg("a1")
    NR = 7
    its = [[] for _ in range(NR)]; nlss = []
    for i in range(n):
        H, g = reduced(cfg, h, b, i)
        nls = H.shape[0] // 3
        M = np.linalg.inv(H); x0 = -M @ g
        N, c0 = rows(nls, 1.0 / float(cfg[1])); c0[5::6] = float(cfg[2])
        xs = []
        for rule in range(NR):
            x, it, q = gi(M, x0, N, c0, rule)
            its[rule].append(it); xs.append(x)
        assert all(np.abs(xs[0] - xs[r]).max() < 1e-5 * max(1, np.abs(xs[0]).max()) for r in range(1, NR)), i
        nlss.append(nls)
    its = np.array(its); nlss = np.array(nlss)
    for rule, nm in enumerate(("most violated", "s / sqrt(c'Mc)", "s / c'Mc", "fz rows first", "friction first", "earliest step", "latest step")):
        print("%-16s iterations mean %.2f  p90 %d  max %d   | nls=40 robots: mean %.1f max %d | nls<40: max %d" % (nm, its[rule].mean(), np.percentile(its[rule], 90), its[rule].max(),
              its[rule][nlss == 40].mean() if (nlss == 40).any() else 0, its[rule][nlss == 40].max() if (nlss == 40).any() else 0, its[rule][nlss < 40].max()))
