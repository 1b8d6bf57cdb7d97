"""Prototype (numpy): Goldfarb-Idnani on the eliminated MPC QP with a WARM START from the previous tick's final working set:
solve the equality-constrained QP on the guessed set in one block (S = N'MN factorised once), drop rows with negative multipliers
until the point is dual feasible, then carry on with ordinary GI iterations.  Counts the working-set changes left after the start,
over a temporally coherent sequence (workload.make_batch_sequence)."""
import sys, importlib.util
sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import oracle_py as O
spec = importlib.util.spec_from_file_location("workload", "/root/repo/quadruped-robot_amd/workload.py"); W = importlib.util.module_from_spec(spec); spec.loader.exec_module(W)

def cvec(n, k, t, im):
    v = np.zeros(n); b = 3 * k
    if t == 0: v[b] = im; v[b + 2] = 1
    elif t == 1: v[b] = -im; v[b + 2] = 1
    elif t == 2: v[b + 1] = im; v[b + 2] = 1
    elif t == 3: v[b + 1] = -im; v[b + 2] = 1
    elif t == 4: v[b + 2] = 1
    else: v[b + 2] = -1
    return v

def gi(M, g, nls, im, fmax, A0=None, maxit=2000):
    n = 3 * nls
    x0 = -M @ g
    x = x0.copy()
    A, u = [], np.zeros(0)
    ci0 = lambda c: fmax[c[0]] if c[1] == 5 else 0.0
    N = lambda: np.array([cvec(n, k, t, im) for k, t in A]).T if A else np.zeros((n, 0))
    start_drops = 0
    if A0:
        A = list(A0)
        while A:
            Nm = N(); S = Nm.T @ M @ Nm
            if np.linalg.cond(S) > 1e13:            # dependent guess: fall back to a cold start
                A = []; break
            s = Nm.T @ x0 + np.array([ci0(c) for c in A])
            u = -np.linalg.solve(S, s)
            if u.min() >= -1e-12:
                x = x0 + M @ (Nm @ u); break
            j = int(np.argmin(u)); del A[j]; start_drops += 1
        if not A: x, u = x0.copy(), np.zeros(0)
    u = list(u)
    changes = 0
    it = 0
    while True:
        best, smin = None, -1e-9
        for k in range(nls):
            for t in range(6):
                if (k, t) in A: continue
                s = cvec(n, k, t, im) @ x + ci0((k, t))
                if s < smin: smin, best = s, (k, t)
        if best is None: return x, A, changes, start_drops
        p = best; cp = cvec(n, p[0], p[1], im); up = 0.0
        while True:
            it += 1
            if it > maxit: return x, A, -1, start_drops
            q = len(A); w = M @ cp; delta = cp @ w
            if q:
                Nm = N(); S = Nm.T @ M @ Nm; d = Nm.T @ w; r = np.linalg.solve(S, d); z = w - M @ (Nm @ r); zc = delta - d @ r
            else:
                r = np.zeros(0); z = w; zc = delta
            t1, l = np.inf, -1
            for j in range(q):
                if r[j] > 0 and u[j] / r[j] < t1: t1, l = u[j] / r[j], j
            sp = cp @ x + ci0(p)
            t2 = -sp / zc if zc > 1e-13 * delta else np.inf
            t = min(t1, t2)
            if t == np.inf: return x, A, -2, start_drops
            if t2 < np.inf: x = x + t * z
            for j in range(q): u[j] -= t * r[j]
            up += t
            changes += 1
            if t == t2:
                A.append(p); u.append(up); break
            del A[l]; del u[l]

if __name__ == "__main__":
    h, n = 10, int(sys.argv[1]) if len(sys.argv) > 1 else 48
    seq = W.make_batch_sequence(n, h, 'a1', seed=0xA1 + 2, steps=6)
    cfg = W.mpc_cfg('a1'); im = float(np.float32(1) / np.float32(0.45))
    prev = [None] * n
    prevg = [None] * n
    for s, b in enumerate(seq):
        cold, warm, sd, bad = [], [], [], 0
        for i in range(n):
            H, g, ub = O.mpc_assemble(cfg, h, b['mpc_state'][i], b['traj'][i], b['gait'][i])
            free = [k for k in range(4 * h) if ub[5 * k + 4] > 0]
            idx = np.array([3 * k + c for k in free for c in range(3)])
            Hd = H.astype(np.float64); Ha = 0.5 * (Hd + Hd.T)
            M = np.linalg.inv(Ha[np.ix_(idx, idx)]); gs = g.astype(np.float64)[idx]
            fm = [float(ub[5 * k + 4]) for k in free]
            xc, Ac, cc, _ = gi(M, gs, len(free), im, fm)
            A0 = None
            if prev[i] is not None:
                pos = {ls: j for j, ls in enumerate(free)}
                # the contact table scrolls as the gait phase advances: take the row shift under which the old table matches the new one best
                gnew = b['gait'][i].reshape(h, 4); gold = prevg[i]
                sh = min(range(3), key=lambda k: np.abs(gold[k:] - gnew[:h - k]).sum() + 0.01 * k)
                A0 = [(pos[ls - 4 * sh], t) for ls, t in prev[i] if (ls - 4 * sh) in pos]
            xw, Aw, cw, sdr = gi(M, gs, len(free), im, fm, A0)
            if cw < 0 or np.abs(xw - xc).max() > 1e-6 * max(1, np.abs(xc).max()): bad += 1
            prev[i] = [(free[k], t) for k, t in Aw]
            prevg[i] = b['gait'][i].reshape(h, 4).copy()
            cold.append(cc); warm.append(cw); sd.append(sdr)
        print("step %d: changes cold mean %.1f max %d | warm mean %.1f max %d | start drops mean %.2f max %d | mismatches %d" % (
            s, np.mean(cold), max(cold), np.mean(warm), max(warm), np.mean(sd), max(sd), bad))
