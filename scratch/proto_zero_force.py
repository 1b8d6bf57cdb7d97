"""Do stance leg-steps that end at zero force (a degenerate vertex: fz >= 0 and the four friction rows all tight) drive the change count?
Oracle solutions of one batch of the bench's coherent sequence against the GPU's change counts of the same tick (scratch/diag_predict.py)."""
import sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from conftest import load_pkg
import oracle_py as O
O.build(); pkg = load_pkg()
n, h = 1024, 10
z = np.load('/root/repo/gpurun_out/slots/predict.npz'); it = z['it']; buf = z['buf']
t = 5
seq = pkg.make_batch_sequence(n, h, "a1", seed=0xA1 + 2, steps=8)
b = seq[5]
cfg = pkg.mpc_cfg("a1"); fmax = float(cfg[2])
nz = np.zeros(n, int); nfr = np.zeros(n, int); nst = np.zeros(n, int)
for r in range(n):
    u, info, rc = O.mpc_solve(cfg, h, b["mpc_state"][r], b["traj"][r], b["gait"][r])
    gait = b["gait"][r].reshape(h, 4)
    for s in range(h):
        for l in range(4):
            if gait[s, l] == 0: continue
            nst[r] += 1
            fx, fy, fz = u[12 * s + 3 * l: 12 * s + 3 * l + 3]
            if fz < 1e-6 * fmax: nz[r] += 1
            else:
                nfr[r] += int(0.45 * fz - abs(fx) < 1e-6 * fmax) + int(0.45 * fz - abs(fy) < 1e-6 * fmax) + int(fz > fmax * (1 - 1e-6))
d = (buf[t, :, 13] - buf[t, :, 12]) / 100.0
itt = it[t].astype(float)
print("robots with k zero-force stance leg-steps: ", {int(k): int((nz == k).sum()) for k in np.unique(nz)})
for k in np.unique(nz):
    m = nz == k
    print("  k = %2d: %4d robots, changes mean %.1f max %d, solve time mean %.0f us, other tight rows mean %.1f" % (k, m.sum(), itt[m].mean(), itt[m].max(), d[m].mean(), nfr[m].mean()))
print("corr(changes, zero-force leg-steps) %.2f; corr(changes, other tight rows) %.2f; corr(changes, all tight rows incl. 5 per zero leg-step) %.2f" % (
    np.corrcoef(itt, nz)[0, 1], np.corrcoef(itt, nfr)[0, 1], np.corrcoef(itt, nfr + 5 * nz)[0, 1]))
o = np.argsort(-itt)[:12]
print("the twelve robots with most changes: (changes, zero-force leg-steps, other tight rows, stance leg-steps, us)", [(int(itt[r]), int(nz[r]), int(nfr[r]), int(nst[r]), int(d[r])) for r in o])
