"""configs[4] per GPU (512 A1 + 512 Lite3, h = 16): the longest robots of each tick of a coherent sequence, followed over the ticks: duration,
changes, final working set, status -- does the warm start hold for them?  (instrumented kernels: qrgpu_debug_cycles)"""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
h, n = 16, 1024
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
G.setup_a1(ctx, pkg, h)
ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0xA1 + 2
sa = pkg.make_batch_sequence(n // 2, h, "a1", seed=seed, steps=8)
sl = pkg.make_batch_sequence(n // 2, h, "lite3", seed=seed + 0xD2, steps=8)
seq = []
for ba, bl in zip(sa, sl):
    b = dict(ba); b["n"] = n
    for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
        b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
    seq.append(b)
tid = pkg.shard.interleave_types(n, 2)
walk = list(range(8)) + list(range(6, 0, -1)) + list(range(8))
D, IT, Q, ST, NS = [], [], [], [], []
for k in walk:
    out = G.run_mpc(ctx, pkg, seq[k], type_id=tid)
    buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    D.append((buf[:, 13] - buf[:, 12]) / 100.0); IT.append(G.iterations(out["status"])); Q.append(buf[:, 14].copy()); ST.append(G.flags(out["status"])); NS.append(buf[:, 7] // 3)
D, IT, Q, ST, NS = map(np.array, (D, IT, Q, ST, NS))
print("per tick: max duration (us), robot, its changes / final q / leg-steps | second longest")
for t in range(len(walk)):
    o = np.argsort(-D[t])[:3]
    print("  t%2d batch %d: " % (t, walk[t]) + " | ".join("robot %4d %6.1f us it %3d q %2d nls %2d" % (r, D[t][r], IT[t][r], Q[t][r], NS[t][r]) for r in o) + "   sum/256 = %.0f us" % (D[t].sum() / 256))
worst = np.unique(np.concatenate([np.argsort(-D[t])[:2] for t in range(4, len(walk))]))
print("the robots that were among the two longest of some tick, over all ticks (duration us / changes / final q / flags):")
for r in worst:
    print("  robot %4d (type %d, nls %s): " % (r, tid[r], sorted(set(NS[:, r].tolist()))) + "  ".join("%.0f/%d/%d%s" % (D[t][r], IT[t][r], Q[t][r], "" if ST[t][r] == 0 else "!%x" % ST[t][r]) for t in range(len(walk))))
