"""configs[4] per GPU at h = 16 with two four-wave workgroups per CU (QRGPU_H16_TWO): how many robots leave the main pass, and what the
solves cost there (instrumented kernels)."""
import sys, os, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
h, n = 16, 1024
ctx = pkg.Context(0, 4096, 16)
lib = ctx._lib
lib.qrgpu_debug_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qrgpu_debug_lists.argtypes = [C.c_void_p, C.c_void_p]
lib.qrgpu_debug_cycles(ctx._h, None, 0)
G.setup_a1(ctx, pkg, h)
ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h)
seed = 0xA1 + 2
sa = pkg.make_batch_sequence(n // 2, h, "a1", seed=seed, steps=8)
sl = pkg.make_batch_sequence(n // 2, h, "lite3", seed=seed + 0xD2, steps=8)
seq = []
for ba, bl in zip(sa, sl):
    b = dict(ba); b["n"] = n
    for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
        b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
    seq.append(b)
tid = pkg.shard.interleave_types(n, 2)
for t, k in enumerate(list(range(8)) + [6, 5, 4]):
    out = G.run_mpc(ctx, pkg, seq[k], type_id=tid)
    buf = np.zeros((n, 16), np.int64); lib.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
    cnt = np.zeros(8, np.int32); lib.qrgpu_debug_lists(ctx._h, cnt.ctypes.data)
    t0, t1 = buf[:, 12] / 100.0, buf[:, 13] / 100.0
    base = t0.min(); t0 -= base; t1 -= base
    d = t1 - t0; q = buf[:, 14]; nls = buf[:, 7] // 3
    ph = np.diff(buf[:, :7], axis=1) / 1000.0
    it = G.iterations(out["status"])
    print("t%d: lists (rescue p0 p1, planned p0 p1) %s | span %.0f us | robots by nls: %s" % (t, cnt.tolist(), t1.max(), {int(v): int((nls == v).sum()) for v in np.unique(nls)}))
    print("     all robots: mean solve %.0f us, mean q %.1f, q > 40: %d, q > 45: %d; phases (k cycles) load %.1f H %.1f sweep %.1f x0 %.1f AS %.1f out %.1f; flags %d" % (
        d.mean(), q.mean(), int((q > 40).sum()), int((q > 45).sum()), *ph.mean(0), int((G.flags(out["status"]) != 0).sum())))
    print("     solve length (us): p50 %.0f p90 %.0f p99 %.0f max %.0f; > 300: %d, > 400: %d, > 500: %d; sum %.1f ms" % (
        np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max(), int((d > 300).sum()), int((d > 400).sum()), int((d > 500).sum()), d.sum() / 1000.0))
    late = np.argsort(-t1)[:4]
    print("     last to end: " + " | ".join("robot %d start %.0f len %.0f q %d nls %d it %d" % (r, t0[r], d[r], q[r], nls[r], it[r]) for r in late))
