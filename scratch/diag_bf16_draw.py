"""What makes one draw of the bf16x3 mixed h = 16 bench slow: per-step worst iteration count and status flags."""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg(); pkg._build.build()
n, h, d = 1024, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 6
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
ctx = pkg.Context(0, n, 16)
ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h); ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
ctx.set_hessian_mode(mode)
seed = 0xA1 + 2 + 1000 * d
sa = pkg.make_batch_sequence(n // 2, h, "a1", seed=seed, steps=8)
sl = pkg.make_batch_sequence(n // 2, h, "lite3", seed=seed + 0xD2, steps=8)
tid = pkg.shard.interleave_types(n, 2)
seq = []
for ba, bl in zip(sa, sl):
    b = dict(ba); b["n"] = n
    for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
        b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
    seq.append(b)
walk = list(range(8)) + list(range(6, 0, -1))
for i in range(30):
    b = seq[walk[i % len(walk)]]
    out = G.run_tick(ctx, pkg, b, type_id=tid)
    st = out["status"].astype(np.int64)
    it = (st >> 8) & 0xffff
    fl = st & 0xff0000ff
    w = int(np.argmax(it))
    print("step %2d (batch %d): worst robot %4d with %4d changes (stance leg-steps %d), mean %.1f, flagged %d %s" % (
        i, walk[i % len(walk)], w, it[w], int((b["gait"][w] > 0).sum()), it.mean(), int((fl != 0).sum()), np.unique(fl[fl != 0]).tolist()))
