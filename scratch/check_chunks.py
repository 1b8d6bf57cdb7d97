"""Large batch through the pipelined tick (chunked WBC launches, QRGPU_WBC_CHUNKS): a coherent 4-tick sequence of N robots queued without a sync,
outputs saved to OUT; run once with QRGPU_WBC_CHUNKS=1 and once with 0 and compare bit for bit (scratch/check_chunks.sh), last tick against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import conftest, gpu_helpers as G, oracle_py as O
O.build()
pkg = conftest.load_pkg()
n, h, K = int(os.environ.get("N", 4096)), 10, 4
seq = pkg.make_batch_sequence(n, h, "a1", seed=0x4096, steps=K, excite=1.0)
ctx = pkg.Context(0, n, 16)
G.setup_a1(ctx, pkg, h)
ctx.set_torque_epilogue(hip_comp=True, clip=True)
S = pkg.to_soa
prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
bufs = [dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
             fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])),
             force=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)),
             qdes=ctx.alloc((24, n)), status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32))) for b in seq]
ctx.sync()
for d in bufs:
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], prev, d["force"], d["tau"], d["status"], qdes=d["qdes"])
ctx.sync()
out = {}
for k, d in enumerate(bufs):
    out["tau%d" % k] = d["tau"].download(); out["force%d" % k] = d["force"].download(); out["status%d" % k] = d["status"].download(); out["qdes%d" % k] = d["qdes"].download()
np.savez(os.environ["OUT"], **out)
d, b = bufs[-1], seq[-1]
prev_in = seq[-2]["wbc_cmd"][:, 12:15]
f, tau, st, sec, pv = O.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"],
                                   np.ascontiguousarray(prev_in, np.float32).copy(), nthreads=32, epilogue=3)
status = out["status%d" % (K - 1)]; to = out["tau%d" % (K - 1)].T
assert np.all(np.isfinite(to))
ok = (G.flags(status) == 0) & (st == 0)
print("n %d: ok %.4f, max torque error %.2e (tolerance 1e-4 relative), flags %d" % (n, ok.mean(), (np.abs(to[ok] - tau[ok]) / np.maximum(1.0, np.abs(tau[ok]))).max(), int((G.flags(status) != 0).sum())))
assert ok.mean() > 0.99 and np.all(np.abs(to[ok] - tau[ok]) <= G.tau_tol(tau[ok], 1e-4))
ctx.close()
