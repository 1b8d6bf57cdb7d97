"""Shared helpers for the -m gpu parity tests: upload a workload batch, run the C-ABI, download."""
import contextlib

import numpy as np

ST_FLAG_MASK = 0xff0000ff          # include/qrgpu.h QRGPU_ST_FLAG_MASK: bits 0-7 and 24-31 (BAD_TYPE) are flags, 8-23 the MPC iteration count


def flags(status):
    """Flag bits of a status word / array; 0 = converged and usable."""
    return np.asarray(status).astype(np.int64) & ST_FLAG_MASK


def iterations(status):
    return (np.asarray(status).astype(np.int64) >> 8) & 0xffff


@contextlib.contextmanager
def cold_start(ctx):
    """Warm start off inside the block: every solve starts from the empty working set, so that two calls on the same inputs return the
    same bits whatever was solved in between (with it on they agree to the solver's tolerance, and the iteration counts differ).  The
    planned list is switched off too: it moves robots between launches from one call to the next."""
    ctx.set_warm_start(False)
    ctx.set_planned_list(False)          # (which launch solves a robot changes its LDS allotment, hence the form of its z update, hence the last bits)
    try:
        yield ctx
    finally:
        ctx.set_warm_start(True)
        ctx.set_planned_list(True)


def run_mpc(ctx, pkg, b, with_tau=True, type_id=None):
    n, h = b["n"], b["horizon"]
    S = pkg.to_soa
    d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])),
             gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])), q=ctx.alloc((12, n)).upload(S(b["fb_state"][:, 13:25])),
             force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32))
    # poisoned outputs: a robot that no launch solves shows as NaN forces and an all-ones flag word, not as whatever the allocator left there
    d["force"].upload(np.full((12, n), np.nan, np.float32)); d["tau"].upload(np.full((12, n), np.nan, np.float32))
    d["status"].upload(np.full((n,), 0x7f0000ff, np.int32))
    tid = ctx.alloc((n,), np.int32).upload(type_id) if type_id is not None else None
    ctx.mpc_solve_batch(n, d["state"], d["traj"], d["gait"], d["q"], d["force"], d["tau"] if with_tau else None, d["status"], tid)
    ctx.sync()
    out = dict(force=d["force"].download().T.copy(), tau=d["tau"].download().T.copy(), status=d["status"].download())
    for v in d.values():
        v.free()
    if tid is not None:
        tid.free()
    return out


def run_assemble(ctx, pkg, b, type_id=None):
    n, h = b["n"], b["horizon"]
    S = pkg.to_soa
    nv = 12 * h
    state = ctx.alloc((28, n)).upload(S(b["mpc_state"])); traj = ctx.alloc((12 * h, n)).upload(S(b["traj"]))
    gait = ctx.alloc((4 * h, n)).upload(S(b["gait"]))
    H = ctx.alloc((n, nv, nv)).upload(np.full((n, nv, nv), np.nan, np.float32)); g = ctx.alloc((n, nv)).upload(np.full((n, nv), np.nan, np.float32))
    tid = ctx.alloc((n,), np.int32).upload(type_id) if type_id is not None else None
    ctx.mpc_assemble_batch(n, state, traj, gait, H, g, tid)
    ctx.sync()
    out = (H.download(), g.download())
    for v in (state, traj, gait, H, g):
        v.free()
    if tid is not None:
        tid.free()
    return out


def run_wbc(ctx, pkg, b, wbc_cmd=None, prev=None, want_qdes=True, type_id=None):
    n = b["n"]
    S = pkg.to_soa
    cmd = b["wbc_cmd"] if wbc_cmd is None else wbc_cmd
    prev = b["prev_ori_vel"] if prev is None else prev
    d = dict(state=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(cmd)),
             prev=ctx.alloc((3, n)).upload(S(prev)), tau=ctx.alloc((12, n)), qdes=ctx.alloc((24, n)), status=ctx.alloc((n,), np.int32))
    tid = ctx.alloc((n,), np.int32).upload(type_id) if type_id is not None else None
    ctx.wbc_run_batch(n, d["state"], d["cmd"], d["prev"], d["tau"], d["qdes"] if want_qdes else None, d["status"], tid)
    ctx.sync()
    q = d["qdes"].download()
    out = dict(tau=d["tau"].download().T.copy(), qdes=q[:12].T.copy(), qddes=q[12:].T.copy(), prev=d["prev"].download().T.copy(),
               status=d["status"].download())
    for v in d.values():
        v.free()
    if tid is not None:
        tid.free()
    return out


def run_wbc_inspect(ctx, pkg, b, wbc_cmd=None, prev=None, type_id=None):
    """qrgpu_wbc_inspect_batch: the WBC tick plus its relaxation QP's solution.  -> tau[n,12], z[n,18], fr[n,12] (optimalFr), status"""
    n = b["n"]
    S = pkg.to_soa
    cmd = b["wbc_cmd"] if wbc_cmd is None else wbc_cmd
    prev = b["prev_ori_vel"] if prev is None else prev
    d = dict(state=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(cmd)), prev=ctx.alloc((3, n)).upload(S(prev)),
             tau=ctx.alloc((12, n)), qp=ctx.alloc((n, 30)).upload(np.full((n, 30), np.nan, np.float32)), status=ctx.alloc((n,), np.int32))
    tid = ctx.alloc((n,), np.int32).upload(type_id) if type_id is not None else None
    ctx.wbc_inspect_batch(n, d["state"], d["cmd"], d["prev"], d["tau"], d["qp"], d["status"], tid)
    ctx.sync()
    qp = d["qp"].download()
    out = dict(tau=d["tau"].download().T.copy(), z=qp[:, :18].copy(), fr=qp[:, 18:].copy(), status=d["status"].download())
    for v in d.values():
        v.free()
    if tid is not None:
        tid.free()
    return out


def run_fb_debug(ctx, pkg, b, type_id=None):
    n = b["n"]
    state = ctx.alloc((37, n)).upload(pkg.to_soa(b["fb_state"]))
    out = ctx.alloc((n, 612))
    tid = ctx.alloc((n,), np.int32).upload(type_id) if type_id is not None else None
    ctx.fb_debug_batch(n, state, out, tid)
    ctx.sync()
    o = out.download()
    state.free(); out.free()
    if tid is not None:
        tid.free()
    return dict(H=o[:, :324].reshape(n, 18, 18), G=o[:, 324:342], C=o[:, 342:360], Jc=o[:, 360:576].reshape(n, 4, 3, 18),
                Jcdqd=o[:, 576:588].reshape(n, 4, 3), pGC=o[:, 588:600].reshape(n, 4, 3), vGC=o[:, 600:612].reshape(n, 4, 3))


def run_tick(ctx, pkg, b, type_id=None, want_qdes=False):
    n, h = b["n"], b["horizon"]
    S = pkg.to_soa
    d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])),
             gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])), fb=ctx.alloc((37, n)).upload(S(b["fb_state"])),
             cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
             force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32), qdes=ctx.alloc((24, n)))
    d["tau"].upload(np.full((12, n), np.nan, np.float32)); d["status"].upload(np.full((n,), 0x7f0000ff, np.int32))      # (poisoned: see run_mpc)
    tid = ctx.alloc((n,), np.int32).upload(type_id) if type_id is not None else None
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], d["tau"], d["status"], tid,
                   qdes=d["qdes"] if want_qdes else None)
    ctx.sync()
    out = dict(force=d["force"].download().T.copy(), tau=d["tau"].download().T.copy(), status=d["status"].download(),
               prev=d["prev"].download().T.copy())
    if want_qdes:
        out["qdes"] = d["qdes"].download().T.copy()
    for v in d.values():
        v.free()
    if tid is not None:
        tid.free()
    return out


def setup_a1(ctx, pkg, horizon=10):
    ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), horizon)
    ctx.wbc_setup_packed(0, pkg.model_desc("a1"))


def tau_tol(tau_ref, rel=1e-4):
    """north_star tolerance: 1e-4 * max(1, |tau_cpu|) per motor."""
    return rel * np.maximum(1.0, np.abs(tau_ref))
