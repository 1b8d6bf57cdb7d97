"""Replay a tick log (ticklog.py / include/qrgpu_ticklog.h) through the batched GPU tick and compare with what the log recorded.

Stateless replay feeds every tick the WBC memory (prev_ori_vel) the log recorded BEFORE that tick, so every tick stands alone;
stateful replay carries the library's own memory from tick to tick, as a controller would.  Errors are relative as BASELINE's
tolerance is stated: |a - b| / max(1, |b|) per motor (torques, stance legs of the WBC only -- swing-leg torques of the fused tick
are the MPC's J^T f) and per force component.
"""
import numpy as np

from . import ticklog
from .qrgpu import status_flags

GAIN_NAMES = ("kp_body_pos", "kd_body_pos", "kp_body_ori", "kd_body_ori", "kp_foot", "kd_foot", "weight_fb", "weight_fr", "mu")


def setup_from_log(ctx, log, type_id=0):
    ctx.mpc_setup_packed(type_id, log.mpc_cfg, log.horizon)
    m = log.model
    ctx.wbc_setup(type_id, float(m[0]), float(m[1]), float(m[2]), [float(x) for x in m[3:6]], **{k: float(v) for k, v in zip(GAIN_NAMES, m[6:15])})


def replay(ctx, log, stateful=False, first=0, count=None, to_soa=None):
    """-> dict(ticks, robot_ticks, worst_force, worst_tau, flagged, recorded_flagged, per_tick=[(force_err, tau_err, flagged)])"""
    if to_soa is None:
        from .workload import to_soa
    n, h = log.n_robots, log.horizon
    last = log.ticks if count is None else min(log.ticks, first + count)
    d = dict(state=ctx.alloc((28, n)), traj=ctx.alloc((12 * h, n)), gait=ctx.alloc((4 * h, n)), fb=ctx.alloc((37, n)), cmd=ctx.alloc((67, n)),
             prev=ctx.alloc((3, n)), force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32))
    worst_f = worst_t = 0.0
    flagged = rec_flagged = 0
    per_tick = []
    ctx.set_warm_start(False)        # a log replays onto itself bit for bit only when no solve depends on the solves before it
    try:
        for k in range(first, last):
            t = log.tick(k)
            d["state"].upload(to_soa(t["mpc_state"])); d["traj"].upload(to_soa(t["traj"])); d["gait"].upload(to_soa(t["gait"]))
            d["fb"].upload(to_soa(t["fb_state"])); d["cmd"].upload(to_soa(t["wbc_cmd"]))
            if not stateful or k == first:
                d["prev"].upload(to_soa(t["prev_ori_vel"]))
            ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], d["tau"], d["status"])
            ctx.sync()
            force, tau, status = d["force"].download().T, d["tau"].download().T, d["status"].download()
            bad = (status_flags(status) != 0) | (status_flags(t["status"]) != 0)
            flagged += int((status_flags(status) != 0).sum()); rec_flagged += int((status_flags(t["status"]) != 0).sum())
            ok = ~bad
            ef = np.abs(force - t["force"]) / np.maximum(1.0, np.abs(t["force"]).max(axis=1, keepdims=True))
            stance = np.repeat(t["wbc_cmd"][:, 63:67] != 0, 3, axis=1)
            et = np.where(stance, np.abs(tau - t["tau"]) / np.maximum(1.0, np.abs(t["tau"])), 0.0)
            f_k = float(ef[ok].max(initial=0.0)); t_k = float(et[ok].max(initial=0.0))
            worst_f, worst_t = max(worst_f, f_k), max(worst_t, t_k)
            per_tick.append((f_k, t_k, int(bad.sum())))
    finally:
        ctx.set_warm_start(True)
        for v in d.values():
            v.free()
    return dict(ticks=last - first, robot_ticks=(last - first) * n, worst_force=worst_f, worst_tau=worst_t, flagged=flagged,
                recorded_flagged=rec_flagged, per_tick=per_tick)


def record(ctx, path, batches, mpc_cfg, model6, robot="", to_soa=None):
    """Run a sequence of input batches (dicts as workload.make_batch returns) through the GPU tick statefully and write the log."""
    if to_soa is None:
        from .workload import to_soa
    b0 = batches[0]
    n, h = b0["n"], b0["horizon"]
    d = dict(state=ctx.alloc((28, n)), traj=ctx.alloc((12 * h, n)), gait=ctx.alloc((4 * h, n)), fb=ctx.alloc((37, n)), cmd=ctx.alloc((67, n)),
             prev=ctx.alloc((3, n)), force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32))
    ctx.set_warm_start(False)        # (see replay)
    try:
        d["prev"].upload(to_soa(b0["prev_ori_vel"]))
        with ticklog.TickLogWriter(path, n, h, mpc_cfg, ticklog.model15(model6), robot) as w:
            for b in batches:
                prev_before = d["prev"].download().T.copy()
                d["state"].upload(to_soa(b["mpc_state"])); d["traj"].upload(to_soa(b["traj"])); d["gait"].upload(to_soa(b["gait"]))
                d["fb"].upload(to_soa(b["fb_state"])); d["cmd"].upload(to_soa(b["wbc_cmd"]))
                ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], d["tau"], d["status"])
                ctx.sync()
                w.append(dict(b, prev_ori_vel=prev_before), d["force"].download().T, d["tau"].download().T, d["status"].download())
    finally:
        ctx.set_warm_start(True)
        for v in d.values():
            v.free()
