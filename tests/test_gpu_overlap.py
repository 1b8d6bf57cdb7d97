"""-m gpu: overlapped ticks (qrgpu_set_tick_overlap) -- tick t + 1's solves start in the slots tick t's drain leaves empty.

What is held here: results against the serial tick (bit for bit for every robot the main pass solves on both sides; to the solver's
tolerance for robots on a list launch, whose kernel variant differs in the overlapped form), against the oracle, that the ticks really were
chained, that a tick which breaks the caller's side of the contract (same output arrays, another batch size, a launch in between) is simply
not chained, and -- in processes of their own, because the bound is read once -- that per-robot waits which give up flag their robots and
still return the right torques."""
import os
import subprocess
import sys

import numpy as np
import pytest

import gpu_helpers as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_sequence(pkg, seq, h, n, mode, sync_every_tick, same_outputs=False, planned=True):
    """mode: 'serial' (WBC launch behind the MPC launches), 'piped' (round 3's pipelined tick), 'overlap'.  -> per tick dict of outputs, stats"""
    ctx = pkg.Context(0, max(1024, n), 16)
    try:
        G.setup_a1(ctx, pkg, h)
        ctx.set_tick_pipeline(mode != "serial")
        ctx.set_planned_list(planned)
        if mode == "overlap":
            assert ctx.set_tick_overlap(True)
        ctx.set_torque_epilogue(hip_comp=True, clip=True)
        S = pkg.to_soa
        d_prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
        bufs = []
        shared = None
        for b in seq:                     # every tick's inputs resident before the first call (the overlapped mode's promise)
            d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])),
                     gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])), fb=ctx.alloc((37, n)).upload(S(b["fb_state"])),
                     cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])))
            if same_outputs and shared is not None:
                d.update(shared)
            else:
                o = dict(force=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)),
                         qdes=ctx.alloc((24, n)), status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32)))
                d.update(o)
                shared = o
            bufs.append(d)
        ctx.sync()
        res = []
        prevs = []
        for d in bufs:
            ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d_prev, d["force"], d["tau"], d["status"], qdes=d["qdes"])
            if sync_every_tick:
                ctx.sync()
                prevs.append(d_prev.download().T.copy())
                if same_outputs:
                    res.append(dict(force=d["force"].download().T.copy(), tau=d["tau"].download().T.copy(), status=d["status"].download(), qdes=d["qdes"].download().T.copy()))
        ctx.sync()
        if not (sync_every_tick and same_outputs):
            res = [dict(force=d["force"].download().T.copy(), tau=d["tau"].download().T.copy(), status=d["status"].download(), qdes=d["qdes"].download().T.copy()) for d in bufs]
        final_prev = d_prev.download().T.copy()
        stats = ctx.tick_overlap_stats()
        return res, prevs, final_prev, stats
    finally:
        ctx.close()


def _compare(pkg, seq, a, b_, h, what):
    """Robots the main pass solves on both sides are bit-identical; robots that see a list launch on either side (all-stance / three-leg gaits, the
    class whose inverse Hessian leaves no room for S^-1 in half a CU, and whoever outgrows 58 rows) agree to the solver's tolerance."""
    n = a[0]["tau"].shape[0]
    exact_all = np.ones(n, bool)
    for k, (x, y) in enumerate(zip(a, b_)):
        assert np.all(G.flags(y["status"]) & 0x02000000 == 0), (what, k, "a per-robot wait gave up")
        assert np.array_equal(G.flags(x["status"]) != 0, G.flags(y["status"]) != 0) or np.sum((G.flags(x["status"]) != 0) != (G.flags(y["status"]) != 0)) <= 2, (what, k)
        same = np.all(x["force"] == y["force"], 1) & np.all(x["tau"] == y["tau"], 1) & np.all(x["qdes"] == y["qdes"], 1)
        exact_all &= same
        ok = (G.flags(x["status"]) == 0) & (G.flags(y["status"]) == 0)
        ef = np.abs(x["force"] - y["force"]).max(1) / np.maximum(1.0, np.abs(x["force"]).max(1))
        et = (np.abs(x["tau"] - y["tau"]) / np.maximum(1.0, np.abs(x["tau"]))).max(1)
        assert ef[ok].max() <= 1e-6 and et[ok].max() <= 1e-5, (what, k, ef[ok].max(), et[ok].max())
    # a trotting robot (at most 28 of the 40 leg-steps of the contact table in stance: neither all-stance nor three-leg) never needs a list launch
    # inside the 8d ranges: bit for bit
    trot = np.all(np.stack([np.asarray(b["gait"]).reshape(n, -1).sum(1) <= 28 for b in seq]), 0)
    assert trot.sum() > n // 2
    assert np.all(exact_all[trot]), (what, int((~exact_all[trot]).sum()))
    return exact_all


def test_overlapped_ticks_are_the_serial_ticks(pkg, oracle):
    """Ten ticks of a temporally coherent sequence queued WITHOUT a sync on the overlapped form (own output arrays per tick, warm start, planned
    list, dispatch history, K12 and the K14 tail on) against the same ten ticks on the serial form, one at a time: depth 2 -- every tick but the
    first is chained to its predecessor (qrgpu_tick_overlap_stats).  The last tick is also held against the oracle, and the orientation task's
    memory after the last tick is the serial run's bit for bit (it is handed from tick to tick per robot, behind wbc_done)."""
    h, n = 10, 1024
    # (trotting robots only: nobody wants a whole CU, no lane ever finds a plan, so every tick but the first is chained)
    seq = pkg.make_batch_sequence(n, h, "a1", seed=0x0E1A, steps=10, excite=1.0, frac_all_stance=0.0, frac_three_leg=0.0)
    ser, prevs, prev_s, _ = _run_sequence(pkg, seq, h, n, "serial", True)
    ovl, _, prev_o, stats = _run_sequence(pkg, seq, h, n, "overlap", False)
    assert stats == (9, 1), stats
    exact = _compare(pkg, seq, ser, ovl, h, "overlap vs serial")
    assert exact.all()                       # bit for bit, every robot of every tick
    assert np.array_equal(prev_s[exact], prev_o[exact]) and np.abs(prev_s - prev_o).max() <= 1e-6
    last, b = ovl[-1], seq[-1]
    f, tau, st, sec, prev, qdes = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"],
                                                    b["fb_state"], b["wbc_cmd"], prevs[-2].copy(), nthreads=8, epilogue=3, want_qdes=True)
    ok = (G.flags(last["status"]) == 0) & (st == 0)
    assert ok.mean() > 0.99
    assert np.all(np.abs(last["tau"][ok] - tau[ok]) <= G.tau_tol(tau[ok], 1e-4)), np.abs(last["tau"][ok] - tau[ok]).max()
    assert np.abs(last["force"][ok] - f[ok]).max() <= 1e-5 * max(1.0, np.abs(f[ok]).max())


def test_overlapped_ticks_with_robots_on_the_list_launches(pkg, oracle):
    """A population with plenty of robots beyond the main pass's rows (30 % all stance, 1.5 x the ranges).  The first ticks are chained, their
    rescued robots re-solved by the half-CU list kernel (S^-1 in the global scratch) with the robots' WBC workgroups waiting on through the
    "on the list pass" flag for it (an overlapped tick has no second WBC pass); then a lane finds a plan and the context goes back to the plain
    pipelined tick (qrgpu_api.hip: QRGPU_OV_PLAN_HOLD).  Every robot of every tick within the solver's tolerance of the serial tick, no robot
    unsolved (poisoned outputs), nobody timed out, the last tick against the oracle."""
    h, n = 10, 512
    seq = pkg.make_batch_sequence(n, h, "a1", seed=0x91BE, steps=6, frac_all_stance=0.3, excite=1.5)
    ser, prevs, prev_s, _ = _run_sequence(pkg, seq, h, n, "serial", True)
    ovl, _, prev_o, stats = _run_sequence(pkg, seq, h, n, "overlap", False)
    assert stats[0] >= 1 and stats[0] + stats[1] <= 6, stats
    for k, (x, y) in enumerate(zip(ser, ovl)):
        assert np.all(np.isfinite(y["tau"])) and np.all(np.isfinite(y["force"])), k
        assert np.all(G.flags(y["status"]) & 0x02000000 == 0), k
        ok = (G.flags(x["status"]) == 0) & (G.flags(y["status"]) == 0)
        assert ok.mean() > 0.97, (k, ok.mean())
        ef = np.abs(x["force"] - y["force"]).max(1) / np.maximum(1.0, np.abs(x["force"]).max(1))
        et = (np.abs(x["tau"] - y["tau"]) / np.maximum(1.0, np.abs(x["tau"]))).max(1)
        assert ef[ok].max() <= 1e-6 and et[ok].max() <= 1e-5, (k, ef[ok].max(), et[ok].max())
    last, b = ovl[-1], seq[-1]
    f, tau, st, sec, prev, qdes = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"],
                                                    b["fb_state"], b["wbc_cmd"], prevs[-2].copy(), nthreads=8, epilogue=3, want_qdes=True)
    ok = (G.flags(last["status"]) == 0) & (st == 0)
    assert np.all(np.abs(last["tau"][ok] - tau[ok]) <= G.tau_tol(tau[ok], 1e-4)), np.abs(last["tau"][ok] - tau[ok]).max()


def test_a_tick_that_reuses_its_predecessors_outputs_is_not_chained(pkg):
    """The caller's side of the contract is checked by the library: ticks that write the SAME output arrays as their predecessor wait for the
    context's stream (an event) instead of overlapping -- the outputs of tick t must not be overwritten under a consumer queued behind tick t
    -- and give the serial tick's results; so does a tick that follows another launch of the context."""
    h, n = 10, 256
    seq = pkg.make_batch_sequence(n, h, "a1", seed=0x0E1B, steps=4, excite=1.0)
    ser, _, prev_s, _ = _run_sequence(pkg, seq, h, n, "serial", True, same_outputs=True)
    ovl, _, prev_o, stats = _run_sequence(pkg, seq, h, n, "overlap", True, same_outputs=True)
    assert stats == (0, 4), stats
    _compare(pkg, seq, ser, ovl, h, "unchained overlap form vs serial")
    # own arrays, but an MPC-only launch of the context between the ticks: the tick behind it is not chained
    ctx = pkg.Context(0, 1024, 16)
    try:
        G.setup_a1(ctx, pkg, h)
        assert ctx.set_tick_overlap(True)
        b = seq[0]
        G.run_tick(ctx, pkg, b); G.run_tick(ctx, pkg, b)
        G.run_mpc(ctx, pkg, b)
        G.run_tick(ctx, pkg, b)
        assert ctx.tick_overlap_stats()[0] == 0              # (run_tick syncs and frees its arrays: nothing is ever chained here)
    finally:
        ctx.close()


_GIVE_UP = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import conftest, gpu_helpers as G, oracle_py as oracle
pkg = conftest.load_pkg()
from test_gpu_overlap import _run_sequence
h, n = int(os.environ.get("QR_TEST_H", 10)), 1024
seq = pkg.make_batch_sequence(n, h, "a1", seed=0x0E1C, steps=6, excite=1.0)
ovl, _, prev_o, stats = _run_sequence(pkg, seq, h, n, "overlap", False, planned=(h > 11))        # (h > 11 overlaps only with the list launches on)
assert stats == (5, 1), stats
timed_out = 0
for k, (y, b) in enumerate(zip(ovl, seq)):
    to = (G.flags(y["status"]) & 0x02000000) != 0
    timed_out += int(to.sum())
    assert to.all() if k > 0 else not to.any()        # the first tick waits for nobody; every robot of a chained tick ran into the bound
    assert np.all(np.isfinite(y["tau"]))
    # the MPC side of a robot whose wait gave up started cold: its forces are still the optimum (the guess is speed only)
    f, tau, st, sec, prev = oracle.tick_batch(0, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"],
                                              b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=8)
    ok = ((G.flags(y["status"]) & ~0x02000000) == 0) & (st == 0)
    assert ok.mean() > 0.99
    assert np.abs(y["force"][ok] - f[ok]).max() <= 1e-5 * max(1.0, np.abs(f[ok]).max()), k
print("TIMED_OUT", timed_out)
assert timed_out > 0
"""


def test_per_robot_waits_that_give_up_flag_their_robots_and_start_cold():
    """Fault injection (a process of its own: the switches are read once): QRGPU_OV_FAULT=1 makes every chained tick wait for an epoch that nobody
    writes, QRGPU_OV_WAIT_US=50 bounds the waits at 50 us.  Every solve of a chained tick then gives up, starts from the empty working set and
    carries QRGPU_ST_PIPE_TIMEOUT -- never silent -- and its forces are still the QP's optimum (the guess is speed only); every WBC workgroup's
    wait for the robot's previous pass gives up and flags the robot likewise.  (With the real epochs and a bound of 1 us no robot of this
    sequence ever runs into it: a robot's previous solve has long ended when its next one reaches the warm-start words.)"""
    env = dict(os.environ, QRGPU_OV_WAIT_US="50", QRGPU_OV_FAULT="1", GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-c", _GIVE_UP % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "TIMED_OUT" in r.stdout


def test_per_robot_waits_that_give_up_at_h16():
    """The same fault injection on the split machine of h > 11 (lanes 3 / 4): every solve of a chained tick gives up its wait, starts cold and is
    flagged; forces still the optimum."""
    env = dict(os.environ, QRGPU_OV_WAIT_US="50", QRGPU_OV_FAULT="1", GPU_MAX_HW_QUEUES="8", QR_TEST_H="16")
    r = subprocess.run([sys.executable, "-c", _GIVE_UP % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "TIMED_OUT" in r.stdout


_SHARED_QUEUE = r"""
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import conftest, gpu_helpers as G
pkg = conftest.load_pkg()
ctx = pkg.Context(0, 256, 16)
G.setup_a1(ctx, pkg, 10)
on = ctx.set_tick_overlap(True, strict=False)
print("OVERLAP", on, ctx.last_error())
b = pkg.make_batch(256, 10, "a1", seed=5)
out = G.run_tick(ctx, pkg, b)
assert (G.flags(out["status"]) == 0).all()
ctx.close()
"""


def test_overlap_is_refused_when_streams_share_a_hardware_queue():
    """GPU_MAX_HW_QUEUES=1: every stream of the process on one hardware queue.  A launch that polls for a launch on another stream then sits
    out its bound, so qrgpu_set_tick_overlap probes for it and refuses the mode (QRGPU_ERR_NOT_SETUP, ticks as before)."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    r = subprocess.run([sys.executable, "-c", _SHARED_QUEUE % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OVERLAP False" in r.stdout and "hardware queue" in r.stdout


_OV16 = r"""
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import conftest, gpu_helpers as G
import test_gpu_overlap as T
pkg = conftest.load_pkg()
h, n = 16, 1024
seq = pkg.make_batch_sequence(n, h, "a1", seed=0x16AB, steps=8, excite=1.0)
ser, prevs, prev_s, _ = T._run_sequence(pkg, seq, h, n, "piped", True)
ovl, _, prev_o, stats = T._run_sequence(pkg, seq, h, n, "overlap", False)
print("STATS", stats)
assert stats[0] >= 5, stats
listed = 0
for k, (x, y) in enumerate(zip(ser, ovl)):
    assert np.all(np.isfinite(y["tau"])) and np.all(np.isfinite(y["force"])), k
    assert np.all(G.flags(y["status"]) & 0x02000000 == 0), (k, int((G.flags(y["status"]) & 0x02000000 != 0).sum()))
    ok = (G.flags(x["status"]) == 0) & (G.flags(y["status"]) == 0)
    assert ok.mean() > 0.97, (k, ok.mean())
    ef = np.abs(x["force"] - y["force"]).max(1) / np.maximum(1.0, np.abs(x["force"]).max(1))
    et = (np.abs(x["tau"] - y["tau"]) / np.maximum(1.0, np.abs(x["tau"]))).max(1)
    assert ef[ok].max() <= 1e-6 and et[ok].max() <= 1e-5, (k, ef[ok].max(), et[ok].max())
    listed += int((np.asarray(seq[k]["gait"]).reshape(n, -1).sum(1) >= 43).sum())
assert listed > 0
assert np.abs(prev_s - prev_o).max() <= 1e-6
print("OV16_OK")
"""


def test_h16_overlapped_ticks_match_the_plain_tick():
    """Overlapped ticks at h = 16 on the machine split by CU masks (a process of its own: it owns another stream set than the tests above): the main
    pass two to a CU on 192 CUs, the big class (>= 43 stance leg-steps: no room for its inverse Hessian in half a CU) on 64 reserved ones, where the
    tick's planned launch also takes what the main pass hands on (robots that changed class since the lane's plan, working sets that outgrew the
    main pass).  Eight ticks queued without a sync against the plain pipelined tick one at a time: chained, nobody timed out, nobody unsolved,
    every robot within the solver's tolerance."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-c", _OV16 % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OV16_OK" in r.stdout


_OV16_MIXED = r"""
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import conftest, gpu_helpers as G
pkg = conftest.load_pkg()
sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import oracle_py as O
O.build()
h, n, K = 16, 1024, 10
sa = pkg.make_batch_sequence(n // 2, h, "a1", seed=0xC4A1, steps=K, excite=1.0)
sl = pkg.make_batch_sequence(n // 2, h, "lite3", seed=0xC4D2, steps=K, excite=1.0)
seq = []
for ba, bl in zip(sa, sl):
    b = dict(ba); b["n"] = n
    for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
        b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
    seq.append(b)
tid = pkg.shard.interleave_types(n, 2)
ctx = pkg.Context(0, n, 16)
ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h); ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
ctx.set_torque_epilogue(True, True)
assert ctx.set_tick_overlap(True)
S = pkg.to_soa
d_type = ctx.alloc((n,), np.int32).upload(tid)
prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
bufs = [dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
             fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])),
             force=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)),
             status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32))) for b in seq]
ctx.sync()
for d in bufs:
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], prev, d["force"], d["tau"], d["status"], d_type)
ctx.sync()
stats = ctx.tick_overlap_stats()
print("STATS", stats)
assert stats[0] >= K - 3, stats
desc = {0: pkg.model_desc("a1"), 1: pkg.model_desc("lite3")}
cfg = {0: pkg.mpc_cfg("a1"), 1: pkg.mpc_cfg("lite3")}
for k in (K - 2, K - 1):
    d, b = bufs[k], seq[k]
    status = d["status"].download(); to = d["tau"].download().T; fo = d["force"].download().T
    assert np.all(np.isfinite(to)) and np.all(np.isfinite(fo)), k
    assert np.all(G.flags(status) & 0x02000000 == 0), (k, int((G.flags(status) & 0x02000000 != 0).sum()))
    prev_in = seq[k - 1]["wbc_cmd"][:, 12:15]
    for t in (0, 1):
        sel = np.nonzero(tid == t)[0]
        f, tau, st, sec, pv = O.tick_batch(1, cfg[t], h, desc[t][:3], desc[t], b["mpc_state"][sel], b["traj"][sel], b["gait"][sel], b["fb_state"][sel], b["wbc_cmd"][sel],
                                           np.ascontiguousarray(prev_in[sel], np.float32).copy(), nthreads=16, epilogue=3)
        ok = (G.flags(status[sel]) == 0) & (st == 0)
        assert ok.mean() > 0.98, (k, t, ok.mean())
        assert np.all(np.abs(to[sel][ok] - tau[ok]) <= G.tau_tol(tau[ok], 1e-4)), (k, t, np.abs(to[sel][ok] - tau[ok]).max())
        assert np.abs(fo[sel][ok] - f[ok]).max() <= 1e-5 * max(1.0, np.abs(f[ok]).max()), (k, t)
print("OV16_MIXED_OK")
"""


def test_configs4_shard_overlapped_ticks_against_the_oracle():
    """BASELINE configs[4]'s per-GPU shard (512 A1 + 512 Lite3, h = 16), ten ticks of a coherent sequence queued without a sync on the overlapped
    form: chained, no robot timed out or unsolved, the last two ticks -- every robot, both types -- against the oracle at the metric's tolerance."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-c", _OV16_MIXED % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OV16_MIXED_OK" in r.stdout


_OV16_GIVE_UP = r"""
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import conftest, gpu_helpers as G
import test_gpu_overlap as T
pkg = conftest.load_pkg()
h, n = 16, 1024
seq = pkg.make_batch_sequence(n, h, "a1", seed=0x16AC, steps=6, excite=1.0)
ser, prevs, prev_s, _ = T._run_sequence(pkg, seq, h, n, "piped", True)
ovl, _, prev_o, stats = T._run_sequence(pkg, seq, h, n, "overlap", False)
print("STATS", stats)
lost = 0
for k, (x, y) in enumerate(zip(ser, ovl)):
    fl = G.flags(y["status"])
    silent = (fl & 0x02000000) == 0
    # a robot without the time-out flag was solved: its torque is the plain tick's
    ok = silent & (G.flags(x["status"]) == 0) & (fl == 0)
    et = (np.abs(x["tau"] - y["tau"]) / np.maximum(1.0, np.abs(x["tau"]))).max(1)
    assert np.all(np.isfinite(y["tau"][silent])), k
    big = np.asarray(seq[k]["gait"]).reshape(n, -1).sum(1) >= 43
    bad = np.nonzero(ok & (et > 1e-5))[0]
    assert bad.size == 0, (k, bad[:8].tolist(), et[bad[:8]].tolist(), big[bad[:8]].tolist(), [int(np.asarray(seq[j]["gait"]).reshape(n, -1).sum(1)[bad[0]]) for j in range(len(seq))] if bad.size else None,
                           [hex(int(G.flags(ovl[j]["status"])[bad[0]])) for j in range(len(seq))] if bad.size else None, G.iterations(y["status"])[bad[:8]].tolist(), G.iterations(x["status"])[bad[:8]].tolist())
    assert (silent & (fl != 0) & (G.flags(x["status"]) == 0)).sum() <= 2, k          # (nobody is flagged otherwise who is not flagged on the plain tick)
    if k < 2:
        # the lanes' first ticks have no plan: the whole big class arrives unannounced, is handed on, and nobody takes it
        assert (~silent)[big].all(), (k, int((~silent)[big].sum()), int(big.sum()))
    lost += int((~silent).sum())
print("LOST", lost)
assert lost > 0
"""


def test_h16_hand_overs_nobody_takes_are_never_silent():
    """Fault injection (QRGPU_OV_FAULT=2, a process of its own): the planned launch's workgroups go home without waiting for the main pass, as if every
    lingering workgroup had run into its bound (MpcLaunch::main_done).  A robot the main pass hands on then -- in the lanes' first ticks the whole
    big class, later a robot that changed class -- is solved by nobody in that tick: its WBC workgroup's wait for the "on the list" flag gives up
    (QRGPU_OV_WAIT_US), the robot carries QRGPU_ST_PIPE_TIMEOUT, and every robot that does NOT carry it has the plain tick's torque."""
    env = dict(os.environ, QRGPU_OV_FAULT="2", QRGPU_OV_WAIT_US="3000", GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-c", _OV16_GIVE_UP % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "LOST" in r.stdout
