"""GPU: the native RCCL all-gather of the C ABI (qrgpu_comm_* / qrgpu_allgather_tau), single rank (the box has one GPU): communicator
from an id blob, asynchronous gather on the context's own stream behind the tick, double-buffer fence.  N > 1 is the same code with
nranks > 1; its host side (id exchange, shard arithmetic, padded split) runs as a world-size-2 gloo test in test_workload_shard.py."""
import numpy as np
import pytest

import gpu_helpers as G

pytestmark = pytest.mark.gpu


def test_allgather_tau_single_rank(pkg, oracle):
    ctx = pkg.Context(device_id=0, max_batch=256, horizon_max=16)
    try:
        G.setup_a1(ctx, pkg, 10)
        blob = pkg.qrgpu.comm_unique_id()
        assert len(blob) == 128
        ctx.comm_init_rank(blob, 1, 0)
        with pytest.raises(pkg.QrgpuError, match="BAD_ARG"):
            ctx.comm_init_rank(blob, 1, 0)                       # one communicator per context
        n, h = 256, 10
        b = pkg.make_batch(n, h, "a1", seed=31)
        S = pkg.to_soa
        d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])),
                 gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])), fb=ctx.alloc((37, n)).upload(S(b["fb_state"])),
                 cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
                 force=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32))
        tau = [ctx.alloc((12, n)), ctx.alloc((12, n))]
        tau_all = ctx.alloc((1, 12, n)).upload(np.full((1, 12, n), np.nan, np.float32))
        # three ticks, gathers overlapped: tick i+1 is queued while gather i runs; a buffer is fenced before it is overwritten
        for i in range(3):
            slot = i & 1
            ctx.allgather_fence(slot)
            d["prev"].upload(S(b["prev_ori_vel"]))
            ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], tau[slot], d["status"])
            ctx.allgather_tau(tau[slot], n, tau_all, slot)
        ctx.comm_sync()
        ctx.sync()
        got = tau_all.download()[0]
        assert np.array_equal(got, tau[0].download())
        _, tau_o, st, _, _ = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"],
                                               b["gait"], b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=4)
        assert np.all(np.abs(got.T - tau_o) <= G.tau_tol(tau_o, 1e-4))
        # a consumer on the compute stream: qrgpu_allgather_wait makes that stream wait for the gather (no host-side comm sync), and
        # the slot's fence is still due afterwards
        tau_all.upload(np.full((1, 12, n), np.nan, np.float32))
        ctx.allgather_fence(1)
        d["prev"].upload(S(b["prev_ori_vel"]))
        ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], tau[1], d["status"])
        ctx.allgather_tau(tau[1], n, tau_all, 1)
        ctx.allgather_wait(1)
        ctx.sync()                                                   # compute stream only
        assert np.array_equal(tau_all.download()[0], tau[1].download())
        ctx.allgather_fence(1)
        # ten ticks queued with no host sync in between (the host far ahead of the GPU: every fence finds its gather still pending and has to
        # make the compute stream wait for it), two batches alternating so that an overwritten source buffer would show in the gathered copy
        b2 = pkg.make_batch(n, h, "a1", seed=32)
        d2 = dict(state=ctx.alloc((28, n)).upload(S(b2["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b2["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b2["gait"])),
                  fb=ctx.alloc((37, n)).upload(S(b2["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b2["wbc_cmd"])))
        alls = [ctx.alloc((1, 12, n)).upload(np.full((1, 12, n), np.nan, np.float32)) for _ in range(10)]
        pin_prev = ctx.alloc_pinned((3, n)); pin_prev.array[...] = S(b["prev_ori_vel"])
        ctx.sync()
        with G.cold_start(ctx):                                   # (no warm start, no planned list, the orientation task's memory put back before every tick:
            for i in range(10):                                   #  a tick on a batch then gives the same bits whatever ran before it)
                slot = i & 1
                src = d if i % 3 else d2
                ctx.allgather_fence(slot)
                d["prev"].copy_from_pinned(pin_prev)
                ctx.tick_batch(n, src["state"], src["traj"], src["gait"], src["fb"], src["cmd"], d["prev"], d["force"], tau[slot], d["status"])
                ctx.allgather_tau(tau[slot], n, alls[i], slot, of_tick=(i >= 4))      # (the last six: the gather waits for its tick's join, not for an event)
            ctx.comm_sync()
            ctx.sync()
        ga = [a.download()[0] for a in alls]
        assert all(np.all(np.isfinite(g)) for g in ga)
        assert np.array_equal(ga[9], tau[1].download()) and np.array_equal(ga[8], tau[0].download())
        for i in range(10):                                       # every gathered copy is its own tick's torques
            for j in range(i + 1, 10):
                same = (i % 3 == 0) == (j % 3 == 0)
                assert np.array_equal(ga[i], ga[j]) if same else (np.abs(ga[i] - ga[j]).max() > 0.5), (i, j)
        ctx.comm_destroy()
        with pytest.raises(pkg.QrgpuError, match="NOT_SETUP"):
            ctx.allgather_tau(tau[0], n, tau_all, 0)
    finally:
        ctx.close()
