"""-m gpu: the HIP WBC kernel (through the C ABI) against the CPU oracle on identical inputs.

The kernel evaluates the reference's formulas in fp64 on the fp32 inputs; the oracle is run both
as written in the reference (fp32, T=float) and in fp64 (T=double).  Bars:
  * vs the fp64 oracle: 1e-6 * max(1,|tau|)   (only the float32 output rounding is left)
  * vs the fp32 oracle: 1e-4 * max(1,|tau|)   (north_star tolerance; the gap is the fp32 oracle's own rounding)
"""
import numpy as np
import pytest

import gpu_helpers as G

pytestmark = pytest.mark.gpu


def _oracle_wbc(oracle, pkg, b, cmd=None, prev=None, dtype=np.float64, robot="a1"):
    md = pkg.model_desc(robot)
    n = b["n"]
    cmd = b["wbc_cmd"] if cmd is None else cmd
    prev = b["prev_ori_vel"] if prev is None else prev
    tau = np.zeros((n, 12)); qdes = np.zeros((n, 12)); qddes = np.zeros((n, 12)); nact = np.zeros(n, int)
    for i in range(n):
        r = oracle.wbc_run(md, b["fb_state"][i].astype(dtype), cmd[i].astype(dtype), prev[i].astype(dtype), dtype=dtype)
        assert r["rc"] == 0
        tau[i], qdes[i], qddes[i], nact[i] = r["tau"], r["qdes"], r["qddes"], r["qp"]["n_active"]
    return tau, qdes, qddes, nact


def test_rigid_body_quantities(gpu_ctx, pkg, oracle):
    """K8-K10: mass matrix, gravity, Coriolis, foot Jacobians, Jdot*qdot, foot positions/velocities."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(32, 10, "a1", seed=41)
    got = G.run_fb_debug(gpu_ctx, pkg, b)
    md = pkg.model_desc("a1")
    for i in range(b["n"]):
        r = oracle.fb_compute(md, b["fb_state"][i].astype(np.float64), np.float64)
        for k, tol in (("H", 2e-6), ("G", 2e-5), ("C", 2e-6), ("Jc", 1e-6), ("Jcdqd", 2e-5), ("pGC", 1e-6), ("vGC", 1e-6)):
            err = np.abs(got[k][i] - r[k]).max()
            assert err <= tol * max(1.0, np.abs(r[k]).max()), (k, i, err)


@pytest.mark.parametrize("seed,n", [(51, 256)])
def test_wbc_parity(gpu_ctx, pkg, oracle, seed, n):
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(n, 10, "a1", seed=seed)
    out = G.run_wbc(gpu_ctx, pkg, b)
    assert np.all(out["status"] == 0), np.unique(out["status"])
    tau64, qd64, qdd64, _ = _oracle_wbc(oracle, pkg, b, dtype=np.float64)
    tau32, qd32, qdd32, _ = _oracle_wbc(oracle, pkg, b, dtype=np.float32)
    assert np.all(np.abs(out["tau"] - tau64) <= G.tau_tol(tau64, 1e-6)), np.abs(out["tau"] - tau64).max()
    assert np.all(np.abs(out["tau"] - tau32) <= G.tau_tol(tau32, 1e-4)), np.abs(out["tau"] - tau32).max()
    assert np.abs(out["qdes"] - qd64).max() <= 1e-5 and np.abs(out["qddes"] - qdd64).max() <= 1e-4
    # quirk 4: prev_ori_vel <- this tick's vBody_Ori_des
    assert np.array_equal(out["prev"], b["wbc_cmd"][:, 12:15])


def test_wbc_friction_active_and_stateful(gpu_ctx, pkg, oracle):
    """Fr_des on/outside the WBC pyramid (mu = 0.4 < MPC's 0.45) makes the relaxation QP's inequalities
    bind; a non-zero prev_ori_vel exercises the stateful orientation-rate quirk."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(64, 10, "a1", seed=61)
    rng = np.random.default_rng(5)
    cmd = b["wbc_cmd"].copy()
    fr = cmd[:, 51:63].reshape(-1, 4, 3)
    fr[:, :, 0] = 0.45 * fr[:, :, 2] * rng.choice([-1, 1], size=fr[:, :, 0].shape)
    fr[:, :, 1] = 0.3 * fr[:, :, 2]
    fr[::4, :, 2] *= 3.0                      # above maxFz for every 4th robot
    cmd[:, 51:63] = fr.reshape(-1, 12)
    prev = rng.uniform(-0.5, 0.5, (64, 3)).astype(np.float32)
    out = G.run_wbc(gpu_ctx, pkg, b, wbc_cmd=cmd, prev=prev)
    assert np.all(out["status"] == 0)
    tau64, _, _, nact = _oracle_wbc(oracle, pkg, b, cmd=cmd, prev=prev, dtype=np.float64)
    assert nact.max() > 0, "test must exercise active inequalities"
    assert np.all(np.abs(out["tau"] - tau64) <= G.tau_tol(tau64, 1e-6)), np.abs(out["tau"] - tau64).max()


def test_wbc_contact_patterns(gpu_ctx, pkg, oracle):
    """flight (no contact: 6 tasks, dummy inequality), stand (4 contacts, no foot task), 3 and 1 contacts."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(16, 10, "a1", seed=71)
    cmd = b["wbc_cmd"].copy()
    pats = [(0, 0, 0, 0), (1, 1, 1, 1), (1, 1, 1, 0), (0, 1, 0, 0), (1, 0, 0, 1), (0, 1, 1, 0), (1, 0, 1, 1), (0, 0, 1, 1)]
    for i in range(16):
        cmd[i, 63:67] = pats[i % 8]
        cmd[i, 51:63] = (np.array(pats[i % 8], np.float32)[:, None] * np.array([1.0, -2.0, 30.0], np.float32)).reshape(12)
    out = G.run_wbc(gpu_ctx, pkg, b, wbc_cmd=cmd)
    assert np.all(out["status"] == 0)
    tau64, qd64, qdd64, _ = _oracle_wbc(oracle, pkg, b, cmd=cmd, dtype=np.float64)
    assert np.all(np.abs(out["tau"] - tau64) <= G.tau_tol(tau64, 1e-6)), np.abs(out["tau"] - tau64).max()
    assert np.abs(out["qdes"] - qd64).max() <= 1e-5


def test_wbc_straight_knee_rank_cut(gpu_ctx, pkg, oracle):
    """Kinematic singularity (knee straight): pseudoInverse()'s singular-value cut changes the rank;
    the kernel must take its eigen-decomposition path and agree with the SVD-based oracle."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(8, 10, "a1", seed=81)
    fb = b["fb_state"].copy()
    for i in range(8):
        fb[i, 13 + 2] = 0.0 if i % 2 == 0 else 1e-5          # FR knee straight / almost straight
        fb[i, 13 + 1] = 0.3
    b["fb_state"] = fb
    cmd = b["wbc_cmd"].copy()
    cmd[:, 63:67] = (0, 1, 1, 0)                              # FR is a swing leg -> its foot task is rank deficient
    cmd[:, 51:63] = (np.array([0, 1, 1, 0], np.float32)[:, None] * np.array([0.0, 0.0, 60.0], np.float32)).reshape(12)
    out = G.run_wbc(gpu_ctx, pkg, b, wbc_cmd=cmd)
    assert np.all(out["status"] == 0)
    tau64, qd64, qdd64, _ = _oracle_wbc(oracle, pkg, b, cmd=cmd, dtype=np.float64)
    assert np.all(np.abs(out["tau"] - tau64) <= G.tau_tol(tau64, 1e-5)), np.abs(out["tau"] - tau64).max()
    assert np.abs(out["qdes"] - qd64).max() <= 1e-4


def test_wbc_single_robot_interface_cadence(gpu_ctx, pkg, oracle):
    """qrWbcLocomotionController::Run computes on every 2nd call and only overwrites stance legs."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(2, 10, "a1", seed=91)
    wbc = pkg.WbcLocomotionController(gpu_ctx, 0)
    md = pkg.model_desc("a1")
    tua = np.full(12, 7.0, np.float32)
    r0 = oracle.wbc_run(md, b["fb_state"][0].astype(np.float64), b["wbc_cmd"][0].astype(np.float64), dtype=np.float64)
    wbc.Run(b["fb_state"][0], b["wbc_cmd"][0], tua)
    contact = b["wbc_cmd"][0, 63:67]
    for leg in range(4):
        sl = slice(3 * leg, 3 * leg + 3)
        if contact[leg]:
            assert np.all(np.abs(tua[sl] - r0["tau"][sl]) <= G.tau_tol(r0["tau"][sl], 1e-6))
        else:
            assert np.all(tua[sl] == 7.0)
    first = wbc.jointTorqueCmd.copy()
    wbc.Run(b["fb_state"][1], b["wbc_cmd"][1], tua)           # odd call: no recompute
    assert np.array_equal(first, wbc.jointTorqueCmd) and wbc.iteration == 2


def test_full_tick(gpu_ctx, pkg, oracle):
    """Config 3 of BASELINE.json in small: MPC -> Fr_des -> WBC, torque = WBC on stance legs, J^T f on swing legs."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(128, 10, "a1", seed=0xA4)
    out = G.run_tick(gpu_ctx, pkg, b)
    assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
    f, tau, st, sec, prev = oracle.tick_batch(1, pkg.mpc_cfg("a1"), 10, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"],
                                              b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=4)
    assert np.all(st == 0)
    assert np.abs(out["force"] - f).max() <= 1e-5 * max(1.0, np.abs(f).max())
    assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau, 1e-4)), np.abs(out["tau"] - tau).max()
    assert np.array_equal(out["prev"], prev)


def test_full_tick_h16_mixed_a1_lite3(gpu_ctx, pkg, oracle):
    """Config 5 of BASELINE.json in small: horizon 16, A1 and Lite3 interleaved (type_id per robot), full tick."""
    h, n = 16, 64
    gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); gpu_ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
    gpu_ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h); gpu_ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
    ba = pkg.make_batch(n // 2, h, "a1", seed=501); bl = pkg.make_batch(n // 2, h, "lite3", seed=502)
    b = dict(ba)
    for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
        b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
    b["n"] = n
    tid = pkg.shard.interleave_types(n, 2)
    out = G.run_tick(gpu_ctx, pkg, b, type_id=tid)
    assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
    for i in range(n):
        robot = "a1" if tid[i] == 0 else "lite3"
        u, st, rc = oracle.mpc_solve(pkg.mpc_cfg(robot), h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert rc == 0
        assert np.abs(out["force"][i] - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max())
        cmd = b["wbc_cmd"][i].copy(); cmd[51:63] = u[:12].astype(np.float32)
        w = oracle.wbc_run(pkg.model_desc(robot), b["fb_state"][i].astype(np.float64), cmd.astype(np.float64), dtype=np.float64)
        tau = oracle.mpc_force_to_torque(pkg.model_desc(robot)[:3], b["fb_state"][i, :4], b["fb_state"][i, 13:25], u[:12]).astype(np.float64)
        for l in range(4):
            if cmd[63 + l]:
                tau[3 * l:3 * l + 3] = w["tau"][3 * l:3 * l + 3]
        assert np.all(np.abs(out["tau"][i] - tau) <= G.tau_tol(tau, 1e-4))
    G.setup_a1(gpu_ctx, pkg, 10)


def test_full_tick_h16_1024_two_workgroups_per_cu(pkg, oracle):
    """BASELINE.json configs[4] at full size (1024 robots, A1 and Lite3 interleaved, horizon 16): from 3.5 robots per CU on, the h > 11 main pass
    runs two four-wave workgroups per CU on half the LDS each; robots whose inverse Hessian does not fit half a CU (a class known from the gait
    table) and the tick's long poles (smoothed cost) are planned onto whole CUs beside it.  Four ticks on the same inputs -- no plan, the plan
    coming into being, the planned form twice -- every robot of every tick against the threaded oracle; by the last tick the planned list is
    there and nobody is re-solved behind the main pass."""
    import ctypes as C
    h, n = 16, 1024
    ctx = pkg.Context(0, n, h)
    try:
        for r, t in (("a1", 0), ("lite3", 1)):
            ctx.mpc_setup_packed(t, pkg.mpc_cfg(r), h); ctx.wbc_setup_packed(t, pkg.model_desc(r))
        ba = pkg.make_batch(n // 2, h, "a1", seed=1601); bl = pkg.make_batch(n // 2, h, "lite3", seed=1602)
        b = dict(ba)
        for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
            b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
        b["n"] = n
        tid = pkg.shard.interleave_types(n, 2)
        f = np.zeros((n, 12), np.float32); tau = np.zeros((n, 12), np.float32); st = np.zeros(n, np.int32)
        for r, t in (("a1", 0), ("lite3", 1)):
            m = tid == t
            f[m], tau[m], st[m] = oracle.tick_batch(1, pkg.mpc_cfg(r), h, pkg.model_desc(r)[:3], pkg.model_desc(r), b["mpc_state"][m], b["traj"][m],
                                                    b["gait"][m], b["fb_state"][m], b["wbc_cmd"][m], b["prev_ori_vel"][m].copy(), nthreads=8)[:3]
        assert np.all(st == 0)
        lists = np.zeros(8, np.int32); ctx._lib.qrgpu_debug_lists.argtypes = [C.c_void_p, C.c_void_p]
        for tick in range(4):
            out = G.run_tick(ctx, pkg, b, type_id=tid)
            assert np.all(G.flags(out["status"]) == 0), (tick, np.unique(G.flags(out["status"])))
            assert np.all(np.abs(out["force"] - f).max(1) <= 1e-5 * np.maximum(1.0, np.abs(f).max(1))), tick
            assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau, 1e-4)), (tick, np.abs(out["tau"] - tau).max())
            assert ctx._lib.qrgpu_debug_lists(ctx._h, lists.ctypes.data) == 0
        assert lists[2] > 0 and lists[3] > 0, lists            # the planned list, both parities
        assert lists[0] == 0 and lists[1] == 0, lists          # nobody handed to the trailing launch
        # without the list launches, the plan or the cost words there is nowhere for the big class to go: the batch runs one workgroup per CU again
        for off in (lambda: ctx.set_planned_list(False), lambda: ctx.set_rescue_pass(False), lambda: ctx.set_lpt_schedule(False)):
            off()
            for tick in range(2):
                out = G.run_tick(ctx, pkg, b, type_id=tid)
                assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
                assert np.all(np.abs(out["force"] - f).max(1) <= 1e-5 * np.maximum(1.0, np.abs(f).max(1)))
                assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau, 1e-4)), np.abs(out["tau"] - tau).max()
            ctx.set_planned_list(True); ctx.set_rescue_pass(True); ctx.set_lpt_schedule(True)
    finally:
        ctx.close()


@pytest.mark.parametrize("h", [12, 13, 14])
def test_full_tick_two_workgroups_per_cu_other_horizons(pkg, oracle, h):
    """The h > 11 main pass two to a CU at the horizons between the two the bench runs: below h = 14 the 96-position kernels keep the second half
    of the r exchange in 256 doubles at the end of the workgroup's LDS (here: of HALF a CU's), and the class that cannot share a CU starts at a
    different stance count at every horizon.  1024 A1 robots of every gait, three ticks (no plan, the plan coming, planned), all against the oracle."""
    n = 1024
    ctx = pkg.Context(0, n, h)
    try:
        G.setup_a1(ctx, pkg, h)
        b = pkg.make_batch(n, h, "a1", seed=1620 + h)
        f, tau, st = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"],
                                       b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=8)[:3]
        assert np.all(st == 0)
        for tick in range(3):
            out = G.run_tick(ctx, pkg, b)
            assert np.all(G.flags(out["status"]) == 0), (tick, np.unique(G.flags(out["status"])))
            assert np.all(np.abs(out["force"] - f).max(1) <= 1e-5 * np.maximum(1.0, np.abs(f).max(1))), tick
            assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau, 1e-4)), (tick, np.abs(out["tau"] - tau).max())
    finally:
        ctx.close()


def test_h16_standing_shard_goes_back_to_one_workgroup_per_cu_and_returns(pkg, oracle):
    """h = 16, 1024 robots that all stand: all stance is the class that cannot share a CU, so the planned list is the whole batch and the calls go
    back to one workgroup per CU (31 calls, then one call two to a CU on the old plan to get a fresh one, and so on); when the robots start to
    trot the calls return to two workgroups per CU.  Every tick of both phases -- across the switches, the stale plans and the probes -- against
    the threaded oracle."""
    h, n = 16, 1024
    ctx = pkg.Context(0, n, h)
    try:
        G.setup_a1(ctx, pkg, h)
        for phase, kw, ticks in (("standing", dict(frac_all_stance=1.0, frac_three_leg=0.0), 40), ("trotting", dict(frac_all_stance=0.0, frac_three_leg=0.0), 72)):
            b = pkg.make_batch(n, h, "a1", seed=1611, **kw)
            f, tau, st = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"],
                                           b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=8)[:3]
            assert np.all(st == 0)
            for tick in range(ticks):
                out = G.run_tick(ctx, pkg, b)
                assert np.all(G.flags(out["status"]) == 0), (phase, tick, np.unique(G.flags(out["status"])))
                assert np.all(np.abs(out["force"] - f).max(1) <= 1e-5 * np.maximum(1.0, np.abs(f).max(1))), (phase, tick)
                assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau, 1e-4)), (phase, tick, np.abs(out["tau"] - tau).max())
    finally:
        ctx.close()


def test_full_tick_1024_with_projection_and_motor_tail(gpu_ctx, pkg, oracle):
    """BASELINE.json configs[2] at full size: 1024 A1 robots, h = 10, the whole tick of SURVEY 8(d) -- K12 (kinematic projection) on and
    the K14 motor tail (abad +-0.9 N m, +-23 N m clip) applied -- every robot against the threaded oracle."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(1024, 10, "a1", seed=0xA1 + 2)
    f, tau, st, sec, prev, qdes = oracle.tick_batch(1, pkg.mpc_cfg("a1"), 10, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"],
                                                    b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=8,
                                                    epilogue=3, want_qdes=True)
    gpu_ctx.set_torque_epilogue(hip_comp=True, clip=True)
    try:
        out = G.run_tick(gpu_ctx, pkg, b, want_qdes=True)
    finally:
        gpu_ctx.set_torque_epilogue(False, False)
    assert np.all(G.flags(out["status"]) == 0) and np.all(st == 0)
    assert np.abs(out["force"] - f).max() <= 1e-5 * max(1.0, np.abs(f).max())
    assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau, 1e-4)), np.abs(out["tau"] - tau).max()
    assert np.abs(out["tau"]).max() <= 23.0
    # the compensation shows exactly on the abad motors of swing legs (their MPC torque is J^T 0 = 0)
    swing = b["wbc_cmd"][:, 63:67] == 0
    comp = np.array([-0.9, 0.9, -0.9, 0.9], np.float32)
    assert np.array_equal(out["tau"][:, 0::3][swing], np.broadcast_to(comp, swing.shape)[swing])
    # K12 outputs: against the oracle's tick in DOUBLE (the kernel's arithmetic) 1e-5 rad, as test_wbc_golden holds its 16 robots
    # (the fp32 oracle's pseudo-inverses are only good for 2e-3 here)
    assert np.abs(out["qdes"] - qdes).max() <= 2e-3
    md = pkg.model_desc("a1")
    q64 = np.stack([oracle.tick_from_forces(md[:3], md, b["fb_state"][i], b["wbc_cmd"][i], b["prev_ori_vel"][i], f[i].astype(np.float64), 1, 3,
                                            wbc_fp64=True, want_qdes=True)[2] for i in range(1024)])
    assert np.abs(out["qdes"] - q64).max() <= 1e-5, np.abs(out["qdes"] - q64).max()
    assert np.array_equal(out["prev"], prev)


def test_tick_without_tail_is_unchanged_by_projection(gpu_ctx, pkg):
    """K12's outputs do not enter the torque: the tick with and without d_qdes returns the same bits."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(64, 10, "a1", seed=77)
    with G.cold_start(gpu_ctx):
        a = G.run_tick(gpu_ctx, pkg, b, want_qdes=False)
        c = G.run_tick(gpu_ctx, pkg, b, want_qdes=True)
    assert np.array_equal(a["tau"], c["tau"]) and np.array_equal(a["force"], c["force"])


def test_mpc_only_motor_tail(gpu_ctx, pkg, oracle):
    """qrgpu_mpc_solve_batch with the K14 tail: +-0.9 on every abad motor (no WBC overwrite on an MPC tick), then the clip."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(64, 10, "a1", seed=78)
    f, tau, st, sec, prev = oracle.tick_batch(0, pkg.mpc_cfg("a1"), 10, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"],
                                              b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), epilogue=3)
    gpu_ctx.set_torque_epilogue(True, True)
    try:
        out = G.run_mpc(gpu_ctx, pkg, b)
    finally:
        gpu_ctx.set_torque_epilogue(False, False)
    assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau, 1e-4)), np.abs(out["tau"] - tau).max()
    raw = G.run_mpc(gpu_ctx, pkg, b)["tau"]
    assert np.abs((out["tau"] - raw)[:, 1::3]).max() == 0.0 or np.abs(raw).max() > 23.0


def test_unknown_type_is_flagged(gpu_ctx, pkg):
    """A type id outside the table or never set up must not read garbage constants: the robot carries QRGPU_ST_BAD_TYPE."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(8, 10, "a1", seed=79)
    tid = np.array([0, 3, 0, 7, -1, 0, 2, 0], np.int32)            # types 2, 3 were never set up on this context's horizon; 7, -1 do not exist
    ctx2 = pkg.Context(device_id=0, max_batch=8, horizon_max=16)
    try:
        G.setup_a1(ctx2, pkg, 10)
        out = G.run_tick(ctx2, pkg, b, type_id=tid)
    finally:
        ctx2.close()
    bad = (out["status"] & 0x01000000) != 0
    assert np.array_equal(bad, tid != 0)
    with G.cold_start(gpu_ctx):
        ref = G.run_tick(gpu_ctx, pkg, b)
    assert np.array_equal(out["tau"][~bad], ref["tau"][~bad])
    assert np.all(np.isfinite(out["tau"]))


def test_pipelined_tick_is_the_serial_tick_bit_for_bit(gpu_ctx, pkg, oracle):
    """qrgpu_tick_batch queues its WBC launch beside the MPC launches (a stream of the context's own, per-robot flags raised by the solves
    behind write-through stores of force / tau / status; qrgpu_set_tick_pipeline).  Scheduling only: forces, torques, status words, K12
    outputs and the orientation task's memory are those of the serial form bit for bit -- over a temporally coherent sequence (warm starts,
    planned list and dispatch history evolve alike on both sides), with K12 and the K14 tail on, and with robots that the main pass hands to
    its trailing list launch (the second, list-driven WBC pass).  The last tick is also held against the oracle."""
    h, n = 10, 512
    outs = {}
    for piped in (False, True):
        ctx = pkg.Context(0, 1024, 16)            # contexts of their own: the same (empty) history on both sides
        try:
            G.setup_a1(ctx, pkg, h)
            ctx.set_tick_pipeline(piped)
            ctx.set_torque_epilogue(hip_comp=True, clip=True)
            seq = pkg.make_batch_sequence(n, h, "a1", seed=0x91BE, steps=5, frac_all_stance=0.3, excite=1.5)     # plenty of robots beyond the main pass's rows
            S = pkg.to_soa
            d_prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
            res = []
            for b in seq:
                d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])),
                         gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])), fb=ctx.alloc((37, n)).upload(S(b["fb_state"])),
                         cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), force=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)),
                         tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)), qdes=ctx.alloc((24, n)),
                         status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32)))
                ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d_prev, d["force"], d["tau"], d["status"], qdes=d["qdes"])
                ctx.sync()
                res.append(dict(force=d["force"].download().T.copy(), tau=d["tau"].download().T.copy(), status=d["status"].download(),
                                qdes=d["qdes"].download().T.copy(), prev=d_prev.download().T.copy()))
                for v in d.values():
                    v.free()
            outs[piped] = res
        finally:
            ctx.close()
    rescued = 0
    for k, (a, b_) in enumerate(zip(outs[False], outs[True])):
        for key in ("force", "tau", "status", "qdes", "prev"):
            assert np.array_equal(a[key], b_[key]), (k, key)
        assert np.all(G.flags(b_["status"]) & 0x02000000 == 0)              # nobody timed out waiting for its forces
    last, b = outs[True][-1], seq[-1]
    prev_in = outs[True][-2]["prev"]
    f, tau, st, sec, prev, qdes = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"],
                                                    b["fb_state"], b["wbc_cmd"], prev_in.copy(), nthreads=8, epilogue=3, want_qdes=True)
    ok = (G.flags(last["status"]) == 0) & (st == 0)
    assert ok.mean() > 0.97
    assert np.all(np.abs(last["tau"][ok] - tau[ok]) <= G.tau_tol(tau[ok], 1e-4)), np.abs(last["tau"][ok] - tau[ok]).max()
    assert np.array_equal(last["prev"], prev)


def test_pipelined_tick_with_many_robots_on_the_list_pass(gpu_ctx, pkg, oracle):
    """Every robot all-stance at twice the ranges: hundreds outgrow the main pass and go through the trailing list launch, whose robots the
    WBC launch running beside the main pass must leave alone (flag bit 0) for the second, list-driven WBC pass.  No robot may come back
    unsolved (poisoned outputs), and every unflagged one is the oracle's."""
    h, n = 10, 512
    G.setup_a1(gpu_ctx, pkg, h)
    b = pkg.make_batch(n, h, "a1", seed=0xBEE6, excite=2.0, frac_all_stance=1.0, frac_three_leg=0.0)
    gpu_ctx.set_rescue_pass(False)
    try:
        flagged = (G.flags(G.run_mpc(gpu_ctx, pkg, b)["status"]) & 0x4) != 0
    finally:
        gpu_ctx.set_rescue_pass(True)
    assert flagged.sum() > 64
    gpu_ctx.set_planned_list(False)           # (so that the list pass, not the planned launch, keeps getting them)
    try:
        out = G.run_tick(gpu_ctx, pkg, b)
        out = G.run_tick(gpu_ctx, pkg, b)
    finally:
        gpu_ctx.set_planned_list(True)
    assert np.all(np.isfinite(out["tau"])) and np.all(G.flags(out["status"]) & 0x02000000 == 0)
    f, tau, st, sec, prev = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), b["mpc_state"], b["traj"], b["gait"],
                                              b["fb_state"], b["wbc_cmd"], b["prev_ori_vel"].copy(), nthreads=8)
    ok = (G.flags(out["status"]) == 0) & (st == 0)
    assert ok.mean() > 0.9 and (ok & flagged).sum() > 64
    assert np.all(np.abs(out["tau"][ok] - tau[ok]) <= G.tau_tol(tau[ok], 1e-4)), np.abs(out["tau"][ok] - tau[ok]).max()


def test_pipelined_tick_with_a_planned_list_longer_than_the_host_saw_it(pkg, oracle):
    """The planned launch's grid is the host's UNSYNCHRONISED copy of the list's length (plus two): when ticks are queued faster than they run,
    the list planned by the tick before may be much longer, and the launch's last workgroup hands the remainder to the trailing list launch.
    In a pipelined tick those robots are solved by nobody who raises a flag -- the main pass skips them, the trailing launch raises none -- so
    the hand-over itself must tell their WBC workgroups to leave them to the second WBC pass (it did not: they waited out the 4 ms bound and
    came back flagged QRGPU_ST_PIPE_TIMEOUT: 52 of them in this test).  Eight hard all-stance robots in the batch the host has seen planned, 120
    in the two queued behind it without a sync; outputs of the last tick against the oracle, nobody timed out, nobody unsolved."""
    h, n = 10, 512
    ctx = pkg.Context(0, 1024, 16)
    try:
        G.setup_a1(ctx, pkg, h)
        ctx.set_tick_pipeline(True)
        trot = pkg.make_batch(n, h, "a1", seed=0x51A7, frac_all_stance=0.0, frac_three_leg=0.0)
        stance = pkg.make_batch(n, h, "a1", seed=0x51A7, frac_all_stance=1.0, frac_three_leg=0.0, excite=2.0)     # (robots that outgrow the main pass)
        def mix(k):
            b = {key: (v.copy() if isinstance(v, np.ndarray) else v) for key, v in trot.items()}
            idx = np.arange(k) * (n // max(k, 1)) + 3
            for key in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
                b[key][idx] = stance[key][idx]
            return b
        few, many = mix(8), mix(120)                 # (5 and 59 of them end up on the planned list: scratch/diag_stale_plan.py)
        S = pkg.to_soa
        def upload(b):
            return dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])),
                        gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])), fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])))
        dF, dM = upload(few), upload(many)
        d_prev = ctx.alloc((3, n)).upload(S(many["prev_ori_vel"]))
        force, tau, status = ctx.alloc((12, n)), ctx.alloc((12, n)), ctx.alloc((n,), np.int32)
        def tick(d):
            ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d_prev, force, tau, status)
        for _ in range(4):                           # the host has seen a plan of five
            tick(dF); ctx.sync()
        d_prev.upload(S(many["prev_ori_vel"]))
        tau.upload(np.full((12, n), np.nan, np.float32)); status.upload(np.full((n,), 0x7f0000ff, np.int32))
        tick(dM); tick(dM)                           # no sync in between: the second one's planned launch is sized from a stale length
        ctx.sync()
        st_gpu, tau_gpu = status.download(), tau.download().T.copy()
    finally:
        ctx.close()
    assert np.all(G.flags(st_gpu) & 0x02000000 == 0), "a WBC workgroup waited for forces nobody flagged"
    assert np.all(np.isfinite(tau_gpu))
    # the second tick's inputs: the same batch, the orientation task's memory as the first tick left it (desiredVel of the command: wbc_cmd[12:15])
    f, tau_o, st, sec, prev = oracle.tick_batch(1, pkg.mpc_cfg("a1"), h, pkg.model_desc("a1")[:3], pkg.model_desc("a1"), many["mpc_state"], many["traj"], many["gait"],
                                                many["fb_state"], many["wbc_cmd"], many["wbc_cmd"][:, 12:15].copy(), nthreads=8)
    ok = (G.flags(st_gpu) == 0) & (st == 0)
    assert ok.mean() > 0.9
    assert np.all(np.abs(tau_gpu[ok] - tau_o[ok]) <= G.tau_tol(tau_o[ok], 1e-4)), np.abs(tau_gpu[ok] - tau_o[ok]).max()


def test_pipelined_ticks_queued_without_a_sync_match_the_serial_ticks(pkg):
    """A dozen ticks of a temporally coherent sequence queued back to back with no host sync in between -- how `bench.py` and a simulator's step
    loop drive the library -- on the pipelined form, against the same sequence on the serial form with a sync after every tick.  The host then
    sizes the planned launches from list lengths several ticks old, the gates, the polled "go" and both polled joins all run against a GPU
    that is behind the host, and robots move between the main pass, the planned launch and the trailing launch from tick to tick (which changes
    the last bits of a solve: the comparison is at the solver's tolerance, not bitwise).  Every tick's outputs in buffers of their own."""
    h, n, T = 10, 512, 12
    seq = pkg.make_batch_sequence(n, h, "a1", seed=0x7E57, steps=T, frac_all_stance=0.25, excite=1.5)
    S = pkg.to_soa
    res = {}
    for piped in (False, True):
        ctx = pkg.Context(0, 1024, 16)
        try:
            G.setup_a1(ctx, pkg, h)
            ctx.set_tick_pipeline(piped)
            ctx.set_torque_epilogue(hip_comp=True, clip=True)
            d_prev = ctx.alloc((3, n)).upload(S(seq[0]["prev_ori_vel"]))
            ins = [dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
                        fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"]))) for b in seq]
            outs = [dict(force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)),
                         status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32)), qdes=ctx.alloc((24, n))) for _ in seq]
            ctx.sync()
            for i, o in zip(ins, outs):
                ctx.tick_batch(n, i["state"], i["traj"], i["gait"], i["fb"], i["cmd"], d_prev, o["force"], o["tau"], o["status"], qdes=o["qdes"])
                if not piped:
                    ctx.sync()
            ctx.sync()
            res[piped] = [dict(tau=o["tau"].download().T.copy(), status=o["status"].download(), qdes=o["qdes"].download().T.copy()) for o in outs]
        finally:
            ctx.close()
    for k, (a, b_) in enumerate(zip(res[False], res[True])):
        assert np.all(G.flags(b_["status"]) & 0x02000000 == 0), k
        assert np.all(np.isfinite(b_["tau"])), k
        # (a robot at the edge of the 96 working-set positions may be flagged on one side only: which launch solves it moves with the plan)
        assert int((G.flags(a["status"]) != G.flags(b_["status"])).sum()) <= 2, k
        ok = (G.flags(a["status"]) == 0) & (G.flags(b_["status"]) == 0)
        assert ok.mean() > 0.9
        assert np.all(np.abs(a["tau"][ok] - b_["tau"][ok]) <= G.tau_tol(a["tau"][ok], 1e-4)), (k, np.abs(a["tau"][ok] - b_["tau"][ok]).max())
        assert np.abs(a["qdes"][ok] - b_["qdes"][ok]).max() < 1e-5, k


def test_pipelined_ticks_of_changing_batch_size_queued_without_a_sync(pkg):
    """Ticks of 1024, 256, 1024, 64 (the smallest pipelined size), 1024, 96 and 1024 robots queued back to back on one context without a sync: the
    gates, joins and flags are cumulative counts and epochs that must survive a change of the batch size from one call to the next (grids,
    expected counts, the planned and rescue lists' histories all change).  Against the same calls on the serial form, at the solver's tolerance."""
    h = 10
    sizes = [1024, 256, 1024, 64, 1024, 96, 1024]
    batches = [pkg.make_batch(n, h, "a1", seed=0xB5 + k, frac_all_stance=0.15, excite=1.3) for k, n in enumerate(sizes)]
    S = pkg.to_soa
    res = {}
    for piped in (False, True):
        ctx = pkg.Context(0, 1024, 16)
        try:
            G.setup_a1(ctx, pkg, h)
            ctx.set_tick_pipeline(piped)
            io = []
            for b in batches:
                n = b["n"]
                io.append(dict(n=n, state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
                               fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
                               force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)),
                               status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32))))
            ctx.sync()
            for d in io:
                ctx.tick_batch(d["n"], d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], d["tau"], d["status"])
                if not piped:
                    ctx.sync()
            ctx.sync()
            res[piped] = [(d["tau"].download().T.copy(), d["status"].download()) for d in io]
        finally:
            ctx.close()
    for k, ((ta, sa), (tb, sb)) in enumerate(zip(res[False], res[True])):
        assert np.all(np.isfinite(tb)) and np.all(G.flags(sb) & 0x02000000 == 0), k
        assert int((G.flags(sa) != G.flags(sb)).sum()) <= 2, k
        ok = (G.flags(sa) == 0) & (G.flags(sb) == 0)
        assert ok.mean() > 0.9, k
        assert np.all(np.abs(ta[ok] - tb[ok]) <= G.tau_tol(ta[ok], 1e-4)), (k, np.abs(ta[ok] - tb[ok]).max())


_GATE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
h, n = 10, 256
ctx = pkg.Context(0, 1024, 16)
G.setup_a1(ctx, pkg, h)
b = pkg.make_batch(n, h, "a1", seed=0x6A7E)
for _ in range(4): ref = G.run_tick(ctx, pkg, b)       # (the first calls of a new batch size end with a stream sync: past those)
S = pkg.to_soa
d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
         fb=ctx.alloc((37, n)), cmd=ctx.alloc((67, n)), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
         force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)).upload(np.full((12, n), np.nan, np.float32)), status=ctx.alloc((n,), np.int32).upload(np.full((n,), 0x7f0000ff, np.int32)))
big = ctx.alloc((1 << 28,))                       # 1 GiB: a zero fill of it is a quarter of a millisecond of the context's stream
pin_fb = ctx.alloc_pinned((37, n)); pin_fb.array[...] = S(b["fb_state"])
pin_cmd = ctx.alloc_pinned((67, n)); pin_cmd.array[...] = S(b["wbc_cmd"])
d["fb"].zero(); d["cmd"].zero()                   # what a WBC launch that ran too early would read
ctx.sync()
for _ in range(200):
    big.zero()
d["fb"].copy_from_pinned(pin_fb); d["cmd"].copy_from_pinned(pin_cmd)      # the caller's producer, on the context's stream, behind ~50 ms of its other work
ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], d["tau"], d["status"])
ctx.sync()
st = d["status"].download(); tau = d["tau"].download().T
print("result", int((G.flags(st) != 0).sum()), int((G.flags(ref["status"]) != 0).sum()), int(np.isfinite(tau).all()), float(np.abs(tau - ref["tau"]).max()))
"""


def test_wbc_gate_that_gives_up_turns_the_tick_into_the_serial_one(pkg):
    """The WBC launch of a pipelined tick waits behind a gate for the tick's main pass to be running -- on a stream of its own, without an event
    from the context's stream -- and that is also what keeps it from reading its inputs before the caller's earlier work on the context's
    stream has produced them.  The gate's wait is bounded (50 ms).  Should it give up, the launch must not run at all (it would compute from
    inputs that are not there yet and write its torques before the solves write theirs: round 3 had this hole behind a 10 ms bound): its
    workgroups leave and the second pass, behind the MPC launches on the context's stream, computes every robot.  A process of its own with the
    bound at 1 ms (QRGPU_PIPE_GATE_MS), ~50 ms of fills queued in front of the tick, and the WBC's inputs produced by the last copies in that queue."""
    import subprocess, sys, os
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, QRGPU_PIPE_GATE_MS="1")
    r = subprocess.run([sys.executable, "-c", _GATE_SCRIPT, here], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [l.split()[1:] for l in r.stdout.strip().splitlines() if l.startswith("result")][0]
    assert int(res[0]) == 0 and int(res[1]) == 0 and int(res[2]) == 1, res
    assert float(res[3]) < 2e-4, res              # (the second tick starts from the first one's working sets: the same optimum to the solver's tolerance)


_PLAN_GO_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
h, n = 10, 512
ctx = pkg.Context(0, 1024, 16)
G.setup_a1(ctx, pkg, h)
trot = pkg.make_batch(n, h, "a1", seed=0x51A7, frac_all_stance=0.0, frac_three_leg=0.0)
stance = pkg.make_batch(n, h, "a1", seed=0x51A7, frac_all_stance=1.0, frac_three_leg=0.0, excite=2.0)
b = {key: (v.copy() if isinstance(v, np.ndarray) else v) for key, v in trot.items()}
idx = np.arange(24) * (n // 24) + 3
for key in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
    b[key][idx] = stance[key][idx]
S = pkg.to_soa
d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
         fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])),
         force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32))
def tick():
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], d["force"], d["tau"], d["status"])
for _ in range(5):                                # a plan exists and the host has seen its length
    tick(); ctx.sync()
ref_tau, ref_st = d["tau"].download().T.copy(), d["status"].download()
big = ctx.alloc((1 << 28,))
pin = ctx.alloc_pinned((28, n)); pin.array[...] = S(b["mpc_state"])
d["state"].zero()                                 # what a planned launch that ran too early would read
d["tau"].upload(np.full((12, n), np.nan, np.float32)); d["status"].upload(np.full((n,), 0x7f0000ff, np.int32))
ctx.sync()
for _ in range(200):
    big.zero()
d["state"].copy_from_pinned(pin)                  # the caller's producer, behind ~30 ms of its other work on the context's stream
tick(); ctx.sync()
st, tau = d["status"].download(), d["tau"].download().T
import ctypes as C
lists = np.zeros(8, np.int32); ctx._lib.qrgpu_debug_lists.argtypes = [C.c_void_p, C.c_void_p]; ctx._lib.qrgpu_debug_lists(ctx._h, lists.ctypes.data)
print("result", int((G.flags(st) != G.flags(ref_st)).sum()), int(np.isfinite(tau).all()), float(np.abs(tau - ref_tau).max()), int(lists[5] == lists[7] and lists[7] > 0), int(lists[6] != 0))
"""


def test_planned_launch_whose_go_never_comes_is_called_off(pkg):
    """The planned launch (listed robots on whole CUs beside the main pass) is released by a "go" that the context's stream gives when it reaches
    the call -- polled by a one-thread launch on the side stream, bounded.  Should that gate give up (the caller had more work queued in front of
    the call than the bound), the planned workgroups must not run on inputs that are not there yet: they leave, and the main pass solves the
    robots it would have skipped.  A process of its own with the bound at 1 ms (QRGPU_PLAN_GO_MS; the WBC gate's too), ~30 ms of fills queued in
    front of the tick and the MPC's state array produced by the last copy in that queue."""
    import subprocess, sys, os
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, QRGPU_PLAN_GO_MS="1", QRGPU_PIPE_GATE_MS="1")
    r = subprocess.run([sys.executable, "-c", _PLAN_GO_SCRIPT, here], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [l.split()[1:] for l in r.stdout.strip().splitlines() if l.startswith("result")][0]
    assert int(res[0]) == 0 and int(res[1]) == 1, res
    assert float(res[2]) < 2e-4, res
    assert int(res[3]) == 1, res          # (the plan was called off in that tick; the WBC launch's gate gives up too unless the streams share a hardware queue)


_BACKLOG3_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
h, n = 10, 512
ctx = pkg.Context(0, 1024, 16)
G.setup_a1(ctx, pkg, h)
trot = pkg.make_batch(n, h, "a1", seed=0x51A7, frac_all_stance=0.0, frac_three_leg=0.0)
stance = pkg.make_batch(n, h, "a1", seed=0x51A7, frac_all_stance=1.0, frac_three_leg=0.0, excite=2.0)
b = {key: (v.copy() if isinstance(v, np.ndarray) else v) for key, v in trot.items()}
idx = np.arange(24) * (n // 24) + 3
for key in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
    b[key][idx] = stance[key][idx]
S = pkg.to_soa
d = dict(state=ctx.alloc((28, n)).upload(S(b["mpc_state"])), traj=ctx.alloc((12 * h, n)).upload(S(b["traj"])), gait=ctx.alloc((4 * h, n)).upload(S(b["gait"])),
         fb=ctx.alloc((37, n)).upload(S(b["fb_state"])), cmd=ctx.alloc((67, n)).upload(S(b["wbc_cmd"])), prev=ctx.alloc((3, n)).upload(S(b["prev_ori_vel"])))
outs = [dict(force=ctx.alloc((12, n)), tau=ctx.alloc((12, n)), status=ctx.alloc((n,), np.int32)) for _ in range(4)]
def tick(o):
    ctx.tick_batch(n, d["state"], d["traj"], d["gait"], d["fb"], d["cmd"], d["prev"], o["force"], o["tau"], o["status"])
for _ in range(6):                                # a plan exists and the host has seen its length
    tick(outs[0]); ctx.sync()
ref_tau, ref_st = outs[0]["tau"].download().T.copy(), outs[0]["status"].download()
big = ctx.alloc((1 << 28,))
pin = ctx.alloc_pinned((28, n)); pin.array[...] = S(b["mpc_state"])
pin_fb = ctx.alloc_pinned((37, n)); pin_fb.array[...] = S(b["fb_state"])
d["state"].zero(); d["fb"].zero()                 # what a launch that ran too early would read
for o in outs[1:]:
    o["tau"].upload(np.full((12, n), np.nan, np.float32)); o["status"].upload(np.full((n,), 0x7f0000ff, np.int32))
ctx.sync()
for _ in range(400):
    big.zero()                                    # ~100 ms of the caller's own work on the context's stream: far beyond twice the 1 ms bounds
d["state"].copy_from_pinned(pin); d["fb"].copy_from_pinned(pin_fb)
for o in outs[1:]:                                # THREE ticks queued behind it without a sync: every one of their gates gives up before the first tick runs
    tick(o)
ctx.sync()
bad = 0
worst = 0.0
for o in outs[1:]:
    st, tau = o["status"].download(), o["tau"].download().T
    bad += int((G.flags(st) != G.flags(ref_st)).sum()) + int(not np.isfinite(tau).all())
    worst = max(worst, float(np.nanmax(np.abs(tau - ref_tau))))
print("result", bad, worst)
"""


def test_several_ticks_queued_behind_a_backlog_longer_than_their_gates_bounds(pkg):
    """the give-up words of the WBC gate and of the planned launch's gate were single words shared by every tick in flight.  Two or more
    ticks queued behind a backlog longer than twice the bound then overwrote each other's word before their own kernels had read it: a tick's
    second WBC pass skipped the whole batch, its main pass skipped listed robots that nobody solved -- stale or raw torques, no flag.  The words
    are rings indexed by the epoch now.  A process of its own with both bounds at 1 ms, ~100 ms of fills in front of THREE ticks queued without a
    sync, the inputs produced behind the fills: every robot of every tick against the tick run alone."""
    import subprocess, sys, os
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, QRGPU_PLAN_GO_MS="1", QRGPU_PIPE_GATE_MS="1")
    r = subprocess.run([sys.executable, "-c", _BACKLOG3_SCRIPT, here], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [l.split()[1:] for l in r.stdout.strip().splitlines() if l.startswith("result")][0]
    assert int(res[0]) == 0, res
    assert float(res[1]) < 2e-4, res


_FLAG_WAIT_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from conftest import load_pkg
import gpu_helpers as G
pkg = load_pkg()
h, n = 10, 1024
b = pkg.make_batch(n, h, "a1", seed=0x7A17)
outs = {}
for piped in (False, True):
    ctx = pkg.Context(0, 1024, 16)
    G.setup_a1(ctx, pkg, h)
    ctx.set_tick_pipeline(piped)
    with G.cold_start(ctx):
        for _ in range(3):
            outs[piped] = G.run_tick(ctx, pkg, b)
    ctx.close()
ser, pip = outs[False], outs[True]
to = (G.flags(pip["status"]) & 0x02000000) != 0
same = np.all(ser["tau"] == pip["tau"], 1)
print("result", int(to.sum()), int((~to & ~same).sum()), int((G.flags(ser["status"]) != 0).sum()))
"""


def test_a_wbc_workgroup_that_gives_up_is_never_silent(pkg):
    """a WBC workgroup of a pipelined tick that gives up waiting for its robot's forces stores torques computed from stale forces and a
    status word with QRGPU_ST_PIPE_TIMEOUT -- and the robot's solve, still running, then stored ITS status word over it: raw MPC torque, no
    WBC, no flag.  Now the workgroup leaves "gave up in this epoch" in the robot's flag word and the solve, raising the flag with an exchange,
    finds it and adds the bit behind its own status word.  A process of its own with the bound at 1 us (QRGPU_PIPE_WAIT_US): every robot whose
    workgroup looked before its solve had ended gives up; every robot either carries the flag or has the serial tick's torques bit for bit."""
    import subprocess, sys, os
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, QRGPU_PIPE_WAIT_US="1")
    r = subprocess.run([sys.executable, "-c", _FLAG_WAIT_SCRIPT, here], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [int(x) for x in [l.split()[1:] for l in r.stdout.strip().splitlines() if l.startswith("result")][0]]
    assert res[0] > 0, res                 # some workgroups did give up ...
    assert res[1] == 0 and res[2] == 0, res   # ... and nobody is wrong without saying so


def test_configs4_per_gpu_shard_1024_mixed_h16(gpu_ctx, pkg, oracle):
    """BASELINE.json configs[4] as one GPU sees it: 512 A1 + 512 Lite3 robots interleaved (type_id per robot), horizon 16, the full tick with
    K12 and the K14 tail on, fp32 Hessian assembly -- the workload `bench.py --mixed --horizon 16` times.  Every robot: no flag, forces inside
    the friction pyramid, zero on swing feet, torques within the clip, bit-identical on a second cold run; a sample of 96 robots (the 32 with
    the most working-set changes among them) against the oracle, A1 and Lite3 each with its own parameters."""
    h, n = 16, 1024
    gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); gpu_ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
    gpu_ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h); gpu_ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
    try:
        ba = pkg.make_batch(n // 2, h, "a1", seed=0xA1 + 2); bl = pkg.make_batch(n // 2, h, "lite3", seed=0xA1 + 2 + 0xD2)
        b = dict(ba)
        for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
            b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
        b["n"] = n
        tid = pkg.shard.interleave_types(n, 2)
        gpu_ctx.set_torque_epilogue(hip_comp=True, clip=True)
        with G.cold_start(gpu_ctx):
            out = G.run_tick(gpu_ctx, pkg, b, type_id=tid, want_qdes=True)
            out2 = G.run_tick(gpu_ctx, pkg, b, type_id=tid, want_qdes=True)
        assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
        for k in ("force", "tau", "status", "qdes"):
            assert np.array_equal(out[k], out2[k]), k
        f = out["force"].reshape(n, 4, 3)
        mu = np.float32(0.45)
        fmaxv = np.where(tid == 0, np.float32(13.0 * 9.81), np.float32(8.742 * 9.81))[:, None]
        assert np.all(f[:, :, 2] >= -1e-5) and np.all(f[:, :, 2] <= fmaxv * (1 + 1e-6))
        assert np.all(np.abs(f[:, :, 0]) <= mu * f[:, :, 2] + 2e-5) and np.all(np.abs(f[:, :, 1]) <= mu * f[:, :, 2] + 2e-5)
        assert np.all(f[b["gait"][:, :4] == 0] == 0)
        assert np.abs(out["tau"]).max() <= 23.0 and np.all(np.isfinite(out["qdes"]))
        it = G.iterations(out["status"])
        sample = sorted(set(np.argsort(-it)[:32].tolist() + list(range(0, n, 16))))
        for i in sample:
            robot = "a1" if tid[i] == 0 else "lite3"
            md = pkg.model_desc(robot)
            ft, tt, st, _, _ = oracle.tick_batch(1, pkg.mpc_cfg(robot), h, md[:3], md, b["mpc_state"][i:i + 1], b["traj"][i:i + 1], b["gait"][i:i + 1], b["fb_state"][i:i + 1],
                                                 b["wbc_cmd"][i:i + 1], b["prev_ori_vel"][i:i + 1].copy(), epilogue=3)
            assert st[0] == 0, i
            assert np.abs(out["force"][i] - ft[0]).max() <= 1e-5 * max(1.0, np.abs(ft[0]).max()), (i, it[i])
            assert np.all(np.abs(out["tau"][i] - tt[0]) <= G.tau_tol(tt[0], 1e-4)), (i, it[i])
        assert it.max() >= 60                       # (the shard holds the hard robots DESIGN.md 8 talks about)
    finally:
        gpu_ctx.set_torque_epilogue(False, False)
        G.setup_a1(gpu_ctx, pkg, 10)
