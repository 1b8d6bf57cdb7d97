"""-m gpu: the HIP MPC kernel (through the C ABI) against the CPU oracle on identical inputs."""
import numpy as np
import pytest

import gpu_helpers as G

pytestmark = pytest.mark.gpu


def _oracle_forces(oracle, pkg, b, robot="a1"):
    cfg = pkg.mpc_cfg(robot)
    n, h = b["n"], b["horizon"]
    f = np.zeros((n, 12)); tau = np.zeros((n, 12), np.float32)
    geom = pkg.model_desc(robot)[:3]
    for i in range(n):
        u, st, rc = oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert rc == 0
        f[i] = u[:12]
        tau[i] = oracle.mpc_force_to_torque(geom, b["fb_state"][i, 0:4], b["fb_state"][i, 13:25], u[:12])
    return f, tau


def test_assembly_bit_exact(gpu_ctx, pkg, oracle):
    """fp32 H and g from the kernel are bit-identical to the oracle's dense k-ordered GEMM
    (all-stance robots so that every entry is produced) and on stance x stance entries for trot."""
    G.setup_a1(gpu_ctx, pkg, 10)
    cfg = pkg.mpc_cfg("a1")
    for fr_all, seed in ((1.0, 3), (0.05, 4)):
        b = pkg.make_batch(16, 10, "a1", seed=seed, frac_all_stance=fr_all, frac_three_leg=0.0 if fr_all == 1.0 else 0.1)
        Hg, gg = G.run_assemble(gpu_ctx, pkg, b)
        for i in range(b["n"]):
            Ho, go, ub = oracle.mpc_assemble(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            free = np.repeat(ub[4::5] > 0, 3)
            m2 = np.outer(free, free)
            assert np.array_equal(Hg[i][m2].view(np.uint32), Ho[m2].view(np.uint32)), "H differs bitwise (robot %d)" % i
            assert np.array_equal(gg[i][free].view(np.uint32), go[free].view(np.uint32)), "g differs bitwise (robot %d)" % i
            assert np.all(np.isnan(Hg[i][~m2]))      # swing entries are never touched


@pytest.mark.parametrize("horizon,n,seed", [(10, 256, 0xA2), (5, 64, 7)])
def test_mpc_parity(gpu_ctx, pkg, oracle, horizon, n, seed):
    """Config 2 of BASELINE.json (256 A1, h=10, MPC only): forces and J^T f torques vs the oracle."""
    G.setup_a1(gpu_ctx, pkg, horizon)
    b = pkg.make_batch(n, horizon, "a1", seed=seed)
    out = G.run_mpc(gpu_ctx, pkg, b)
    assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
    f, tau = _oracle_forces(oracle, pkg, b)
    fmax = np.abs(f).max(axis=1, keepdims=True)
    assert np.abs(out["force"] - f).max() <= 1e-5 * max(1.0, np.abs(f).max()), np.abs(out["force"] - f).max()
    assert np.all(np.abs(out["force"] - f) <= 1e-6 * np.maximum(1.0, fmax) + 1e-5)
    assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau)), np.abs(out["tau"] - tau).max()


def test_mpc_edge_cases(gpu_ctx, pkg, oracle):
    """all swing (flight), all stance, three legs, one stance leg-step only."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(8, 10, "a1", seed=11)
    g = b["gait"].reshape(8, 10, 4)
    g[0] = 0.0                       # flight: no free variable, forces 0
    g[1] = 1.0                       # stand
    g[2] = 1.0; g[2, :, 2] = 0.0     # three legs
    g[3] = 0.0; g[3, 0, 1] = 1.0     # a single stance leg-step
    g[4] = 0.0; g[4, 5:, :] = 1.0    # flight now, stance later: first-step forces 0 but the QP is not empty
    b["gait"] = g.reshape(8, 40)
    out = G.run_mpc(gpu_ctx, pkg, b)
    assert np.all(G.flags(out["status"]) == 0)
    f, tau = _oracle_forces(oracle, pkg, b)
    assert np.all(out["force"][0] == 0) and np.all(out["tau"][0] == 0)
    assert np.all(out["force"][4] == 0)
    assert np.abs(out["force"] - f).max() <= 1e-5 * max(1.0, np.abs(f).max())
    assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau))


def test_mpc_kkt_full_size(gpu_ctx, pkg, oracle):
    """Size-independent property at the bench size (1024 robots): the first-step forces satisfy the
    pyramid constraints, and perturbing the state slightly perturbs the forces slightly (no
    active-set garbage).  Feasibility is checked for every robot, optimality through the oracle on
    a sample."""
    G.setup_a1(gpu_ctx, pkg, 10)
    b = pkg.make_batch(1024, 10, "a1", seed=0xA3)
    out = G.run_mpc(gpu_ctx, pkg, b)
    assert np.all(G.flags(out["status"]) == 0)
    f = out["force"].reshape(1024, 4, 3)
    mu = np.float32(0.45); fmaxv = np.float32(13 * 9.81)
    assert np.all(f[:, :, 2] >= -1e-6) and np.all(f[:, :, 2] <= fmaxv * (1 + 1e-6))
    assert np.all(np.abs(f[:, :, 0]) <= mu * f[:, :, 2] + 1e-5) and np.all(np.abs(f[:, :, 1]) <= mu * f[:, :, 2] + 1e-5)
    contact = b["gait"][:, :4]
    assert np.all(f[contact == 0] == 0)
    cfg = pkg.mpc_cfg("a1")
    for i in range(0, 1024, 64):
        u, st, rc = oracle.mpc_solve(cfg, 10, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert np.abs(out["force"][i] - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max())


def test_mpc_lite3_and_mixed_types(gpu_ctx, pkg, oracle):
    """type_id selects the parameter set per robot (config 5 mixes A1 and Lite3)."""
    G.setup_a1(gpu_ctx, pkg, 10)
    gpu_ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), 10)
    gpu_ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
    ba = pkg.make_batch(16, 10, "a1", seed=21); bl = pkg.make_batch(16, 10, "lite3", seed=22)
    b = dict(ba)
    for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
        b[k] = np.concatenate([ba[k], bl[k]], axis=0)
    b["n"] = 32
    tid = np.array([0] * 16 + [1] * 16, np.int32)
    out = G.run_mpc(gpu_ctx, pkg, b, type_id=tid)
    assert np.all(G.flags(out["status"]) == 0)
    fa, ta = _oracle_forces(oracle, pkg, ba, "a1"); fl, tl = _oracle_forces(oracle, pkg, bl, "lite3")
    f = np.concatenate([fa, fl]); tau = np.concatenate([ta, tl])
    assert np.abs(out["force"] - f).max() <= 1e-5 * max(1.0, np.abs(f).max())
    assert np.all(np.abs(out["tau"] - tau) <= G.tau_tol(tau))


def test_mpc_single_robot_interface(gpu_ctx, pkg, oracle):
    """SetupProblem / SolveMPCKernel / GetMPCSolution, as SolveDenseMPC calls them."""
    mpc = pkg.MPCInterface(gpu_ctx, 0)
    c = pkg.mpc_cfg("a1")
    assert mpc.GetMPCSolution(0) == 0.0            # has_solved == 0
    mpc.SetupProblem(c[0], 10, c[1], c[2], c[3], c[4:7], c[7:19], c[19])
    b = pkg.make_batch(4, 10, "a1", seed=31)
    for i in range(4):
        s = b["mpc_state"][i]
        mpc.SolveMPCKernel(s[0:3], s[3:6], s[6:10], s[10:13], s[13:25].reshape(4, 3).T, s[25:28], b["traj"][i], b["gait"][i])
        u, st, rc = oracle.mpc_solve(c, 10, s, b["traj"][i], b["gait"][i])
        got = np.array([mpc.GetMPCSolution(k) for k in range(12)])
        assert G.flags(mpc.status) == 0
        assert np.abs(got - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max())


def test_wave_helpers(gpu_ctx):
    """DPP min / sum all-reduce, first_lane and readlane helpers used by the active-set loop."""
    import ctypes as C
    out = np.zeros(256)
    lib = gpu_ctx._lib
    lib.qrgpu_selftest.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.qrgpu_selftest(gpu_ctx._h, out.ctypes.data) == 0
    v = ((np.arange(64) * 37 + 11) % 64) - 20.5
    assert np.all(out[:64] == v.min())
    assert np.all(out[64:128] == 64 * 65 / 2)
    assert np.all(out[128:192] == np.argmin(v))
    assert np.all(out[192:256] == v[17])


def test_beyond_64_rows_goes_through_the_list_pass(gpu_ctx, pkg, oracle):
    """h = 10: the main pass holds 64 working-set positions (one per lane).  A solve that needs more is put on the rescue list and
    re-solved by the list pass (whole CU's LDS, positions 64..95 in a second register set).  Three-leg-stance robots at 3x SURVEY.md 8d's
    ranges reach 74 active rows; without the list pass they carry QRGPU_ST_MPC_OVERFLOW, never a silent answer."""
    h = 10
    G.setup_a1(gpu_ctx, pkg, h)
    b = pkg.make_batch(96, h, "a1", seed=0x3e9, excite=3.0, frac_all_stance=0.0, frac_three_leg=1.0)
    gpu_ctx.set_rescue_pass(False)
    try:
        flagged = (G.run_mpc(gpu_ctx, pkg, b)["status"] & 0x4) != 0
    finally:
        gpu_ctx.set_rescue_pass(True)
    out = G.run_mpc(gpu_ctx, pkg, b)
    assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
    cfg = pkg.mpc_cfg("a1")
    big = 0
    for i in range(96):
        u, st, rc = oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert rc == 0
        big += st["n_active"] > 64
        assert np.abs(out["force"][i] - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max()), i
    assert big >= 10 and flagged.sum() >= big, "the batch must exercise positions beyond 64"


def test_list_pass_serves_more_robots_than_it_has_workgroups(gpu_ctx, pkg, oracle):
    """The list pass has a fixed grid of 64 workgroups; workgroup b takes entries b, b + 64, ... so that a batch with more than 64
    overflowing robots is still solved completely (and the same way every time: the list order varies, the results do not)."""
    h = 10
    G.setup_a1(gpu_ctx, pkg, h)
    b = pkg.make_batch(512, h, "a1", seed=0xBEE6, excite=2.0, frac_all_stance=1.0, frac_three_leg=0.0)
    gpu_ctx.set_rescue_pass(False)
    try:
        flagged = (G.run_mpc(gpu_ctx, pkg, b)["status"] & 0x4) != 0
    finally:
        gpu_ctx.set_rescue_pass(True)
    assert flagged.sum() > 64, flagged.sum()
    with G.cold_start(gpu_ctx):
        out = G.run_mpc(gpu_ctx, pkg, b)
        out2 = G.run_mpc(gpu_ctx, pkg, b)
    assert np.array_equal(out["force"], out2["force"]) and np.array_equal(out["status"], out2["status"])
    cfg = pkg.mpc_cfg("a1")
    # what even 96 positions cannot hold keeps the flag (at twice the ranges a few all-stance robots end with ~100 active rows)
    still = np.where((out["status"] & 0x4) != 0)[0]
    assert len(still) <= 8 and len(still) < flagged.sum() - 64
    for i in still:
        u, st, rc = oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert st["n_active"] >= 85, (i, st["n_active"])
    ok = G.flags(out["status"]) == 0
    assert ok.mean() > 0.95                 # (2x the ranges: a few robots may carry the exit-check flag)
    for i in np.where(flagged & ok)[0][::4]:
        u, st, rc = oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert rc == 0
        assert np.abs(out["force"][i] - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max()), i


def test_h16_beyond_64_rows_stays_in_the_multi_wave_loop(gpu_ctx, pkg, oracle):
    """h = 16: working sets beyond the 64 lanes keep a second position per lane (64..95) in the control / worker loop of the MAXB = 9
    variants.  The batch is the A1 half of `bench.py --mixed --horizon 16` (one robot ends
    with 84 active rows, an all-stance one with 66); every robot the oracle finds beyond 60 rows, and a sample of the others, is
    compared with the oracle."""
    h, n = 16, 512
    gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); gpu_ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
    try:
        b = pkg.make_batch(n, h, "a1", seed=0xA1 + 2, excite=1.0)
        out = G.run_mpc(gpu_ctx, pkg, b)
        assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
        it = G.iterations(out["status"])
        cfg = pkg.mpc_cfg("a1")
        cand = list(np.argsort(-it)[:24]) + list(range(0, n, 37))
        big = 0
        for i in cand:
            u, st, rc = oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            assert rc == 0
            big += st["n_active"] > 64
            assert np.abs(out["force"][i] - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max()), (i, st["n_active"])
        assert big >= 2, "the batch must exercise positions beyond 64"
        with G.cold_start(gpu_ctx):
            out1 = G.run_mpc(gpu_ctx, pkg, b)
            out2 = G.run_mpc(gpu_ctx, pkg, b)
        assert np.array_equal(out1["force"], out2["force"]) and np.array_equal(out1["status"], out2["status"])      # deterministic
        assert np.abs(out1["force"] - out["force"]).max() <= 1e-5 * np.abs(out["force"]).max()
    finally:
        G.setup_a1(gpu_ctx, pkg, 10)


def test_rescue_pass_lds_limited_robots(gpu_ctx, pkg, oracle):
    """All-stance robots at h = 10 leave LDS for 56 rows of S^-1 (no hand-over possible): beyond that the robot is flagged
    QRGPU_ST_MPC_OVERFLOW without the rescue pass and re-solved in a second launch with the whole CU's LDS with it."""
    h = 10
    G.setup_a1(gpu_ctx, pkg, h)
    b = pkg.make_batch(64, h, "a1", seed=0xBEE5, excite=1.5, frac_all_stance=1.0, frac_three_leg=0.0)
    gpu_ctx.set_rescue_pass(False)
    try:
        flagged = (G.run_mpc(gpu_ctx, pkg, b)["status"] & 0x4) != 0
    finally:
        gpu_ctx.set_rescue_pass(True)
    with G.cold_start(gpu_ctx):
        out = G.run_mpc(gpu_ctx, pkg, b)
        out2 = G.run_mpc(gpu_ctx, pkg, b)                 # second call: the ping-pong counters
    assert 0 < flagged.sum() <= 64, "the batch must exercise the overflow path"
    assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
    assert np.array_equal(out["force"], out2["force"]) and np.array_equal(out["status"], out2["status"])
    cfg = pkg.mpc_cfg("a1")
    for i in np.where(flagged)[0]:
        u, st, rc = oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert rc == 0 and st["n_active"] > 50
        assert np.abs(out["force"][i] - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max()), i


def test_dispatch_order_does_not_change_results(gpu_ctx, pkg):
    """Longest-first dispatch is scheduling only: slot order, first history-less launch and history-ordered launches agree bit for bit."""
    h, n = 10, 512
    G.setup_a1(gpu_ctx, pkg, h)
    b = pkg.make_batch(n, h, "a1", seed=0x51)
    gpu_ctx.set_warm_start(False)          # (the history that must not matter here is the dispatch order's)
    gpu_ctx.set_planned_list(False)
    gpu_ctx.set_lpt_schedule(False)
    try:
        ref = G.run_mpc(gpu_ctx, pkg, b)
    finally:
        gpu_ctx.set_lpt_schedule(True)
    first = G.run_mpc(gpu_ctx, pkg, b)          # no history yet for this n
    second = G.run_mpc(gpu_ctx, pkg, b)         # ordered by the costs of `first`
    b2 = pkg.make_batch(n, h, "a1", seed=0x52)   # different robots, stale history
    third = G.run_mpc(gpu_ctx, pkg, b2)
    gpu_ctx.set_lpt_schedule(False)
    try:
        ref2 = G.run_mpc(gpu_ctx, pkg, b2)
    finally:
        gpu_ctx.set_lpt_schedule(True)
    gpu_ctx.set_warm_start(True)
    gpu_ctx.set_planned_list(True)
    for o in (first, second):
        assert np.array_equal(o["force"], ref["force"]) and np.array_equal(o["tau"], ref["tau"]) and np.array_equal(o["status"], ref["status"])
    assert np.array_equal(third["force"], ref2["force"]) and np.array_equal(third["status"], ref2["status"])


def test_kernel_timing_sampling_and_pause(gpu_ctx, pkg):
    """qrgpu_enable_timing(ctx, N): HIP events around every N-th launch of a kernel; -1 pauses and a later N carries on; 0 forgets."""
    h, n = 10, 64
    G.setup_a1(gpu_ctx, pkg, h)
    b = pkg.make_batch(n, h, "a1", seed=0x33)
    try:
        gpu_ctx.enable_timing(2)
        for _ in range(4):
            G.run_tick(gpu_ctx, pkg, b)
        ms, cnt = gpu_ctx.get_timing(0)
        assert cnt == 2 and 0.0 < ms < 50.0
        assert gpu_ctx.get_timing(1)[1] == 2
        gpu_ctx.enable_timing(-1)
        for _ in range(2):
            G.run_tick(gpu_ctx, pkg, b)
        assert gpu_ctx.get_timing(0)[1] == 2                  # paused: nothing new, nothing lost
        gpu_ctx.enable_timing(2)
        for _ in range(2):
            G.run_tick(gpu_ctx, pkg, b)
        assert gpu_ctx.get_timing(0)[1] == 3                  # carried on: launches 5 and 6 of the count, one of them bracketed
        gpu_ctx.enable_timing(True)
        gpu_ctx.enable_timing(False)
        gpu_ctx.enable_timing(True)                           # off in between: a fresh start
        G.run_tick(gpu_ctx, pkg, b)
        assert gpu_ctx.get_timing(0)[1] == 1
    finally:
        gpu_ctx.enable_timing(False)


def test_instrumented_kernels_give_the_same_bits(pkg):
    """libqrgpu.so holds the MPC (and WBC) kernels twice, compiled from one source: lean for the timed path, instrumented (executed-arithmetic
    counters, inspection stores, cycle stamps) for the calls that ask for those.  Same robots, same history: the two must agree bit for bit,
    here through the full tick, h = 10 (main pass + list launches) and h = 16, and the counters must have counted."""
    for h, kind in ((10, "a1"), (16, "a1")):
        n = 256
        outs = []
        for counted in (False, True):
            ctx = pkg.Context(0, 1024, 16)            # a context of its own: the same (empty) history on both sides
            G.setup_a1(ctx, pkg, h)
            ctx.enable_flop_count(counted)
            seq = pkg.make_batch_sequence(n, h, kind, seed=0x77, steps=3)
            for b in seq:
                o = G.run_tick(ctx, pkg, b, want_qdes=True)
            outs.append(o)
            if counted:
                fl = ctx.mpc_flop_counts()
                assert fl["fp64_sweep"] > 0 and fl["fp32_matrix"] > 0 and fl["fp64_active_set"] > 0
            del ctx
        a, b_ = outs
        for k in ("tau", "status", "qdes"):
            assert np.array_equal(a[k], b_[k]), (h, k)


def test_warm_start_over_a_coherent_sequence(gpu_ctx, pkg, oracle):
    """Warm start (default on): a robot slot's solve starts from the working set its previous solve ended with, realigned to the scrolling
    contact table.  Over a temporally coherent sequence the answers must be those of a cold start (one optimum; 1e-7 of the force scale
    here, and the oracle's to the usual tolerance), the iteration counts must drop, and a stale guess -- another population in the same
    slots -- must cost speed only."""
    h, n = 10, 256
    G.setup_a1(gpu_ctx, pkg, h)
    seq = pkg.make_batch_sequence(n, h, "a1", seed=0x5EC, steps=5)
    with G.cold_start(gpu_ctx):
        cold = [G.run_mpc(gpu_ctx, pkg, b) for b in seq]
    gpu_ctx.set_warm_start(True)            # forgets everything: the first call is a cold one
    warm = [G.run_mpc(gpu_ctx, pkg, b) for b in seq]
    cfg = pkg.mpc_cfg("a1")
    for k, (c, w, b) in enumerate(zip(cold, warm, seq)):
        assert np.all(G.flags(w["status"]) == 0) and np.all(G.flags(c["status"]) == 0)
        scale = np.maximum(1.0, np.abs(c["force"]).max(1))
        assert (np.abs(w["force"] - c["force"]).max(1) / scale).max() <= 1e-7, k
        for i in range(0, n, 16):
            u, st, rc = oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            assert np.abs(w["force"][i] - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max())
    it_cold = np.mean([G.iterations(c["status"]).mean() for c in cold[1:]])
    it_warm = np.mean([G.iterations(w["status"]).mean() for w in warm[1:]])
    assert np.array_equal(warm[0]["status"], cold[0]["status"])          # nothing stored yet: the same solve
    assert it_warm < 0.6 * it_cold, (it_warm, it_cold)
    # a stale guess: different robots in the same slots
    other = pkg.make_batch(n, h, "a1", seed=0x5ED)
    stale = G.run_mpc(gpu_ctx, pkg, other)
    with G.cold_start(gpu_ctx):
        fresh = G.run_mpc(gpu_ctx, pkg, other)
    assert np.all(G.flags(stale["status"]) == 0)
    assert (np.abs(stale["force"] - fresh["force"]).max(1) / np.maximum(1.0, np.abs(fresh["force"]).max(1))).max() <= 1e-7


def test_planned_list_takes_over_from_the_rescue_pass(gpu_ctx, pkg, oracle):
    """A robot that needed the rescue pass in one call is on the planned list of the next call with the same n: solved beside the main
    launch (which skips it), not after it.  Results are those of the oracle either way; with every robot of the batch all-stance and
    excited, the second call must not leave anything for the trailing rescue pass that the first one had to re-solve."""
    h, n = 10, 256
    G.setup_a1(gpu_ctx, pkg, h)
    b = pkg.make_batch(n, h, "a1", seed=0xBEE7, excite=1.5, frac_all_stance=1.0, frac_three_leg=0.0)
    gpu_ctx.set_warm_start(False); gpu_ctx.set_planned_list(True)
    try:
        gpu_ctx.set_rescue_pass(False)
        flagged = (G.run_mpc(gpu_ctx, pkg, b)["status"] & 0x4) != 0          # who overflows the main pass
        gpu_ctx.set_rescue_pass(True)
        first = G.run_mpc(gpu_ctx, pkg, b)        # no plan yet: main pass + rescue
        second = G.run_mpc(gpu_ctx, pkg, b)       # planned list beside the main pass
        third = G.run_mpc(gpu_ctx, pkg, b)
    finally:
        gpu_ctx.set_warm_start(True)
    assert flagged.sum() >= 8
    cfg = pkg.mpc_cfg("a1")
    for o in (first, second, third):
        assert np.all(G.flags(o["status"]) == 0)
        assert (np.abs(o["force"] - first["force"]).max(1) / np.maximum(1.0, np.abs(first["force"]).max(1))).max() <= 1e-7
    assert np.array_equal(second["force"], third["force"])
    for i in np.where(flagged)[0][:12]:
        u, st, rc = oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        assert np.abs(second["force"][i] - u[:12]).max() <= 1e-5 * max(1.0, np.abs(u[:12]).max()), i


def test_plan_survives_single_robot_and_small_calls_in_between(gpu_ctx, pkg, oracle):
    """the planned list's counters ping-pong on a parity that every trailing list launch flips.  A single-robot call
    (qrgpu_mpc_solve1) or a batch below 64 robots between two planned calls of the same n used to flip it without planning, and the next
    planned call read the counters of the plan before last: listed robots skipped by the main pass and solved by nobody (stale outputs,
    no flag).  Small calls now run as one whole-CU launch that leaves the parity alone, and a trailing launch that does not plan
    forgets the plan.  Every robot of every batched call is compared with the oracle."""
    h, n = 10, 256
    G.setup_a1(gpu_ctx, pkg, h)
    cfg = pkg.mpc_cfg("a1")
    # all stance and excited: a dozen robots outgrow the main pass and live on the planned list
    b = pkg.make_batch(n, h, "a1", seed=0xBEE7, excite=1.5, frac_all_stance=1.0, frac_three_leg=0.0)
    small = pkg.make_batch(24, h, "a1", seed=0xBEE8, excite=1.5, frac_all_stance=1.0, frac_three_leg=0.0)
    f_o = np.stack([oracle.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])[0][:12] for i in range(n)])
    f_s = np.stack([oracle.mpc_solve(cfg, h, small["mpc_state"][i], small["traj"][i], small["gait"][i])[0][:12] for i in range(24)])
    mpc = pkg.MPCInterface(gpu_ctx, 0)
    mpc.SetupProblem(cfg[0], h, cfg[1], cfg[2], cfg[3], cfg[4:7], cfg[7:19], cfg[19])

    def solve1(i):
        s = b["mpc_state"][i]
        mpc.SolveMPCKernel(s[0:3], s[3:6], s[6:10], s[10:13], s[13:25].reshape(4, 3).T, s[25:28], b["traj"][i], b["gait"][i])
        got = np.array([mpc.GetMPCSolution(k) for k in range(12)])
        assert G.flags(mpc.status) == 0
        assert np.abs(got - f_o[i]).max() <= 1e-5 * max(1.0, np.abs(f_o[i]).max()), i

    def check(out, ref, what):
        assert np.all(G.flags(out["status"]) == 0), (what, np.unique(G.flags(out["status"])))
        err = np.abs(out["force"] - ref).max(1) / np.maximum(1.0, np.abs(ref).max(1))
        assert err.max() <= 1e-5, (what, int(err.argmax()), err.max())

    gpu_ctx.set_warm_start(False); gpu_ctx.set_planned_list(True)
    try:
        gpu_ctx.set_rescue_pass(False)
        listed = (G.flags(G.run_mpc(gpu_ctx, pkg, b)["status"]) & 0x4) != 0
        gpu_ctx.set_rescue_pass(True)
        assert listed.sum() >= 8
        check(G.run_mpc(gpu_ctx, pkg, b), f_o, "first (no plan yet)")
        check(G.run_mpc(gpu_ctx, pkg, b), f_o, "second (planned)")
        # odd numbers of small calls between planned calls, in every mix
        for k, between in enumerate(((1, 0), (0, 1), (3, 0), (1, 1), (2, 1), (1, 2))):
            for j in range(between[0]):
                solve1(int(np.where(listed)[0][(k + j) % listed.sum()]))
            for j in range(between[1]):
                check(G.run_mpc(gpu_ctx, pkg, small), f_s, "small batch")
            # (run_mpc poisons its output buffers: a robot that no launch solves comes back as NaN with every flag set)
            check(G.run_mpc(gpu_ctx, pkg, b), f_o, "planned call after %r small calls" % (between,))
    finally:
        gpu_ctx.set_warm_start(True)


@pytest.mark.parametrize("h,n,mixed", [(10, 256, False), (16, 1024, True)])
def test_bf16x3_solve_vs_oracle_on_its_own_hessian(gpu_ctx, pkg, oracle, h, n, mixed):
    """BASELINE.json configs[4]'s arithmetic ("fp32 QP + bf16 Hessian MFMA"): the Hessian contraction of qr_mpc_interface.cpp:396-412 on
    v_mfma_f32_16x16x32_bf16 with three bf16 limbs per fp32 operand, on the configs[4] per-GPU shard (512 A1 + 512 Lite3 interleaved, h = 16)
    and on configs[1]'s size (256 A1, h = 10).  The mode is ANOTHER ROUNDING of H -- within an ulp or two of the exact fp32 assembly, not
    bit-identical -- and an ulp of H is worth up to 1e-3 of force on this QP (alpha = 4e-6), so the comparator is not the f32 mode's answer but
    the optimum of the QP that this mode states: the (H, g) the kernel assembled are downloaded and the ORACLE solves exactly that data
    (oracle.mpc_solve_hg: the solve of :428-438 as the oracle states it; bounds and friction rows from the gait table).  Asserted for every robot:
      * |dH| <= 4e-7 max|H| against the exact fp32 assembly, the gradient bit-identical (it stays on the fp32 vector path);
      * forces  <= 1e-5 max(1, |f|max)  against the oracle's solve of the mode's own (H, g);
      * full-tick K14 torque  <= 1e-4 max(1, |tau|)  against the oracle's tick tail fed with those oracle forces (north_star's bar)."""
    gpu_ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); gpu_ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
    gpu_ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h); gpu_ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
    try:
        ba = pkg.make_batch(n // 2, h, "a1", seed=0xA1 + 4); bl = pkg.make_batch(n // 2, h, "lite3" if mixed else "a1", seed=0xA1 + 4 + 0xD2)
        b = dict(ba)
        for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
            b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
        b["n"] = n
        tid = pkg.shard.interleave_types(n, 2) if mixed else np.zeros(n, np.int32)
        gpu_ctx.set_torque_epilogue(hip_comp=True, clip=True)
        with G.cold_start(gpu_ctx):
            gpu_ctx.set_hessian_mode("f32")
            Hx, gx = G.run_assemble(gpu_ctx, pkg, b, type_id=tid)
            gpu_ctx.set_hessian_mode("bf16x3")
            Hs, gs = G.run_assemble(gpu_ctx, pkg, b, type_id=tid)
            out = G.run_tick(gpu_ctx, pkg, b, type_id=tid)
        assert np.all(G.flags(out["status"]) == 0), np.unique(G.flags(out["status"]))
        differs = 0
        worst_f = worst_t = 0.0
        for i in range(n):
            robot = "a1" if tid[i] == 0 else "lite3"
            cfg, md = pkg.mpc_cfg(robot), pkg.model_desc(robot)
            m = np.isfinite(Hx[i])
            assert np.array_equal(m, np.isfinite(Hs[i]))
            assert np.abs(Hs[i][m].astype(np.float64) - Hx[i][m]).max() <= 4e-7 * np.abs(Hx[i][m]).max(), i
            differs += not np.array_equal(Hs[i][m], Hx[i][m])
            assert np.array_equal(gs[i][np.isfinite(gx[i])], gx[i][np.isfinite(gx[i])])
            u, st, rc = oracle.mpc_solve_hg(cfg, h, b["gait"][i], Hs[i], gs[i])
            assert rc == 0, (i, rc)
            f_o = u[:12]
            ef = np.abs(out["force"][i] - f_o).max() / max(1.0, np.abs(f_o).max())
            worst_f = max(worst_f, ef)
            assert ef <= 1e-5, (i, ef, st)
            tau_o, _ = oracle.tick_from_forces(md[:3], md, b["fb_state"][i], b["wbc_cmd"][i], b["prev_ori_vel"][i], f_o, mode=1, epilogue=3)
            et = np.abs(out["tau"][i] - tau_o) / np.maximum(1.0, np.abs(tau_o))
            worst_t = max(worst_t, et.max())
            assert np.all(et <= 1e-4), (i, et.max(), st)
        assert differs > n // 2                                                      # (it really is another arithmetic)
        print("bf16x3 h=%d n=%d: worst rel force %.2e, worst rel full-tick torque %.2e vs the oracle on the mode's own (H, g); %d of %d Hessians differ "
              "from the exact fp32 assembly" % (h, n, worst_f, worst_t, differs, n))
    finally:
        gpu_ctx.set_hessian_mode("f32")
        gpu_ctx.set_torque_epilogue(False, False)
        G.setup_a1(gpu_ctx, pkg, 10)
