#!/usr/bin/env python3
"""Headline benchmark: MPC+WBC control ticks/s over a batch of A1 robots (BASELINE.json configs[2]:
1024 A1 instances, horizon 10, full MPC+WBC tick per robot), one process per GPU.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: this process starts N rank processes itself and never
                                                          touches a GPU; any rank failing makes it exit non-zero -- it never falls back to 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one qrgpu_tick_batch over this rank's 1024 robots -- K1-K14 of SURVEY.md 8(a): MPC, then WBC with the kinematic projection
(K12) on and the motor tail (K14: abad compensation, +-23 N m clip) applied -- with inputs already resident in HBM, and, for N > 1,
one all-gather of the per-robot torques (qrgpu_allgather_tau: RCCL over xGMI on the context's own stream, overlapped with the next
tick; weak scaling: every rank owns 1024 robots; the timed region ends when the last gather has landed).

What is stepped through.  Not one batch over and over: the K timed steps are split over D = 8 robot populations ("draws", different
generator seeds, different on every rank), and inside a draw consecutive steps see consecutive batches of a temporally coherent
sequence (the same robots 0.03 s later: workload.make_batch_sequence, 8 batches walked back and forth), so the scheduler's history
(longest-first dispatch from the previous step's solve times) is a prediction, never a replay.  Every draw gets W untimed warm-up steps
and its share of the K timed steps, bracketed by barrier + synchronize; `value` is the MEDIAN over the draws of the draw's rate (max
over ranks of its time), min / max / per-draw rates beside it.

No GPU array library is involved: device buffers, pinned host buffers, the stream, event timing and the collective are all behind the C
ABI of include/qrgpu.h (qrgpu_malloc, qrgpu_host_alloc, qrgpu_enable_timing, qrgpu_mark, qrgpu_allgather_tau).  torch is not imported
at any N: with N > 1 the launcher's plumbing -- rendezvous, barrier, max-reduce of the draw times, handing rank 0's 128-byte communicator
id to the other ranks -- is a localhost socket (quadruped-robot_amd/rendezvous.py; the ranks find each other through RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT as torch.distributed.run or this program's own launcher set them).  Rank 0 prints ONE JSON line.

  python bench.py --mode single     the drop-in boundary's single-robot latency (qrgpu_mpc_solve1 / qrgpu_wbc_run1, and both through the C++
                                    adapters): p50 / p99 over 1000 calls at h = 5, 10, 16 beside the reference's solver on the same QPs
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

# SURVEY.md 8(d) algorithmic figures at h = 10 (dense counts, mul+add = 2)
FLOP_K4_HESSIAN = 2 * 120 * 130 * 120            # 3.744 MFLOP
FLOP_K4_GRADIENT = 2 * 120 * 130 + 2 * 130 * 13  # 34.6 kFLOP
FLOP_K6_FACTOR = 120 ** 3 / 3.0                  # 0.576 MFLOP
FLOP_K6_PER_ITER = 4 * 120 ** 2                  # 57.6 kFLOP per working-set change
FLOP_WBC = 0.35e6
BYTES_PER_TICK = 1216                            # algorithmic HBM bytes per full tick at h = 10
PEAK_F32_MATRIX_TFLOPS = 157.3                   # MI355X_MICROARCH.md: f32-in MFMA = f32 vector peak
PEAK_F64_VECTOR_TFLOPS = 78.6                    # MI355X FP64 vector peak (spec: half the FP32 vector rate); scratch/ubench/lat.hip measures 59 TFLOP/s of
                                                 # v_fma_f64 at two waves per SIMD
PEAK_HBM_GBS = 8000.0
EST_IN_ROWS, EST_OUT_ROWS = 54, 42                # QRGPU_EST_IN_ROWS / QRGPU_EST_OUT_ROWS (include/qrgpu.h)


def _load_pkg():
    d = os.path.join(ROOT, "quadruped-robot_amd")
    spec = importlib.util.spec_from_file_location("quadruped_robot_amd", os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["quadruped_robot_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_rendezvous():
    """quadruped-robot_amd/rendezvous.py on its own (standard library only): the dry run of the launcher test has no built extension to load."""
    if "quadruped_robot_amd" in sys.modules:
        return sys.modules["quadruped_robot_amd"].rendezvous
    spec = importlib.util.spec_from_file_location("qrgpu_rendezvous", os.path.join(ROOT, "quadruped-robot_amd", "rendezvous.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(pkg, b, horizon, mode=1, epilogue=3, budget_s=6.0):
    """The CPU restatement (oracle/, kind "port") timed on this box's host cores over a bounded sample: the same tick the GPU runs
    (K12 always runs on the CPU side, as in the reference; K14 tail on)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    cfg, md = pkg.mpc_cfg("a1"), pkg.model_desc("a1")
    args = (mode, cfg, horizon, md[:3], md, b["mpc_state"], b["traj"], b["gait"], b["fb_state"], b["wbc_cmd"])
    n = b["n"]
    # single thread on a 128-robot slice, all cores on the whole batch
    m1 = min(n, 128)
    sl = {k: (v[:m1] if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    t0 = time.perf_counter()
    O.tick_batch(mode, cfg, horizon, md[:3], md, sl["mpc_state"], sl["traj"], sl["gait"], sl["fb_state"], sl["wbc_cmd"],
                 sl["prev_ori_vel"].copy(), nthreads=1, epilogue=epilogue)
    t1 = time.perf_counter() - t0
    passes, wall = 0, 0.0
    while wall < budget_s and passes < 8:
        t0 = time.perf_counter()
        f_cpu, tau_cpu, st_cpu, _, _, q_cpu = O.tick_batch(*args, b["prev_ori_vel"].copy(), nthreads=cores, epilogue=epilogue, want_qdes=True)
        wall += time.perf_counter() - t0
        passes += 1
    cpu_baseline.outputs = (f_cpu, tau_cpu, st_cpu, q_cpu)      # the checker's answer on this batch, for max_rel_*_err_vs_cpu
    out = dict(value=passes * n / wall, unit="ticks/s", cores=cores, kind="port",
               sample="%d passes over one %d-robot batch (draw 0, first batch) on %d threads: oracle/ CPU restatement, own fp64 active-set QP, "
                      "K12 and the K14 tail included" % (passes, n, cores),
               single_thread_value=m1 / t1)
    out.update(reference_solver(pkg, O, b, horizon))
    if out.get("reference_solver_ratio"):
        # DERIVED, not measured: what this box's cores would do with the reference's own solver in the port's place (>= 95 % of the reference's tick is
        # its QP solve) -- the port's rate divided by how much slower the reference's solver is on the same QPs
        out["reference_equivalent_ticks_per_s"] = out["value"] / out["reference_solver_ratio"]
        out["reference_equivalent_is"] = "derived: value / reference_solver_ratio (the reference's Eigen glue cannot be built here; its solver can, and is >= 95 % of its tick)"
    return out


def reference_solver(pkg, O, b, horizon, sample=24):
    """How the port's solver relates to the reference's: qpOASES 3.2.0 compiled from the reference tree (oracle/_ref, travels as a built
    .so) fed exactly what qr_mpc_interface.cpp:418-438 feeds it (asymmetric fp32 H, nWSR = 100), against the port's solver on the same QPs,
    one thread, `sample` robots of the bench batch.  Also returns the reference's as-called first-step forces for the parity figure."""
    if O.ref() is None:
        return dict(reference_solver_ratio=None, reference_solver_note="oracle/_ref not present on this box")
    cfg = pkg.mpc_cfg("a1")
    A = O.mpc_constraint_matrix(horizon, float(cfg[1]))
    idx = np.arange(min(sample, b["n"]))
    t_ref = t_port = 0.0
    f_ref = np.zeros((idx.size, 12)); nwsr = np.zeros(idx.size, int); rc = np.zeros(idx.size, int)
    for j, i in enumerate(idx):
        H, g, ub = O.mpc_assemble(cfg, horizon, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        t0 = time.perf_counter()
        x, info = O.ref_qpoases_mpc(H.astype(np.float64), g.astype(np.float64), A, np.zeros(20 * horizon), ub.astype(np.float64), 100)
        t_ref += time.perf_counter() - t0
        f_ref[j], nwsr[j], rc[j] = x[:12], info["nWSR"], info["init_rc"]
        t0 = time.perf_counter()
        O.mpc_solve(cfg, horizon, b["mpc_state"][i], b["traj"][i], b["gait"][i])
        t_port += time.perf_counter() - t0
    reference_solver.as_called = (idx, f_ref, nwsr, rc)
    return dict(reference_solver_ratio=t_ref / t_port,
                reference_solver_ms_per_solve=1e3 * t_ref / idx.size, port_solver_ms_per_solve=1e3 * t_port / idx.size,
                reference_solver_note="qpOASES 3.2.0 from the reference tree as qr_mpc_interface.cpp:428-438 calls it vs the port's assemble+solve, "
                                      "%d QPs of the bench batch, one thread (the port's time includes its fp32 assembly)" % idx.size)


class Dev:
    """Device-buffer helpers over the package's own allocator (qrgpu_malloc / qrgpu_memcpy_*): what torch used to be asked for."""

    def __init__(self, pkg, ctx):
        self.pkg, self.ctx = pkg, ctx

    def soa(self, a):
        """robot-major host array [n][f] -> device array [f][n]"""
        h = self.pkg.to_soa(a)
        return self.ctx.alloc(h.shape, h.dtype).upload(h)

    def zeros(self, shape, dtype=np.float32):
        return self.ctx.alloc(shape, dtype).upload(np.zeros(shape, dtype))

    def put(self, a):
        a = np.ascontiguousarray(a)
        return self.ctx.alloc(a.shape, a.dtype).upload(a)


def side_mode(args, pkg, ctx):
    """The SURVEY 8f rows on one GPU: `vmc` = force-balance stance QP (ComputeContactForce), `frontend` = MPC front-end
    (SetupCommand/Run/UpdateMPC).  A step = one batched call over --robots robots; kernel time by marks on the launch stream."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build()
    n, h = args.robots, args.horizon
    D_ = Dev(pkg, ctx)
    T = D_.soa
    W = pkg.workload
    if args.mode == "vmc":
        cfg, geom = W.vmc_cfg("a1"), pkg.model_desc("a1")[:3]
        ctx.vmc_setup_packed(0, cfg, geom)
        vin, q = W.make_vmc_batch(n, seed=0xB2)
        d_in, d_q = T(vin), T(q)
        d_f, d_t = D_.zeros((12, n)), D_.zeros((12, n))
        d_s = D_.zeros((n,), np.int32)
        step = lambda: ctx.vmc_force_batch(n, d_in, d_q, d_f, d_t, d_s)
        cpu = lambda i: O.vmc_solve(cfg, geom, vin[i], q[i])
        alg_bytes = (37 + 12 + 12 + 12 + 1) * 4
        metric, unit, kernel = "force-balance QP solves/s (batched robots)", "solves/s", "qr_vmc_kernel"
        # assembly 12x12x6 + 6x12, inverse 12^3, ~20 working-set changes of ~4*12^2 (dense count, as SURVEY 8d does for the MPC QP)
        alg_flop = 2 * 12 * 12 * 6 + 2 * 6 * 12 + 12 ** 3 + 20 * 4 * 12 ** 2
    elif args.mode == "estimator":
        cfg = W.estimator_cfg("a1")
        xs, stamps = W.make_estimator_sequence(n, 4, seed=0xE5)
        d_in = T(xs[0]); d_tick = D_.put(stamps[0].astype(np.int32))
        S_ = ctx.estimator_state_doubles(int(cfg[6]))
        d_state = D_.zeros((S_, n), np.float64)
        d_out = D_.zeros((EST_OUT_ROWS, n))          # QRGPU_EST_OUT_ROWS of include/qrgpu.h
        step = lambda: ctx.estimator_update_batch(n, cfg, d_in, d_tick, d_state, d_out)
        seq = np.repeat(xs[0][None, :1], 200, 0)[:, 0]
        cpu_all = lambda: O.estimator_run(cfg, seq, (1000 + 2 * np.arange(200)).astype(np.uint32))
        cpu = None
        assert xs[0].shape[1] == EST_IN_ROWS
        alg_bytes = (EST_IN_ROWS + 1 + EST_OUT_ROWS) * 4 + 2 * (32 + 6) * 8     # inputs + tick + outputs, and the touched part of the filter memory (read + write)
        metric, unit, kernel = "velocity-estimator robot-ticks/s (batched robots)", "robot-ticks/s", "qr_estimator_kernel"
        alg_flop = 0
    else:
        vin, st = W.make_frontend_batch(n, seed=0xFE)
        d_in, d_st = T(vin), T(st)
        d_traj, d_gait = D_.zeros((12 * h, n)), D_.zeros((4 * h, n))
        d_cmd, d_u = D_.zeros((67, n)), D_.zeros((n,), np.int32)
        step = lambda: ctx.mpc_frontend_batch(n, d_in, d_st, d_traj, d_gait, d_cmd, d_u)
        cpu = lambda i: O.mpc_frontend(h, 2, vin[i], st[i])
        alg_bytes = (64 + 8 + 8 + 16 * h + 19 + 1) * 4
        metric, unit, kernel = "MPC front-end robot-ticks/s (batched robots)", "robot-ticks/s", "qr_frontend_kernel"
        alg_flop = 0
    for _ in range(args.warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    ctx.mark(0)
    for _ in range(args.steps):
        step()
    ctx.mark(1)
    ctx.sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = ctx.mark_elapsed_ms(0, 1) / args.steps
    # CPU restatement, one thread, bounded sample
    m = min(n, 2000)
    c0 = time.perf_counter()
    if cpu is None:
        for _ in range(10):
            cpu_all()
        m = 2000
    else:
        for i in range(m):
            cpu(i)
    cpu_rate = m / (time.perf_counter() - c0)
    gbs = alg_bytes * n / (kernel_ms * 1e-3) / 1e9
    out = {"metric": metric, "value": n * args.steps / elapsed, "unit": unit, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 assembly / f64 QP" if args.mode == "vmc" else "f32 (f64 where the reference promotes)", "data": "synthetic",
           "config": {"workload": "%d A1 robots, %s" % (n, {"vmc": "ComputeContactForce + J^T f per robot", "frontend": "horizon %d front-end per robot" % h,
                                                             "estimator": "UpdateDataFlow kinematics + velocity estimator update per robot"}[args.mode]),
                      "robots_per_gpu": n},
           "roofline": {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                        "kernel_ms": kernel_ms, "algorithmic_bytes_per_robot": alg_bytes, "algorithmic_flop_per_robot": alg_flop,
                        "note": "launch- and latency-bound at this batch size, not bandwidth-bound"},
           "cpu_baseline": {"value": cpu_rate, "unit": unit, "cores": 1, "kind": "port",
                            "sample": "%d robot-ticks through the oracle's C++ restatement via ctypes, one thread" % m}}
    return out


def _pct(us):
    us = np.sort(np.asarray(us, np.float64))
    return dict(p50_us=float(us[us.size // 2]), p99_us=float(us[(us.size * 99) // 100]), mean_us=float(us.mean()), min_us=float(us[0]), calls=int(us.size))


def single_mode(args, pkg):
    """The drop-in boundary's real caller is ONE robot per process with a 1 ms tick budget at the reference's 500 Hz control rate (dt = 0.002 s;
    the MPC re-solves every 15th tick, qr_mpc_stance_leg_controller.cpp:342; the WBC runs on every other of the remaining ticks,
    qr_wbc_locomotion_controller.cpp:111).  Latency of qrgpu_mpc_solve1 and qrgpu_wbc_run1 -- host arrays in, one launch, host arrays out --
    per call, p50 / p99 over `--steps` calls (>= 1000) at h = 5 (the reference's hard-coded planning horizon,
    qr_mpc_stance_leg_controller.cpp:42), 10 and 16: (a) through the Python mirror of the reference interface (ctypes: a few us of
    interpreter per call on top), (b) through the C++ adapters of include/qrgpu_adapters.hpp in a compiled program (tests/stubs/adapter_demo
    --latency), which is what a maintainer's build would run.  Beside them the reference's own solver (qpOASES 3.2.0 from the reference tree, as
    called) and the port's CPU tick on the same inputs, one thread."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build()
    calls = max(1000, args.steps)
    so = pkg._build.build()
    exe = os.path.join(ROOT, "tests", "stubs", "adapter_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "stubs"),
                           os.path.join(ROOT, "tests", "stubs", "adapter_demo.cpp"), "-o", exe, so, "-Wl,-rpath," + os.path.dirname(so)])
    cfg, md = pkg.mpc_cfg("a1"), pkg.model_desc("a1")
    res = {}
    for h in (5, 10, 16):
        ctx = pkg.Context(device_id=0, max_batch=1, horizon_max=16)
        mpc = pkg.MPCInterface(ctx, 0)
        mpc.SetupProblem(cfg[0], h, cfg[1], cfg[2], cfg[3], cfg[4:7], cfg[7:19], cfg[19])
        ctx.wbc_setup_packed(0, md)
        b = pkg.make_batch(16, h, "a1", seed=0x51 + h, excite=args.excite)
        r = {}
        # (a) Python mirror: 16 different robots round robin (a fresh QP every call; the warm start sees another robot's working set)
        for kind in ("mpc", "wbc"):
            us = []
            prev = np.zeros(3, np.float32)
            for it in range(calls + 20):
                i = it % 16
                s_ = b["mpc_state"][i]
                t0 = time.perf_counter()
                if kind == "mpc":
                    mpc.SolveMPCKernel(s_[0:3], s_[3:6], s_[6:10], s_[10:13], s_[13:25].reshape(4, 3).T, s_[25:28], b["traj"][i], b["gait"][i])
                else:
                    ctx.wbc_run1(b["fb_state"][i], b["wbc_cmd"][i], prev)
                dt_ = time.perf_counter() - t0
                if it >= 20:
                    us.append(1e6 * dt_)
            r["python_%s" % kind] = _pct(us)
        ctx.close()
        # (b) the C++ adapters, robot 0's inputs on every call (the working set of the last call is this call's warm start, as on a real robot
        # between two MPC ticks 0.03 s apart)
        vals = [h] + list(cfg) + list(b["mpc_state"][0]) + list(b["traj"][0]) + list(b["gait"][0]) + list(b["fb_state"][0]) + list(b["wbc_cmd"][0])
        inp = " ".join(repr(float(v)) if not isinstance(v, int) else str(v) for v in vals)
        out = subprocess.run([exe, "--latency", str(calls)], input=inp.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        if out.returncode != 0:
            raise RuntimeError("adapter_demo failed: %s" % out.stderr.decode())
        for line in out.stdout.decode().splitlines():
            w = line.split()
            if w and w[0] in ("latency_mpc_us", "latency_wbc_us"):
                r["adapters_%s" % w[0][8:11]] = dict(p50_us=float(w[1]), p99_us=float(w[2]), mean_us=float(w[3]), min_us=float(w[4]), calls=calls)
        # the CPU side on the same 16 QPs, one thread
        A = O.mpc_constraint_matrix(h, float(cfg[1]))
        t_ref = t_port = t_wbc = 0.0
        capped = 0
        for i in range(16):
            H, g, ub = O.mpc_assemble(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            if O.ref() is not None:
                t0 = time.perf_counter()
                x, info = O.ref_qpoases_mpc(H.astype(np.float64), g.astype(np.float64), A, np.zeros(20 * h), ub.astype(np.float64), 100)
                t_ref += time.perf_counter() - t0
                capped += int(info["nWSR"] >= 100 or info["init_rc"] != 0)
            t0 = time.perf_counter()
            O.mpc_solve(cfg, h, b["mpc_state"][i], b["traj"][i], b["gait"][i])
            t_port += time.perf_counter() - t0
            t0 = time.perf_counter()
            O.wbc_run(md, b["fb_state"][i], b["wbc_cmd"][i], dtype=np.float32)
            t_wbc += time.perf_counter() - t0
        r["cpu"] = dict(reference_qpoases_as_called_ms=(1e3 * t_ref / 16) if O.ref() is not None else None, reference_hit_nwsr_100=capped,
                        port_mpc_assemble_solve_ms=1e3 * t_port / 16, port_wbc_tick_ms=1e3 * t_wbc / 16)
        res["h%d" % h] = r
    a10 = res["h10"]["adapters_mpc"]
    out = {"metric": "single-robot MPC solve latency through the drop-in adapters, h = 10 (p50)", "value": a10["p50_us"], "unit": "us", "n_gpus": 1,
           "steps": calls, "warmup": 20, "ms_per_step": a10["mean_us"] * 1e-3, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 assembly / f64 QP+WBC", "data": "synthetic",
           "config": {"workload": "ONE A1 robot per call through qrgpu_mpc_solve1 / qrgpu_wbc_run1 (host arrays in and out, one launch, synchronised): the reference's "
                                  "own calling pattern", "calls": calls, "excite": args.excite,
                      "staging": "zero copy: pinned, mapped host block read and written by the kernel in place" if os.environ.get("QRGPU_SINGLE_COPIES", "0") in ("", "0")
                                 else "three hipMemcpyAsync per call (QRGPU_SINGLE_COPIES=1)",
                      "latency": res,
                      "tick_budget_us": 1000.0,
                      "budget_note": "the reference's control tick is 2 ms of simulated time but its README quotes a 1 kHz-class loop; every figure here is to be read "
                                     "against 1000 us per tick"},
           "roofline": {"bound": "latency", "kernel": "qr_mpc_kernel<2,BIG,.,512> (one workgroup)", "achieved": None, "peak": None, "unit": "us", "frac": None, "traffic": None,
                        "note": "one workgroup on one CU: the call is launch + one robot's dependent chain + completion; no throughput roof applies"}}
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters: this parent never initialises a GPU),
    one per device, with the environment torch.distributed.run would give them; rank 0's stdout is this process's stdout.  Exit status: 0
    only when every rank exited 0 -- a failed rank takes the others down and the run is a failure, never a silent 1-GPU run."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   QRGPU_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = None
    # an overall deadline as well: ranks that all hang (a collective that never completes) are killed -- fresh children only, this parent never
    # touched a GPU -- and the run fails instead of sitting there (QRGPU_BENCH_DEADLINE_S, default 1500 s)
    overall = time.time() + float(os.environ.get("QRGPU_BENCH_DEADLINE_S", "1500"))
    while procs:
        if time.time() > overall and rc == 0:
            print("bench.py: the ranks did not finish within the deadline: killing them", file=sys.stderr)
            rc = 124
            for p in procs:
                p.kill()
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                deadline = time.time() + 20.0          # the others are probably stuck in a rendezvous with the dead rank
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()
        time.sleep(0.05)
    if rc:
        print("bench.py: a rank process failed (exit %d): no result" % rc, file=sys.stderr)
    return rc


def predicted_weak_scaling(ctx, step, fence, reset, D, warmup, steps=32):
    """With no 8-GPU node to measure on: the eight draws stand in for eight ranks (a rank's population is one more draw of the same
    generator).  Per-step device times t_d(s) of every draw from marks on the stream (one event per step), then
      per_step  = mean over steps s of  mean_d t_d(s) / max_d t_d(s)   -- the ranks meeting at every step (the torque gather of step s needs all of them)
      whole_run = mean_d T_d / max_d T_d,  T_d = sum_s t_d(s)          -- the ranks only meeting at the end (gathers two steps behind the compute stream)
    The real run sits between the two; x D is the predicted speed-up over one GPU.  The collective itself (48 KB per rank, on its own stream) is not priced."""
    t = np.zeros((D, steps))
    for d in range(D):
        reset(d)
        for _ in range(warmup):
            step()
        fence()
        ctx.mark(0)
        for s_ in range(steps):
            step()
            ctx.mark(1 + s_)
        fence()
        t[d] = [ctx.mark_elapsed_ms(s_, s_ + 1) for s_ in range(steps)]
    per_step = float(np.mean(t.mean(0) / t.max(0)))
    whole = float(t.sum(1).mean() / t.sum(1).max())
    return dict(per_step_sync=per_step, whole_run=whole, draws=D, steps_per_draw=steps, step_ms_mean=float(t.mean()), step_ms_max=float(t.max()),
                predicted_speedup_at_8=[8.0 * per_step, 8.0 * whole] if D == 8 else None,
                what="eight draws as eight ranks, per-step device times from stream marks; efficiency if the ranks met at every step / only at the end "
                     "(the run is in between); the all-gather itself is not priced")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps in total, split over the draws")
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps in front of every draw's timed steps")
    ap.add_argument("--robots", type=int, default=1024, help="robots per GPU")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--excite", type=float, default=1.0)
    ap.add_argument("--draws", type=int, default=8, help="robot populations (generator seeds) the timed steps are split over; value = median")
    ap.add_argument("--seq", type=int, default=8, help="batches per temporally coherent sequence (walked back and forth)")
    ap.add_argument("--walk", default="pingpong", choices=["pingpong", "forward"],
                    help="how consecutive steps move through a sequence: back and forth (every step a neighbour of the last; on the way back time runs backwards), or forward only with one jump back to the start per --seq steps")
    ap.add_argument("--same-seed-ranks", action="store_true",
                    help="control experiment: every rank draws the same populations (identical work per GPU) instead of its own")
    ap.add_argument("--mode", default="tick", choices=["tick", "mpc", "wbc", "vmc", "frontend", "estimator", "single"],
                    help="tick = the headline; vmc / frontend / estimator = the SURVEY 8f rows, single GPU; single = single-robot latency of the drop-in calls")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements (slot-order dispatch, K12 off, replayed batch, PCIe, 8f kernels, scaling prediction)")
    ap.add_argument("--trot-only", action="store_true", help="experiment: no all-stance / three-leg robots in the batch")
    ap.add_argument("--hessian", default="f32", choices=["f32", "bf16x3"],
                    help="K4 arithmetic: f32 = exact fp32 matrix instruction (default); bf16x3 = three-limb bf16 on the bf16 matrix cores (BASELINE.json configs[4])")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE.json configs[4] per GPU: A1 and Lite3 interleaved (type_id per robot), usually with --horizon 16; mode tick only")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    # The one JSON line is the only thing this program may put on its standard output -- and libraries it loads write there at C level (RCCL
    # prints a five-line banner when its first communicator comes up, gloo announces its connections): from here on file descriptor 1 IS
    # standard error, and emit() writes the line to the real one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a launcher's world size is the number of GPUs of this run; a mismatch is a mistake of the command line, not something to paper over
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # QRGPU_BENCH_REHEARSAL=1: every rank on device 0, the gather through gloo on host copies -- a functional rehearsal of the N > 1 code
    # path on a one-GPU box (RCCL refuses two ranks on one device); its numbers mean nothing
    rehearsal = os.environ.get("QRGPU_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    grp = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # launcher plumbing on the CPU, over a localhost socket (quadruped-robot_amd/rendezvous.py): barrier, max-reduce of times, the communicator
        # id -- no GPU array library is imported at any N
        grp = _load_rendezvous().Group(rank, world)
    dry = os.environ.get("QRGPU_BENCH_DRY", "")
    if dry:
        # launcher self-test (tests/test_bench_launch.py, no GPU): rendezvous, one max-reduce, one JSON line; "fail<r>" makes rank r exit 3 first
        if dry == "fail%d" % rank:
            sys.exit(3)
        if dry == "hang":
            time.sleep(3600)
        v = float(rank + 1)
        if os.environ.get("QRGPU_BENCH_DRY_ASSERT_NO_TORCH") and "torch" in sys.modules:
            sys.exit(5)
        if os.environ.get("QRGPU_BENCH_DRY_NOISE"):
            os.write(1, b"a library's banner on file descriptor 1\n")      # what RCCL does when its first communicator comes up (the launcher test)
        if world > 1:
            v = grp.allreduce_max(v)
            grp.barrier(); grp.close()
        if rank == 0:
            emit({"metric": "launcher self-test", "value": v, "n_gpus": world, "dry": True})
        return

    pkg = _load_pkg()
    if rank == 0 or (local_rank == 0 and not rehearsal):
        pkg._build.build()          # one build per node; the other ranks wait (the build is also file-locked)
    if world > 1:
        grp.barrier()
    if args.mode == "single":
        emit(single_mode(args, pkg))
        return
    n, h = args.robots, args.horizon
    ctx = pkg.Context(device_id=local_rank, max_batch=n, horizon_max=16)   # raises without gfx950 / built library
    ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h)
    ctx.wbc_setup_packed(0, pkg.model_desc("a1"))

    if args.mode in ("vmc", "frontend", "estimator"):
        if rank == 0:
            emit(side_mode(args, pkg, ctx))
        ctx.close()
        return

    # QRGPU_BENCH_FORCE_COMM=1 (one GPU): a one-rank RCCL communicator and the whole exchange of the N > 1 path -- fence, gather behind every tick,
    # check of the gathered block -- with world = 1: what that path costs a rank's ticks, measured where there is only one GPU
    comm_on = (world > 1 and not rehearsal) or (world == 1 and os.environ.get("QRGPU_BENCH_FORCE_COMM") == "1")
    if world > 1 and not rehearsal:
        ctx.comm_init_rank(pkg.rendezvous.exchange_comm_id(grp, pkg.qrgpu.comm_unique_id), world, rank)     # RCCL communicator owned by the context
    elif comm_on:
        ctx.comm_init_rank(pkg.qrgpu.comm_unique_id(), 1, 0)
    ctx.set_torque_epilogue(hip_comp=True, clip=True)          # K14 tail is part of the tick SURVEY 8(d) defines
    ctx.set_hessian_mode(args.hessian)

    # ---- populations: D draws x a coherent sequence each, all resident in HBM before anything is timed -----------------------
    # (at least five timed steps per draw: a draw's barrier-to-barrier region starts with an idle GPU, and two or three steps -- the driver's
    #  --steps 20 over eight draws -- measure that start-up as much as the tick: four draws then)
    D = max(1, min(args.draws, max(1, args.steps // 5)))
    SEQ = max(1, args.seq)
    S = pkg.to_soa
    dv = Dev(pkg, ctx)
    T = dv.soa
    seed_rank = 0 if args.same_seed_ranks else rank
    extra = dict(frac_all_stance=0.0, frac_three_leg=0.0) if args.trot_only else {}
    d_type = None
    if args.mixed:
        if args.mode != "tick" or n % 2:
            raise SystemExit("--mixed needs --mode tick and an even --robots")
        ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h)
        ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
        d_type = dv.put(pkg.shard.interleave_types(n, 2))

    def population(d):
        seed = 0xA1 + 2 + 1000 * d + 100000 * seed_rank          # draw 0 of rank 0 is make_batch(seed = 0xA1 + 2): BASELINE configs[2]'s seed rule
        if not args.mixed:
            return pkg.make_batch_sequence(n, h, "a1", seed=seed, steps=SEQ, excite=args.excite, **extra)
        sa = pkg.make_batch_sequence(n // 2, h, "a1", seed=seed, steps=SEQ, excite=args.excite)
        sl = pkg.make_batch_sequence(n // 2, h, "lite3", seed=seed + 0xD2, steps=SEQ, excite=args.excite)
        out = []
        for ba, bl in zip(sa, sl):
            b = dict(ba); b["n"] = n
            for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
                b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
            out.append(b)
        return out

    keys = ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd")
    host0 = None
    dev_seq = []
    for d in range(D):
        seq = population(d)
        if d == 0:
            host0 = seq
        dev_seq.append([[T(b[k]) for k in keys] for b in seq])
    walk = list(range(SEQ)) + list(range(SEQ - 2, 0, -1))        # 0 1 .. S-1 S-2 .. 1 | 0 1 ..: every step's batch is a neighbour of the last one
    if args.walk == "forward":
        walk = list(range(SEQ))                                  # 0 1 .. S-1 | 0 1 ..: time only runs forwards, one discontinuity per SEQ steps
    d_prev = dv.zeros((3, n))
    # Output arrays are double-buffered by step parity: consecutive ticks write different arrays, which is what lets the library start tick
    # t + 1's solves in the slots tick t's drain leaves empty (qrgpu_set_tick_overlap, include/qrgpu.h; QRGPU_BENCH_OVERLAP=0: one set of
    # arrays, the mode off -- rounds 1-3's form).  The mode is a promise about INPUTS too: every batch of every sequence is resident in HBM
    # before the first timed step (above), and nothing is queued on the context's stream between ticks.
    # (one rank only: with the exchange of N > 1 in the loop -- a fence in front of and a gather behind every tick -- it gains nothing, measured with
    #  a one-rank communicator: 4.45 M ticks/s either way -- and the first run on several GPUs is to be a boring one)
    want_overlap = args.mode == "tick" and world == 1 and not comm_on and os.environ.get("QRGPU_BENCH_OVERLAP", "1") != "0"
    overlap_on = want_overlap and ctx.set_tick_overlap(True, strict=False)
    nbuf = 2 if (overlap_on or world > 1 or comm_on) else 1
    d_force2 = [dv.zeros((12, n)) for _ in range(nbuf)]
    d_qdes2 = [dv.zeros((24, n)) for _ in range(nbuf)]
    d_status2 = [dv.zeros((n,), np.int32) for _ in range(nbuf)]
    d_force, d_qdes, d_status = d_force2[0], d_qdes2[0], d_status2[0]
    d_tau2 = [dv.zeros((12, n)) for _ in range(nbuf)]
    d_tau_all = dv.zeros((world, 12, n)) if (world > 1 or comm_on) else None   # rank-major
    nstep = [0]
    cur = dict(draw=0, k12=True, fixed=None)

    def step():
        i = nstep[0]
        nstep[0] += 1
        slot = i & 1 if nbuf == 2 else 0
        tau = d_tau2[slot]
        d_force, d_qdes, d_status = d_force2[slot], d_qdes2[slot], d_status2[slot]
        ds, dt_, dg, dfb, dcmd = dev_seq[cur["draw"]][cur["fixed"] if cur["fixed"] is not None else walk[i % len(walk)]]
        if comm_on:
            ctx.allgather_fence(slot)                          # the gather of two steps ago has finished reading this buffer
        if args.mode == "tick":
            ctx.tick_batch(n, ds, dt_, dg, dfb, dcmd, d_prev, d_force, tau, d_status, d_type, qdes=d_qdes if cur["k12"] else None)
        elif args.mode == "mpc":
            ctx.mpc_solve_batch(n, ds, dt_, dg, dfb.row(13), d_force, tau, d_status)
        else:
            ctx.wbc_run_batch(n, dfb, dcmd, d_prev, tau, d_qdes, d_status)
        if world > 1 or comm_on:
            if rehearsal:
                ctx.sync()
                parts = grp.allgather_bytes(np.ascontiguousarray(tau.download(), np.float32).tobytes())
                d_tau_all.upload(np.stack([np.frombuffer(p_, np.float32).reshape(12, n) for p_ in parts]))
            else:
                ctx.allgather_tau(tau, n, d_tau_all, slot, of_tick=(args.mode == "tick"))     # RCCL over xGMI on the context's own stream: the only exchange of the path

    def last_slot():
        return (nstep[0] - 1) & 1 if nbuf == 2 else 0

    def fence():
        if comm_on:
            ctx.comm_sync()
        if world > 1:
            ctx.sync()
            grp.barrier()
        ctx.sync()

    def reset(draw, k12=True, fixed=None, lpt=True):
        cur.update(draw=draw, k12=k12, fixed=fixed)
        ctx.set_lpt_schedule(lpt)                               # forgets the dispatch history: another population
        d_prev.zero()
        nstep[0] = 0

    def timed(steps, warmup, draw, k12=True, fixed=None, lpt=True):
        """`warmup` untimed then `steps` timed steps on one population with a fresh scheduler history.  -> seconds (max over ranks)"""
        reset(draw, k12, fixed, lpt)
        for _ in range(warmup):
            step()
        fence()
        if cur.get("ktime"): ctx.enable_timing(cur["ktime"])
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        el = time.perf_counter() - t0
        if cur.get("ktime"): ctx.enable_timing(-1)
        if world > 1:
            el = grp.allreduce_max(el)
        return el

    # ---- the timed region: K steps split over the draws -------------------------------------------------------------------
    share = [args.steps // D + (1 if d < args.steps % D else 0) for d in range(D)]
    # HIP events around the two kernels of every 8th step (every 4th / 2nd in shorter runs; on the context's stream).  Around every step they
    # cost 17 us of a 0.29 ms step (3.50 against 3.73 M ticks/s with none: QRGPU_BENCH_KERNEL_TIMING=0 / 1 / N for the A/B; every 4th 3.65, every
    # 8th 3.69 M)
    # -- and only around timed steps (the warm-up steps of a draw start cold)
    ktime = int(os.environ.get("QRGPU_BENCH_KERNEL_TIMING", "8" if args.steps >= 160 else "4"))
    ctx.enable_timing(0)
    if ktime:
        ctx.enable_timing(ktime); ctx.enable_timing(-1)        # the event pool is made here, paused: nothing of it inside a draw's barrier-to-barrier region
    if world > 1 or comm_on:
        # The first 8 steps of the run, untimed, every one of them checked: this rank's block of the gathered torques IS its local torque array of
        # that step, on every rank (a gather that reads a buffer too early, too late or from the wrong slot shows here, not in the rate).
        reset(0)
        for k_ in range(8):
            step()
            fence()
            own = d_tau_all.download()[rank]
            if not np.array_equal(own, d_tau2[last_slot()].download()):
                raise RuntimeError("rank %d, step %d: all-gathered torques differ from the local ones" % (rank, k_))
    draw_s, draw_flags, draw_itmax, draw_itmean = [], [], [], []
    cur["ktime"] = ktime
    for d in range(D):
        draw_s.append(timed(share[d], args.warmup, d))
        st_d = d_status2[last_slot()].download()          # (outside the timed region: the last step's status words of this draw)
        draw_flags.append(int((pkg.status_flags(st_d) != 0).sum())); draw_itmax.append(int(pkg.status_iterations(st_d).max()))
        draw_itmean.append(float(pkg.status_iterations(st_d).mean()))
    if world > 1 or comm_on:
        own = d_tau_all.download()[rank]
        if not np.array_equal(own, d_tau2[last_slot()].download()):
            raise RuntimeError("rank %d: all-gathered torques differ from the local ones" % rank)
    mpc_ms, mpc_cnt = ctx.get_timing(0)
    wbc_ms, wbc_cnt = ctx.get_timing(1)
    if not mpc_cnt and not wbc_cnt: mpc_ms = wbc_ms = float("nan")            # (QRGPU_BENCH_KERNEL_TIMING=0)
    ctx.enable_timing(False)
    cur["ktime"] = 0
    status = d_status2[last_slot()].download()
    rates = [world * n * share[d] / draw_s[d] for d in range(D)]
    value = float(np.median(rates))

    # diagnostic (QRGPU_BENCH_TIMELINE=1, stderr): where a pipelined tick's time goes, from the kernels' own stamps on the shared 100 MHz clock
    if os.environ.get("QRGPU_BENCH_TIMELINE") and args.mode == "tick" and world == 1:
        import ctypes as C_
        lib_ = ctx._lib
        lib_.qrgpu_debug_timeline.argtypes = [C_.c_void_p, C_.c_void_p]
        reset(int(os.environ.get("QRGPU_BENCH_TIMELINE_DRAW", "0")) % D)
        for _ in range(12): step()
        fence()
        if lib_.qrgpu_debug_timeline(ctx._h, None) != 0:
            raise SystemExit("QRGPU_BENCH_TIMELINE needs a library built with QRGPU_EXTRA_FLAGS=-DQR_TIMELINE")
        for _ in range(48): step()
        fence()
        tl = np.zeros((65, 8), np.int64)
        lib_.qrgpu_debug_timeline(ctx._h, tl.ctypes.data)
        ep = int(tl[64, 0])
        rows = []
        for e in range(ep - 40, ep - 1):
            a, b_ = tl[e & 63], tl[(e + 1) & 63]
            t0 = a[0]
            rows.append([(a[1] - t0) / 100, (a[2] - t0) / 100, (a[3] - t0) / 100, (a[4] - t0) / 100, (a[5] - t0) / 100, (a[6] - t0) / 100,
                         (a[7] - t0) / 100 if a[7] else float("nan"), (b_[0] - t0) / 100])
        rows = np.array(rows)
        rows = rows[np.isfinite(rows[:, 0])]
        names = ["last main start", "last solve published", "first WBC workgroup", "last WBC workgroup done", "trailing launch starts", "trailing launch ends",
                 "second WBC pass ends", "NEXT tick's first main workgroup"]
        print("timeline of a pipelined tick (us after its first main-pass workgroup; median / mean over %d ticks):" % len(rows), file=sys.stderr)
        for k, nm in enumerate(names):
            print("   %-34s %7.1f %7.1f" % (nm, np.nanmedian(rows[:, k]), np.nanmean(rows[:, k])), file=sys.stderr)
        tr = np.zeros((4, n), np.int32)
        lib_.qrgpu_debug_timeline_robots.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_int]
        if lib_.qrgpu_debug_timeline_robots(ctx._h, tr.ctypes.data, n) == 0:
            t0_ = int(tl[(ep) & 63][0]) & 0xffffffff
            rel = ((tr.astype(np.int64) - t0_ + (1 << 31)) % (1 << 32) - (1 << 31)) / 100.0        # us after the last tick's first main workgroup
            ws, wf, we, ms = rel
            late = ws > ms
            print("   last tick, per robot: WBC workgroup started after its solve was published: %d of %d; of those, start - published: median %.1f max %.1f us" % (
                int(late.sum()), n, np.median((ws - ms)[late]) if late.any() else 0, (ws - ms).max()), file=sys.stderr)
            print("      flag seen - max(published, WBC start + 0): median %.1f  p99 %.1f us;  done - flag seen (QP + store): median %.1f p90 %.1f max %.1f us;  WBC start -> flag wait begins is not stamped" % (
                np.median(wf - np.maximum(ms, ws)), np.percentile(wf - np.maximum(ms, ws), 99), np.median(we - wf), np.percentile(we - wf, 90), (we - wf).max()), file=sys.stderr)
            print("      done - published: median %.1f p90 %.1f max %.1f us;  the last 5 robots done: (published, WBC start, flag seen, done) %s" % (
                np.median(we - ms), np.percentile(we - ms, 90), (we - ms).max(), [(round(ms[k], 1), round(ws[k], 1), round(wf[k], 1), round(we[k], 1)) for k in np.argsort(-we)[:5]]), file=sys.stderr)
            hist = np.histogram(ws, bins=np.arange(130, 240, 10))[0]
            print("      WBC workgroup starts per 10 us from 130 us: %s;  solves published per 10 us from 130: %s" % (hist.tolist(), np.histogram(ms, bins=np.arange(130, 240, 10))[0].tolist()), file=sys.stderr)
        print("   gap last WBC done -> next main start: median %.1f us;  last solve -> last WBC done: median %.1f us" % (
            np.median(rows[:, 7] - np.maximum(rows[:, 3], rows[:, 5])), np.median(rows[:, 3] - rows[:, 1])), file=sys.stderr)

    # executed arithmetic of the MPC kernel (qrgpu_enable_flop_count): four more, untimed, steps on draw 0 with the counters on
    flop = None
    if args.mode != "wbc":
        timed(2, args.warmup, 0)
        ctx.enable_flop_count(True)
        acc = {}
        for _ in range(4):
            step()
            for k, v in ctx.mpc_flop_counts().items():
                acc[k] = acc.get(k, 0.0) + v / 4.0
        ctx.enable_flop_count(False)
        fence()
        flop = acc
        # roofline.imbalance: what bounds the main pass is not a phase's mean but the packing of a thousand solves of very different length into 512
        # slots -- 1 - (sum of solve times / slots) / span, from the instrumented kernel's own stamps on the shared 100 MHz clock (four more untimed,
        # MPC-only launches on draw 0; h <= 11: two workgroups per CU)
        imbalance = None
        if args.mode == "tick" and h <= 11:
            import ctypes as C_
            lib_ = ctx._lib
            lib_.qrgpu_debug_cycles.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_int]
            lib_.qrgpu_debug_cycles(ctx._h, None, 0)
            vals = []
            for k_ in range(6):
                ds, dt_, dg, dfb, dcmd = dev_seq[0][walk[k_ % len(walk)]]
                ctx.mpc_solve_batch(n, ds, dt_, dg, dfb.row(13), d_force2[0], d_tau2[0], d_status2[0])
                ctx.sync()
                if k_ < 2:
                    continue
                buf = np.zeros((n, 16), np.int64)
                lib_.qrgpu_debug_cycles(ctx._h, buf.ctypes.data, n)
                t0_, t6_ = buf[:, 12].astype(np.float64), buf[:, 13].astype(np.float64)
                good = t6_ > t0_
                span = t6_[good].max() - t0_[good].min()
                slots = 2 * ctx.device_info()["cus"]
                vals.append(dict(imb=1.0 - ((t6_[good] - t0_[good]).sum() / slots) / span, span_us=span / 100.0, mean_us=float((t6_[good] - t0_[good]).mean() / 100.0),
                                 max_us=float((t6_[good] - t0_[good]).max() / 100.0)))
            lib_.qrgpu_debug_cycles(ctx._h, None, -1)
            imbalance = {"value": float(np.mean([v["imb"] for v in vals])), "span_us": float(np.mean([v["span_us"] for v in vals])),
                         "mean_solve_us": float(np.mean([v["mean_us"] for v in vals])), "longest_solve_us": float(np.mean([v["max_us"] for v in vals])),
                         "what": "1 - (sum of the robots' solve times / resident slots) / span of the main pass, mean over 4 instrumented MPC launches of draw 0: the share of "
                                 "the dominant launch's slot-time that is empty (coarse bin packing: ~2 rounds of solves of 40-140 us)"}

    side = {}
    if world == 1 and not args.no_side and args.mode == "tick":
        ks = max(10, args.steps // 4)
        side["ticks_per_s_slot_order_dispatch"] = n * ks / timed(ks, 3, 0, lpt=False)          # no scheduling history at all
        side["ticks_per_s_without_k12"] = n * ks / timed(ks, args.warmup, 0, k12=False)
        side["ticks_per_s_same_batch_replayed"] = n * ks / timed(ks, args.warmup, 0, fixed=0)   # round 1's methodology (history = replay)
        if overlap_on:
            # the same draw with consecutive ticks NOT overlapped (every tick waits for its predecessor's join: round 3's pipelined tick)
            ctx.set_tick_overlap(False)
            side["ticks_per_s_no_tick_overlap"] = n * ks / timed(ks, args.warmup, 0)
            ctx.set_tick_overlap(True)
        # one tick on its own, from its call to its join on the context's stream (stream marks; warm history, nothing queued behind it): what a caller
        # that needs tick t's torques before it can state tick t + 1 waits for.  `value` is a rate of ticks in flight, not the reciprocal of this.
        timed(3, args.warmup, 0)
        lat = []
        for _ in range(12):
            ctx.tick_fence() if overlap_on else None
            ctx.mark(0); step(); ctx.mark(1); ctx.sync()
            lat.append(ctx.mark_elapsed_ms(0, 1))
        side["tick_latency_ms"] = float(np.median(lat))
        # the serial tick (WBC launch behind the MPC launches on one stream, rounds 1-2's form) on the same draw, with the two kernels' own times
        ctx.set_tick_pipeline(False)
        ctx.enable_timing(4); ctx.enable_timing(-1)
        cur["ktime"] = 4
        side["ticks_per_s_serial_tick"] = n * ks / timed(ks, args.warmup, 0)
        cur["ktime"] = 0
        side["serial_tick_kernel_ms"] = {"mpc": ctx.get_timing(0)[0], "wbc": ctx.get_timing(1)[0]}
        ctx.enable_timing(False)
        ctx.set_tick_pipeline(True)
        # the eight draws as eight ranks: what 8 GPUs would lose to the spread between populations
        side["predicted_weak_scaling_8"] = predicted_weak_scaling(ctx, step, fence, reset, D, min(args.warmup, 10))
        # PCIe-inclusive rate (never `value`): pinned host buffers in, torques out, every step
        host_in = []
        for k in keys:
            pa = ctx.alloc_pinned(S(host0[0][k]).shape); pa.array[...] = S(host0[0][k]); host_in.append(pa)
        host_tau = ctx.alloc_pinned((12, n))
        reset(0, True, 0)
        ctx.sync()
        tp0 = time.perf_counter()
        for _ in range(20):
            for hsrc, ddst in zip(host_in, dev_seq[0][0]):
                ddst.copy_from_pinned(hsrc)
            step()
            d_tau2[last_slot()].copy_to_pinned(host_tau)
            ctx.sync()
        side["pcie_inclusive_ticks_per_s"] = n * 20 / (time.perf_counter() - tp0)
        # SURVEY 8f kernels in front of the tick, timed on their own (never part of `value`)
        fe, fst = pkg.workload.make_frontend_batch(n, seed=0xFE)
        d_fe, d_fst = T(fe), T(fst)
        d_traj2, d_gait2, d_cmd2 = dv.zeros((12 * h, n)), dv.zeros((4 * h, n)), dv.zeros((67, n))
        d_upd = dv.zeros((n,), np.int32)
        for _ in range(5):
            ctx.mpc_frontend_batch(n, d_fe, d_fst, d_traj2, d_gait2, d_cmd2, d_upd)
        ctx.mark(0)
        for _ in range(50):
            ctx.mpc_frontend_batch(n, d_fe, d_fst, d_traj2, d_gait2, d_cmd2, d_upd)
        ctx.mark(1)
        side["frontend_kernel_us"] = 1e3 * ctx.mark_elapsed_ms(0, 1) / 50
        ctx.vmc_setup_packed(0, pkg.workload.vmc_cfg("a1"), pkg.model_desc("a1")[:3])
        vin, vq = pkg.workload.make_vmc_batch(n, seed=0xB2)
        d_vin, d_vq = T(vin), T(vq)
        d_vf, d_vt = dv.zeros((12, n)), dv.zeros((12, n))
        d_vs = dv.zeros((n,), np.int32)
        for _ in range(3):
            ctx.vmc_force_batch(n, d_vin, d_vq, d_vf, d_vt, d_vs)
        ctx.mark(0)
        for _ in range(20):
            ctx.vmc_force_batch(n, d_vin, d_vq, d_vf, d_vt, d_vs)
        ctx.mark(1)
        side["vmc_qp_kernel_us"] = 1e3 * ctx.mark_elapsed_ms(0, 1) / 20

    if rank == 0:
        iters = pkg.status_iterations(status).astype(np.float64)
        flags = pkg.status_flags(status)
        ms_per_step = 1e3 * world * n / value
        ms_pooled = 1e3 * sum(draw_s) / args.steps
        it_mean = float(np.mean(draw_itmean)) if args.mode != "wbc" else 0.0
        # SURVEY 8(d)'s dense count with the change count this run measured (mean over the draws' last steps; a cold start takes 20.3)
        flop_dense = FLOP_K4_HESSIAN + FLOP_K4_GRADIENT + FLOP_K6_FACTOR + it_mean * FLOP_K6_PER_ITER
        traffic, traffic_src = committed_traffic("qr_mpc_kernel" if args.mode != "wbc" else "qr_wbc_kernel", n, h)
        if args.mode == "wbc":
            dom_ms, dom_name = wbc_ms, "qr_wbc_kernel"
            roof = {"bound": "latency", "kernel": dom_name, "achieved": FLOP_WBC * n / (dom_ms * 1e-3) / 1e12, "peak": PEAK_F64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                    "frac": FLOP_WBC * n / (dom_ms * 1e-3) / 1e12 / PEAK_F64_VECTOR_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                    "kernel_ms": dom_ms, "kernel_launches": wbc_cnt,
                    "note": "SURVEY 8(d)'s estimate of 0.35 MFLOP per robot (fp64); one wavefront per robot, dependent small-matrix chains"}
        else:
            dom_ms, dom_name = mpc_ms, "qr_mpc_kernel"
            sec = dom_ms * 1e-3
            f64 = (flop["fp64_sweep"] + flop["fp64_active_set"]) / sec / 1e12
            m32 = flop["fp32_matrix"] / sec / 1e12
            v32 = flop["fp32_vector"] / sec / 1e12
            roof = {
                # what bounds the kernel: dependent chains (a barrier + LDS round trip + 3x3 inverse per sweep pivot, ~4 k cycles of wave 0 per
                # working-set change), not a throughput roof -- DESIGN.md 5 holds the counter evidence (SQ_INSTS_VALU per SIMD-cycle, SQ_WAIT_*,
                # SQ_VALU_MFMA_BUSY_CYCLES) from profiles/; every figure here is EXECUTED arithmetic counted by the kernel itself
                "bound": "latency", "kernel": dom_name,
                "achieved": f64, "peak": PEAK_F64_VECTOR_TFLOPS, "unit": "TFLOP/s", "frac": f64 / PEAK_F64_VECTOR_TFLOPS,
                "achieved_is": "executed fp64 flops (block sweep, x0, active set) / kernel time; fp64 is >= 85 % of the kernel's cycles",
                "fp32_matrix": {"achieved": m32, "peak": PEAK_F32_MATRIX_TFLOPS, "frac": m32 / PEAK_F32_MATRIX_TFLOPS,
                                "what": "v_mfma_f32_16x16x4_f32 flops issued by K4 (2*16*16*4 each, zero padding of the tiles included) = MFMA utilisation over the whole kernel"},
                "fp32_vector": {"achieved": v32, "peak": PEAK_F32_MATRIX_TFLOPS, "frac": v32 / PEAK_F32_MATRIX_TFLOPS},
                "executed_flop_per_robot": {k: v / n for k, v in flop.items()},
                "dense_yardstick": {"flop_per_robot": flop_dense, "working_set_changes": it_mean, "achieved": flop_dense * n / sec / 1e12, "peak": PEAK_F32_MATRIX_TFLOPS,
                                    "frac": flop_dense * n / sec / 1e12 / PEAK_F32_MATRIX_TFLOPS,
                                    "what": "SURVEY.md 8(d)'s dense count (12h x 13h x 12h GEMM, n^3/3 factorisation, 4 n^2 per change at the change count this run "
                                            "measured) over the same kernel time: a yardstick against that accounting, not work the kernel does (swing variables are "
                                            "eliminated, zero terms skipped)"},
                "traffic": traffic, "traffic_source": traffic_src,
                "imbalance": imbalance,
                "kernel_ms": dom_ms, "kernel_launches": mpc_cnt, "other_kernel_ms": wbc_ms,
                "other_kernel_ms_is": "span of the WBC launch on its own stream; in the pipelined tick it runs BESIDE the MPC launches and includes the wait for "
                                      "each robot's forces (config.serial_tick_kernel_ms.wbc is the kernel on its own)",
                "kernel_ms_is": "mean over the launches bracketed by HIP events on the launching stream: every %d-th timed step" % ktime,
                "outside_kernels_ms": (ms_pooled - mpc_ms) if (args.mode == "tick" and not overlap_on) else None,
                "kernel_alone": ({"kernel_ms": side["serial_tick_kernel_ms"]["mpc"], "frac": f64 * sec / (side["serial_tick_kernel_ms"]["mpc"] * 1e-3) / PEAK_F64_VECTOR_TFLOPS,
                                  "what": "the same launch with the machine to itself (serial tick, config.serial_tick_kernel_ms): with overlapped ticks `kernel_ms` is the "
                                          "span of a launch that shares the machine with its predecessor's drain and its successor's first workgroups"}
                                 if (overlap_on and side.get("serial_tick_kernel_ms")) else None),
                "outside_kernels_is": "all timed steps pooled: ms per step minus the MPC main pass's mean time = what of a tick is not hidden behind the main pass "
                                      "(the tail of the WBC launch that runs beside it, the trailing list launch, the list-driven WBC pass, stream hand-overs); "
                                      "null with overlapped ticks, where a launch's span is longer than a step",
                "hbm_algorithmic_GBs": BYTES_PER_TICK * n / (dom_ms * 1e-3) / 1e9, "hbm_frac": BYTES_PER_TICK * n / (dom_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
        what = "full MPC+WBC tick (K1-K14: kinematic projection on, motor tail on)"
        out = {
            "metric": "MPC+WBC control ticks/s (batched robots)" if args.mode == "tick" else "%s-only control ticks/s (batched robots)" % args.mode.upper(),
            "value": value, "unit": "ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_pooled": ms_pooled, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16x3 Hessian (fp32 accumulate) / f32 assembly / f64 QP+WBC" if args.hessian == "bf16x3" else "f32 assembly / f64 QP+WBC"), "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[4] per GPU: %d A1 + %d Lite3 robots interleaved, horizon %d, %s (%s, fp64 QP)" % (n // 2, n // 2, h, what, "bf16x3 Hessian MFMA" if args.hessian == "bf16x3" else "fp32 assembly")) if args.mixed
                       else "BASELINE.json configs[2]: %d A1 robots per GPU, horizon %d, %s" % (n, h, what)
                       if args.mode == "tick" else "%d A1 robots per GPU, horizon %d, %s only" % (n, h, args.mode),
                       "robots_per_gpu": n, "horizon": h, "excite": args.excite,
                       "value_is": "median over %d draws (robot populations) of the draw's rate; each draw: %d warm-up + its share of the %d timed steps over a "
                                   "temporally coherent sequence of %d batches 0.03 s apart" % (D, args.warmup, args.steps, SEQ),
                       "draws": D, "ticks_per_s_min": float(min(rates)), "ticks_per_s_max": float(max(rates)), "ticks_per_s_per_draw": [float(r) for r in rates],
                       "ticks_per_s_all_steps": world * n * args.steps / sum(draw_s),
                       "rank_batches": "every rank draws the same populations (control)" if args.same_seed_ranks else "every rank draws its own populations",
                       "parallelism": ("robots sharded over %d GPU(s); qrgpu_allgather_tau_of_tick (RCCL, context-owned stream) overlapped with the next tick" % world) + (" [QRGPU_BENCH_FORCE_COMM: one-rank communicator, the exchange of the N > 1 path on one GPU]" if (comm_on and world == 1) else ""),
                       "host_plumbing": "no GPU array library: device / pinned buffers, stream, events and the collective behind the C ABI" +
                                        ("; the launcher's barrier / max-reduce / id hand-over over a localhost socket (rendezvous.py)" if world > 1 else "") + "; torch not imported at any N",
                       "mean_active_set_iterations": it_mean, "status_flags_nonzero": int((flags != 0).sum()),
                       "status_flags_nonzero_per_draw": draw_flags, "max_active_set_changes_per_draw": draw_itmax,
                       "dispatch": "longest-first from each robot's solve time of the previous steps, smoothed (running mean, weight 1/2: a prediction -- consecutive steps see different batches)"
                                   + ("; h > 11 from 3.5 robots per CU on: two workgroups per CU for robots whose inverse Hessian fits half a CU's LDS, the others and the "
                                      "tick's long poles (by that smoothed cost) on whole CUs beside them (QRGPU_H16_TWO)" if h > 11 and n >= 896 else ""),
                       "tick_form": "pipelined: the WBC launch of a tick runs on a stream of the context's own beside that tick's MPC launches and takes each robot's "
                                    "forces when its solve raises the robot's flag (qrgpu_set_tick_pipeline, default); outputs complete in stream order as before"
                                    + ("; consecutive ticks OVERLAP (qrgpu_set_tick_overlap): tick t + 1's solves start in the slots tick t's drain leaves empty, every "
                                       "robot behind its own tick-t solve; output arrays double-buffered by step parity, inputs resident before the timed region; "
                                       "`value` is a rate of ticks in flight -- config.tick_latency_ms is one tick on its own, config.ticks_per_s_no_tick_overlap the "
                                       "rate with every tick behind its predecessor's join" if overlap_on else ""),
                       "tick_overlap": bool(overlap_on),
                       **side},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and args.mixed and args.mode == "tick":
            # configs[4] per GPU: the CPU restatement type by type (it has one parameter set per call), and the error of this run's
            # arithmetic against the exact fp32 path on the same batch
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py as O
            O.build()
            cores = max(1, min(len(os.sched_getaffinity(0)), 64))
            b0 = host0[0]
            tid_h = pkg.shard.interleave_types(n, 2)
            f_cpu = np.zeros((n, 12), np.float32); tau_cpu = np.zeros((n, 12), np.float32); st_cpu = np.zeros(n, np.int32)
            wall = 0.0
            for robot, t in (("a1", 0), ("lite3", 1)):
                m = tid_h == t
                t0 = time.perf_counter()
                fr, tr, sr, _, _ = O.tick_batch(1, pkg.mpc_cfg(robot), h, pkg.model_desc(robot)[:3], pkg.model_desc(robot), b0["mpc_state"][m], b0["traj"][m], b0["gait"][m],
                                                b0["fb_state"][m], b0["wbc_cmd"][m], b0["prev_ori_vel"][m].copy(), nthreads=cores, epilogue=3)
                wall += time.perf_counter() - t0
                f_cpu[m], tau_cpu[m], st_cpu[m] = fr, tr, sr
            out["cpu_baseline"] = dict(value=n / wall, unit="ticks/s", cores=cores, kind="port",
                                       sample="one pass over the %d-robot mixed batch (draw 0, first batch), A1 and Lite3 halves in turn, %d threads" % (n, cores))
            ds, dt_, dg, dfb, dcmd = dev_seq[0][0]
            res = {}
            for mode in ("f32", args.hessian):
                ctx.set_hessian_mode(mode); ctx.set_warm_start(False)
                d_prev.zero()
                d_f1, d_t1, d_s1 = dv.zeros((12, n)), dv.zeros((12, n)), dv.zeros((n,), np.int32)
                ctx.tick_batch(n, ds, dt_, dg, dfb, dcmd, d_prev, d_f1, d_t1, d_s1, d_type, qdes=d_qdes)
                ctx.sync()
                res[mode] = (d_f1.download().T, d_t1.download().T, d_s1.download())
            ctx.set_warm_start(True)
            fg, tg, sg = res[args.hessian]
            ok = (pkg.status_flags(sg) == 0) & (st_cpu == 0)
            cfgo = out["config"]
            cfgo["max_rel_torque_err_vs_cpu"] = float((np.abs(tg - tau_cpu) / np.maximum(1.0, np.abs(tau_cpu))).max(1)[ok].max())
            cfgo["max_rel_force_err_vs_cpu"] = float((np.abs(fg - f_cpu).max(1) / np.maximum(1.0, np.abs(f_cpu).max(1)))[ok].max())
            cfgo["max_rel_torque_err_vs_fp32_path"] = float((np.abs(tg - res["f32"][1]) / np.maximum(1.0, np.abs(res["f32"][1]))).max())
            cfgo["robots_compared_with_cpu"] = int(ok.sum())
        if world == 1 and not args.no_cpu_baseline and not args.mixed and args.mode != "wbc":
            b0 = host0[0]
            out["cpu_baseline"] = cpu_baseline(pkg, b0, h, mode=0 if args.mode == "mpc" else 1)
            # BASELINE's metric quotes the torque error beside the rate: one more (untimed) call on draw 0's first batch from fresh WBC
            # memory, against what the CPU pass returned for the same inputs; robots flagged on either side are counted, not compared
            d_prev.zero()
            d_f1, d_t1, d_s1, d_q1 = dv.zeros((12, n)), dv.zeros((12, n)), dv.zeros((n,), np.int32), dv.zeros((24, n))
            ds, dt_, dg, dfb, dcmd = dev_seq[0][0]
            if args.mode == "tick":
                ctx.tick_batch(n, ds, dt_, dg, dfb, dcmd, d_prev, d_f1, d_t1, d_s1, d_type, qdes=d_q1)
            else:
                ctx.mpc_solve_batch(n, ds, dt_, dg, dfb.row(13), d_f1, d_t1, d_s1)
            ctx.sync()
            f_cpu, tau_cpu, st_cpu, q_cpu = cpu_baseline.outputs
            ok = (pkg.status_flags(d_s1.download()) == 0) & (st_cpu == 0)
            f_gpu, tau_gpu = d_f1.download().T, d_t1.download().T
            cfgo = out["config"]
            cfgo["max_rel_force_err_vs_cpu"] = float((np.abs(f_gpu - f_cpu).max(1) / np.maximum(1.0, np.abs(f_cpu).max(1)))[ok].max())
            cfgo["max_rel_torque_err_vs_cpu"] = float((np.abs(tau_gpu - tau_cpu) / np.maximum(1.0, np.abs(tau_cpu))).max(1)[ok].max())
            if args.mode == "tick":
                cfgo["max_abs_qdes_err_vs_cpu"] = float(np.abs(d_q1.download().T - q_cpu)[ok].max())
            cfgo["robots_compared_with_cpu"] = int(ok.sum())
            if getattr(reference_solver, "as_called", None) is not None:
                # against the reference's solver exactly as the reference calls it (asymmetric fp32 H, nWSR = 100): DESIGN.md 2
                idx, f_ref, nwsr, rc = reference_solver.as_called
                conv = (rc == 0) & (nwsr < 100) & ok[idx]
                rel = np.abs(f_gpu[idx] - f_ref).max(1) / np.maximum(1.0, np.abs(f_ref).max(1))
                cfgo["parity_vs_reference_as_called"] = {
                    "robots": int(idx.size), "reference_converged": int(conv.sum()), "reference_hit_nwsr_100": int((nwsr >= 100).sum()),
                    "max_rel_force_err": float(rel[conv].max()) if conv.any() else None,
                    "note": "first-step forces of this batch's first robots vs qpOASES 3.2.0 fed the asymmetric fp32 H with nWSR = 100 "
                            "(qr_mpc_interface.cpp:418-438): these 24 robots are rows 62-85 of tests/golden/mpc_golden.npz, where the full-tick torque is "
                            "checked too; north_star's 1e-4 is not met against this comparator -- nor by the reference against itself (DESIGN.md 2, "
                            "tests/golden/parity_as_called.json, tests/test_gpu_golden.py)"}
        emit(out)
    if world > 1:
        grp.barrier()
        grp.close()
    ctx.close()


def committed_traffic(kernel, n, h):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE,
    the gfx950 correction of MI355X_MICROARCH.md), which are separate profiler runs of this same command: profiles/r04_pmc_summary.json names
    the commit they were taken at.  Only quoted for the configuration they were collected on (1024 robots, h = 10)."""
    p = os.path.join(ROOT, "profiles", "r04_pmc_summary.json")
    if not os.path.exists(p):
        p = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")
    if not os.path.exists(p) or n != 1024 or h != 10:
        return None, None
    try:
        d = json.load(open(p))
        k = d["kernels"][kernel]
        return float(k["hbm_bytes_per_launch"]), os.path.relpath(p, ROOT) + " (commit %s; %s)" % (d.get("commit", "?"), d.get("recipe", "rocprofv3 --pmc passes of bench.py"))
    except Exception:
        return None, None


if __name__ == "__main__":
    main()
