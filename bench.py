#!/usr/bin/env python3
"""Headline benchmark: MPC+WBC control ticks/s over a batch of A1 robots (BASELINE.json configs[2]:
1024 A1 instances, horizon 10, full MPC+WBC tick per robot), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one qrgpu_tick_batch over this rank's 1024 robots (inputs already resident in HBM) and, for
N > 1, one RCCL all-gather of the per-robot torques on a second stream, overlapped with the next tick (weak scaling: every rank
owns 1024 robots; the timed region ends when the last gather has landed).
torch is plumbing only (device buffers, the stream, torch.distributed); the tick itself is two
hand-written HIP kernels behind the C ABI of include/qrgpu.h.  Rank 0 prints ONE JSON line.
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
PEAK_HBM_GBS = 8000.0


def _load_pkg():
    d = os.path.join(ROOT, "quadruped-robot_amd")
    spec = importlib.util.spec_from_file_location("quadruped_robot_amd", os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["quadruped_robot_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(pkg, b, horizon, mode=1):
    """The CPU restatement (oracle/, kind "port") timed on this box's host cores over a bounded sample."""
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
    sl = {k: (v[:128] if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    t0 = time.perf_counter()
    O.tick_batch(mode, cfg, horizon, md[:3], md, sl["mpc_state"], sl["traj"], sl["gait"], sl["fb_state"], sl["wbc_cmd"],
                 sl["prev_ori_vel"].copy(), nthreads=1)
    t1 = time.perf_counter() - t0
    passes, wall = 0, 0.0
    while wall < 4.0 and passes < 8:
        t0 = time.perf_counter()
        f_cpu, tau_cpu, st_cpu, _, _ = O.tick_batch(*args, b["prev_ori_vel"].copy(), nthreads=cores)
        wall += time.perf_counter() - t0
        passes += 1
    cpu_baseline.outputs = (f_cpu, tau_cpu, st_cpu)      # the checker's answer on this batch, for max_rel_*_err_vs_cpu
    return dict(value=passes * n / wall, unit="ticks/s", cores=cores, kind="port",
                sample="%d passes over the same %d-robot batch on %d threads (oracle/ CPU restatement, own fp64 active-set QP)"
                       % (passes, n, cores),
                single_thread_value=128 / t1)


def side_mode(args, pkg, ctx, torch, dev, stream):
    """The SURVEY 8f rows on one GPU: `vmc` = force-balance stance QP (ComputeContactForce), `frontend` = MPC front-end
    (SetupCommand/Run/UpdateMPC).  A step = one batched call over --robots robots; kernel time by events on the launch stream."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build()
    n, h = args.robots, args.horizon
    T = lambda a: torch.from_numpy(pkg.to_soa(a)).to(dev)
    W = pkg.workload
    if args.mode == "vmc":
        cfg, geom = W.vmc_cfg("a1"), pkg.model_desc("a1")[:3]
        ctx.vmc_setup_packed(0, cfg, geom)
        vin, q = W.make_vmc_batch(n, seed=0xB2)
        d_in, d_q = T(vin), T(q)
        d_f, d_t = torch.empty((12, n), dtype=torch.float32, device=dev), torch.empty((12, n), dtype=torch.float32, device=dev)
        d_s = torch.zeros((n,), dtype=torch.int32, device=dev)
        step = lambda: ctx.vmc_force_batch(n, d_in, d_q, d_f, d_t, d_s)
        cpu = lambda i: O.vmc_solve(cfg, geom, vin[i], q[i])
        alg_bytes = (37 + 12 + 12 + 12 + 1) * 4
        metric, unit, kernel = "force-balance QP solves/s (batched robots)", "solves/s", "qr_vmc_kernel"
        # assembly 12x12x6 + 6x12, inverse 12^3, ~20 working-set changes of ~4*12^2 (dense count, as SURVEY 8d does for the MPC QP)
        alg_flop = 2 * 12 * 12 * 6 + 2 * 6 * 12 + 12 ** 3 + 20 * 4 * 12 ** 2
    elif args.mode == "estimator":
        cfg = W.estimator_cfg("a1")
        xs, stamps = W.make_estimator_sequence(n, 4, seed=0xE5)
        d_in = T(xs[0]); d_tick = torch.from_numpy(stamps[0].astype(np.int64)).to(dev).to(torch.int32)
        S_ = ctx.estimator_state_doubles(int(cfg[6]))
        d_state = torch.zeros((S_, n), dtype=torch.float64, device=dev); d_out = torch.zeros((36, n), dtype=torch.float32, device=dev)
        step = lambda: ctx.estimator_update_batch(n, cfg, d_in, d_tick, d_state, d_out)
        seq = np.repeat(xs[0][None, :1], 200, 0)[:, 0]
        cpu_all = lambda: O.estimator_run(cfg, seq, (1000 + 2 * np.arange(200)).astype(np.uint32))
        cpu = None
        alg_bytes = (41 + 1 + 36) * 4 + 2 * (32 + 6) * 8
        metric, unit, kernel = "velocity-estimator robot-ticks/s (batched robots)", "robot-ticks/s", "qr_estimator_kernel"
        alg_flop = 0
    else:
        vin, st = W.make_frontend_batch(n, seed=0xFE)
        d_in, d_st = T(vin), T(st)
        d_traj = torch.zeros((12 * h, n), dtype=torch.float32, device=dev); d_gait = torch.zeros((4 * h, n), dtype=torch.float32, device=dev)
        d_cmd = torch.zeros((67, n), dtype=torch.float32, device=dev); d_u = torch.zeros((n,), dtype=torch.int32, device=dev)
        step = lambda: ctx.mpc_frontend_batch(n, d_in, d_st, d_traj, d_gait, d_cmd, d_u)
        cpu = lambda i: O.mpc_frontend(h, 2, vin[i], st[i])
        alg_bytes = (64 + 8 + 8 + 16 * h + 19 + 1) * 4
        metric, unit, kernel = "MPC front-end robot-ticks/s (batched robots)", "robot-ticks/s", "qr_frontend_kernel"
        alg_flop = 0
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--robots", type=int, default=1024, help="robots per GPU")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--excite", type=float, default=1.0)
    ap.add_argument("--mode", default="tick", choices=["tick", "mpc", "wbc", "vmc", "frontend", "estimator"],
                    help="tick = the headline; vmc / frontend = the SURVEY 8f rows (force-balance QP, MPC front-end), single GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compare-dispatch", action="store_true",
                    help="also time the loop with slot-order dispatch (no longest-first history); off by default so that a profile of the "
                         "default command holds only launches of the measured configuration")
    ap.add_argument("--trot-only", action="store_true", help="experiment: no all-stance / three-leg robots in the batch")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE.json configs[4] per GPU: A1 and Lite3 interleaved (type_id per robot), usually with --horizon 16; mode tick only")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run for N>1" % (args.gpus, world), file=sys.stderr)
        args.gpus = world
    # QRGPU_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend -- a functional rehearsal of the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); its numbers mean nothing
    rehearsal = os.environ.get("QRGPU_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    pkg = _load_pkg()
    if rank == 0 or (local_rank == 0 and not rehearsal):
        pkg._build.build()          # one build per node; the other ranks wait (the build is also file-locked)
    if world > 1:
        dist.barrier()
    n, h = args.robots, args.horizon
    ctx = pkg.Context(device_id=local_rank, max_batch=n, horizon_max=16)   # raises without gfx950 / built library
    ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h)
    ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    if args.mode in ("vmc", "frontend", "estimator"):
        if rank == 0:
            print(json.dumps(side_mode(args, pkg, ctx, torch, dev, stream)))
        ctx.close()
        return

    # every rank owns its own contiguous shard of the global robot population
    # Weak scaling wants the SAME work on every GPU.  A 1024-robot draw is not that: the kernel time is its slowest robot's solve, and
    # draws differ by 2.5x (seeds 0..7 of this generator: 0.29 - 0.78 ms per step on one GPU, the slow ones holding one robot that needs
    # the rescue pass) -- a max over ranks of different draws would measure sampling noise, not the system.  So every rank draws its
    # robots with the same seed; QRGPU_BENCH_SEED_RANK=r times draw r instead (scratch/rank_seeds.sh lists all eight).
    seed_rank = int(os.environ.get("QRGPU_BENCH_SEED_RANK", "0"))
    b = pkg.make_batch(n, h, "a1", seed=0xA1 + 2 + 1000 * seed_rank, excite=args.excite,
                       **(dict(frac_all_stance=0.0, frac_three_leg=0.0) if args.trot_only else {}))
    d_type = None
    if args.mixed:
        if args.mode != "tick" or n % 2:
            raise SystemExit("--mixed needs --mode tick and an even --robots")
        ctx.mpc_setup_packed(1, pkg.mpc_cfg("lite3"), h)
        ctx.wbc_setup_packed(1, pkg.model_desc("lite3"))
        ba = pkg.make_batch(n // 2, h, "a1", seed=0xA1 + 2 + 1000 * seed_rank, excite=args.excite)
        bl = pkg.make_batch(n // 2, h, "lite3", seed=0x173 + 1000 * seed_rank, excite=args.excite)
        for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
            b[k] = np.empty((n,) + ba[k].shape[1:], ba[k].dtype); b[k][0::2] = ba[k]; b[k][1::2] = bl[k]
        d_type = torch.from_numpy(pkg.shard.interleave_types(n, 2)).to(dev)
    S = pkg.to_soa
    T = lambda a: torch.from_numpy(S(a)).to(dev)
    d_state, d_traj, d_gait = T(b["mpc_state"]), T(b["traj"]), T(b["gait"])
    d_fb, d_cmd, d_prev = T(b["fb_state"]), T(b["wbc_cmd"]), T(b["prev_ori_vel"])
    d_force = torch.zeros((12, n), dtype=torch.float32, device=dev)
    d_tau = torch.zeros((12, n), dtype=torch.float32, device=dev)
    d_qdes = torch.zeros((24, n), dtype=torch.float32, device=dev)
    d_status = torch.zeros((n,), dtype=torch.int32, device=dev)
    d_tau_all = torch.zeros((world * 12, n), dtype=torch.float32, device=dev) if world > 1 else None   # rank-major [world][12][n]
    # N > 1: the all-gather of tick i runs on its own stream while tick i+1 computes (the ticks never wait for a collective; they only
    # wait, two ticks later, for the gather that is still reading the torque buffer they are about to overwrite)
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    d_tau2 = [d_tau, torch.zeros_like(d_tau)] if world > 1 else [d_tau]
    gathered = [None, None]
    nstep = [0]

    def step():
        par = nstep[0] & 1 if world > 1 else 0
        nstep[0] += 1
        tau = d_tau2[par]
        if gathered[par] is not None:
            stream.wait_event(gathered[par])
        if args.mode == "tick":
            ctx.tick_batch(n, d_state, d_traj, d_gait, d_fb, d_cmd, d_prev, d_force, tau, d_status, d_type)
        elif args.mode == "mpc":
            ctx.mpc_solve_batch(n, d_state, d_traj, d_gait, d_fb[13:25], d_force, tau, d_status)
        else:
            ctx.wbc_run_batch(n, d_fb, d_cmd, d_prev, tau, d_qdes, d_status)
        if world > 1:
            comm_stream.wait_stream(stream)                    # this tick's torques are complete
            with torch.cuda.stream(comm_stream):
                if rehearsal:
                    dist.all_gather(list(d_tau_all.view(world, 12, n).unbind(0)), tau)
                else:
                    dist.all_gather_into_tensor(d_tau_all, tau)    # RCCL over xGMI: the only exchange of the path
                ev = torch.cuda.Event()
                ev.record(comm_stream)
            gathered[par] = ev

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        # the last gather's block for this rank must be this rank's last torques (every rank checks its own block)
        own = d_tau_all.view(world, 12, n)[rank]
        if not torch.equal(own, d_tau2[(nstep[0] - 1) & 1]):
            raise RuntimeError("rank %d: all-gathered torques differ from the local ones" % rank)
    mpc_ms, mpc_cnt = ctx.get_timing(0)
    wbc_ms, wbc_cnt = ctx.get_timing(1)
    ctx.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same loop without the longest-first dispatch history (DESIGN.md "tail"): what a batch with no temporal coherence gets
    value_no_lpt = None
    if world == 1 and args.compare_dispatch:
        ctx.set_lpt_schedule(False)
        for _ in range(3):
            step()
        fence()
        tq0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        value_no_lpt = n * args.steps / (time.perf_counter() - tq0)
        ctx.set_lpt_schedule(True)

    # PCIe-inclusive rate (never `value`): host buffers in, torques out, every step
    pcie_value = None
    if world == 1 and args.mode == "tick":
        host_in = [torch.from_numpy(S(b[k])).pin_memory() for k in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd")]
        dev_in = [d_state, d_traj, d_gait, d_fb, d_cmd]
        host_tau = torch.empty((12, n), dtype=torch.float32).pin_memory()
        ctx.enable_timing(False)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(20):
            for hsrc, ddst in zip(host_in, dev_in):
                ddst.copy_(hsrc, non_blocking=True)
            step()
            host_tau.copy_(d_tau, non_blocking=True)
            torch.cuda.synchronize()
        pcie_value = n * 20 / (time.perf_counter() - tp0)

    # MPC front-end timed on its own: it is a streaming kernel in front of the tick, not part of `value`
    fe_us = None
    if world == 1:
        fe, fst = pkg.workload.make_frontend_batch(n, seed=0xFE)
        d_fe, d_fst = T(fe), T(fst)
        d_traj2, d_gait2, d_cmd2 = torch.empty_like(d_traj), torch.empty_like(d_gait), torch.empty_like(d_cmd)
        d_upd = torch.zeros((n,), dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            ctx.mpc_frontend_batch(n, d_fe, d_fst, d_traj2, d_gait2, d_cmd2, d_upd)
        e0.record(stream)
        for _ in range(50):
            ctx.mpc_frontend_batch(n, d_fe, d_fst, d_traj2, d_gait2, d_cmd2, d_upd)
        e1.record(stream)
        torch.cuda.synchronize()
        fe_us = 1e3 * e0.elapsed_time(e1) / 50
    # force-balance (VMC) stance QP, also timed on its own
    vmc_us = None
    if world == 1:
        ctx.vmc_setup_packed(0, pkg.workload.vmc_cfg("a1"), pkg.model_desc("a1")[:3])
        vin, vq = pkg.workload.make_vmc_batch(n, seed=0xB2)
        d_vin, d_vq = T(vin), T(vq)
        d_vf, d_vt = torch.empty((12, n), dtype=torch.float32, device=dev), torch.empty((12, n), dtype=torch.float32, device=dev)
        d_vs = torch.zeros((n,), dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ctx.vmc_force_batch(n, d_vin, d_vq, d_vf, d_vt, d_vs)
        e0.record(stream)
        for _ in range(20):
            ctx.vmc_force_batch(n, d_vin, d_vq, d_vf, d_vt, d_vs)
        e1.record(stream)
        torch.cuda.synchronize()
        vmc_us = 1e3 * e0.elapsed_time(e1) / 20

    status = d_status.cpu().numpy()
    iters = (status >> 8).astype(np.float64)
    flags = status & 0xff

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * n * args.steps / elapsed
        it_mean = float(iters.mean()) if args.mode != "wbc" else 0.0
        flop_mpc = FLOP_K4_HESSIAN + FLOP_K4_GRADIENT + FLOP_K6_FACTOR + it_mean * FLOP_K6_PER_ITER
        if args.mode == "wbc":
            dom_ms, dom_flop, dom_name = wbc_ms, FLOP_WBC, "qr_wbc_kernel"
        else:
            dom_ms, dom_flop, dom_name = mpc_ms, flop_mpc, "qr_mpc_kernel"
        achieved = (dom_flop * n) / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp) and n == 1024 and h == 10 and args.mode == "tick":     # the PMC passes profile the default command only
            try:
                traffic = json.load(open(tp)).get(dom_name)
            except Exception:
                traffic = None
        out = {
            "metric": "MPC+WBC control ticks/s (batched robots)" if args.mode == "tick" else "%s-only control ticks/s (batched robots)" % args.mode.upper(),
            "value": value, "unit": "ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 assembly / f64 QP+WBC", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[4] per GPU: %d A1 + %d Lite3 robots interleaved, horizon %d, full MPC+WBC tick (fp32 assembly, fp64 QP)" % (n // 2, n // 2, h)) if args.mixed
                       else "BASELINE.json configs[2]: %d A1 robots per GPU, horizon %d, full MPC+WBC tick" % (n, h)
                       if args.mode == "tick" else "%d A1 robots per GPU, horizon %d, %s only" % (n, h, args.mode),
                       "robots_per_gpu": n, "horizon": h, "excite": args.excite, "rank_batches": "every rank draws its robots with the same seed (identical work per GPU)", "parallelism": "robots sharded over %d GPU(s), all-gather of torques overlapped with the next tick" % world,
                       "mean_active_set_iterations": it_mean, "status_flags_nonzero": int((flags != 0).sum()),
                       "pcie_inclusive_ticks_per_s": pcie_value, "frontend_kernel_us": fe_us, "vmc_qp_kernel_us": vmc_us,
                       "dispatch": "longest-first from the previous step's per-robot solve time", "ticks_per_s_slot_order_dispatch": value_no_lpt},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": achieved, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MATRIX_TFLOPS, "traffic": traffic,
                         "kernel_ms": dom_ms, "kernel_launches": mpc_cnt if dom_name == "qr_mpc_kernel" else wbc_cnt,
                         "algorithmic_flop_per_robot": dom_flop, "other_kernel_ms": wbc_ms if dom_name == "qr_mpc_kernel" else mpc_ms,
                         "hbm_algorithmic_GBs": BYTES_PER_TICK * n / (ms_per_step * 1e-3) / 1e9, "hbm_frac": BYTES_PER_TICK * n / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
        if world == 1 and not args.no_cpu_baseline and not args.mixed:
            out["cpu_baseline"] = cpu_baseline(pkg, b, h, mode=0 if args.mode == "mpc" else 1) if args.mode != "wbc" else None
            if out["cpu_baseline"] is not None:
                # BASELINE's metric quotes the torque error beside the rate: one more (untimed) call from the batch's initial WBC memory,
                # against what the CPU pass above returned for the same inputs; robots either side flags are counted, not compared
                d_prev.copy_(T(b["prev_ori_vel"]))
                d_f1, d_t1, d_s1 = torch.zeros_like(d_force), torch.zeros_like(d_tau), torch.zeros_like(d_status)
                if args.mode == "tick":
                    ctx.tick_batch(n, d_state, d_traj, d_gait, d_fb, d_cmd, d_prev, d_f1, d_t1, d_s1)
                else:
                    ctx.mpc_solve_batch(n, d_state, d_traj, d_gait, d_fb[13:25], d_f1, d_t1, d_s1)
                torch.cuda.synchronize()
                f_cpu, tau_cpu, st_cpu = cpu_baseline.outputs
                ok = ((d_s1.cpu().numpy() & 0xff) == 0) & (st_cpu == 0)
                f_gpu, tau_gpu = d_f1.cpu().numpy().T, d_t1.cpu().numpy().T
                out["config"]["max_rel_force_err_vs_cpu"] = float((np.abs(f_gpu - f_cpu).max(1) / np.maximum(1.0, np.abs(f_cpu).max(1)))[ok].max())
                if args.mode == "tick":
                    out["config"]["max_rel_torque_err_vs_cpu"] = float((np.abs(tau_gpu - tau_cpu) / np.maximum(1.0, np.abs(tau_cpu))).max(1)[ok].max())
                out["config"]["robots_compared_with_cpu"] = int(ok.sum())
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
