"""Experiment: how much of the WBC kernel hides under the MPC kernel's tail when launched on a second, lower-priority stream?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_pkg
pkg = load_pkg()
n, h = 1024, 10
dev = torch.device("cuda", 0)
ctx = pkg.Context(0, n, 16)
ctx.mpc_setup_packed(0, pkg.mpc_cfg("a1"), h); ctx.wbc_setup_packed(0, pkg.model_desc("a1"))
b = pkg.make_batch(n, h, "a1", seed=0xA1 + 2, excite=1.0)
T = lambda a: torch.from_numpy(pkg.to_soa(a)).to(dev)
d_state, d_traj, d_gait, d_fb, d_cmd, d_prev = T(b["mpc_state"]), T(b["traj"]), T(b["gait"]), T(b["fb_state"]), T(b["wbc_cmd"]), T(b["prev_ori_vel"])
d_force = torch.zeros((12, n), device=dev); d_tau = torch.zeros((12, n), device=dev); d_tau2 = torch.zeros((12, n), device=dev); d_st = torch.zeros((n,), dtype=torch.int32, device=dev)
d_st2 = torch.zeros((n,), dtype=torch.int32, device=dev)
hi = torch.cuda.Stream(priority=-1); lo = torch.cuda.Stream(priority=0)
def seq():
    ctx.set_stream(hi.cuda_stream)
    ctx.mpc_solve_batch(n, d_state, d_traj, d_gait, d_fb[13:25], d_force, d_tau, d_st)
    ctx.wbc_run_batch(n, d_fb, d_cmd, d_prev, d_tau2, None, d_st2)
def par(first_wbc=False):
    ev = torch.cuda.Event()
    if first_wbc:
        ctx.set_stream(lo.cuda_stream); ctx.wbc_run_batch(n, d_fb, d_cmd, d_prev, d_tau2, None, d_st2); ev.record(lo)
        ctx.set_stream(hi.cuda_stream); ctx.mpc_solve_batch(n, d_state, d_traj, d_gait, d_fb[13:25], d_force, d_tau, d_st)
    else:
        ctx.set_stream(hi.cuda_stream); ctx.mpc_solve_batch(n, d_state, d_traj, d_gait, d_fb[13:25], d_force, d_tau, d_st)
        ctx.set_stream(lo.cuda_stream); ctx.wbc_run_batch(n, d_fb, d_cmd, d_prev, d_tau2, None, d_st2); ev.record(lo)
    hi.wait_event(ev)
def timeit(f, k=100):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / k
print("sequential            %.4f ms" % timeit(seq))
print("parallel (mpc first)  %.4f ms" % timeit(par))
print("parallel (wbc first)  %.4f ms" % timeit(lambda: par(True)))
print("sequential again      %.4f ms" % timeit(seq))
