"""Compile libqrgpu.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libqrgpu.so")
SOURCES = ["qr_mpc_kernel.hip", "qr_mpc_kernel_fl.hip", "qr_wbc_kernel.hip", "qr_wbc_kernel_dbg.hip", "qr_frontend_kernel.hip", "qr_vmc_kernel.hip", "qr_estimator_kernel.hip", "qrgpu_api.hip", "qrgpu_comm.hip"]
# The fp32 MPC assembly must execute exactly the written fmaf chain (bit-identical to the CPU oracle, see
# DESIGN.md "bit-exact assembly"): those functions carry `#pragma clang fp contract(off)`; everything else
# (fp64 sweep / active set / WBC) is free to fuse multiply-adds.  NB plain -ffp-contract=fast would IGNORE those pragmas.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast-honor-pragmas", "-fPIC", "-Wno-unused-value", "-I/opt/rocm/include"]


def _flags():
    # QRGPU_EXTRA_FLAGS=-DQR_TIMELINE: the one diagnostic build (per-tick stamps of the pipelined / overlapped tick on the shared clock: scratch/diag_overlap.py)
    return FLAGS + os.environ.get("QRGPU_EXTRA_FLAGS", "").split()


def _stale():
    if not os.path.exists(SO):
        return True
    stamp = SO + ".flags"                       # a build with other flags (diagnostic -D switches) is another build
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(_flags()):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(HERE, "..", "include", "qrgpu.h"))
    deps.append(os.path.abspath(__file__))            # flags live here
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc cross-compiles without a GPU.  Returns the path of the shared library.  Serialised across processes with a file lock
    (several ranks of one node may import the package at the same moment)."""
    if os.environ.get("QRGPU_LIB"):             # an A/B run names its own build: nothing to compile
        return os.environ["QRGPU_LIB"]
    if not force and not _stale():
        return SO
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not force and not _stale():          # another process built it while we waited
                return SO
            return _build_locked(verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


# what each translation unit includes beyond itself: an object is rebuilt only when one of these (or the flags) is newer than it
KERNEL_DEPS = ["qr_device_types.h", "qr_wave_helpers.h"]
HOST_DEPS = KERNEL_DEPS + ["qrgpu_ctx.h", os.path.join("..", "..", "include", "qrgpu.h")]
EXTRA_DEPS = {"qr_mpc_kernel_fl.hip": ["qr_mpc_kernel.hip"], "qr_wbc_kernel_dbg.hip": ["qr_wbc_kernel.hip"]}


def _obj_stale(src, obj, flags):
    stamp = obj + ".flags"
    if not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + (HOST_DEPS if src.startswith("qrgpu_") else KERNEL_DEPS) + EXTRA_DEPS.get(src, [])
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps) or os.path.getmtime(os.path.abspath(__file__)) > t


def _build_locked(verbose):
    flags = _flags()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if not _obj_stale(src, obj, flags):
            continue
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
        with open(cmd[-1] + ".flags", "w") as f:
            f.write(" ".join(flags))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    with open(SO + ".flags", "w") as f:
        f.write(" ".join(flags))
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
