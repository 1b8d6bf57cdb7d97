"""Host-side mirror of the reference's interfaces on top of libqrgpu.so (ctypes).

Reference interface                                   -> here
  Quadruped::SetupProblem / SolveMPCKernel / GetMPCSolution
      (QI/controllers/mpc/qr_mpc_interface.h:157,200,215) -> MPCInterface.SetupProblem / SolveMPCKernel / GetMPCSolution
  qrWbcLocomotionController<float>::Run
      (QI/controllers/wbc/qr_wbc_locomotion_controller.hpp:59) -> WbcLocomotionController.Run
  batched ticks (no reference equivalent; n independent robots) -> Context.mpc_solve_batch / wbc_run_batch / tick_batch

There is no CPU path in this module: if the HIP library is missing or no gfx950 device is
usable, construction raises (MissingExtension / QrgpuError).  torch is used only by callers
for device memory, streams and torch.distributed; this module takes raw device pointers
(ints) or anything with a .data_ptr() method.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERRORS = {0: "OK", 1: "NO_DEVICE", 2: "BAD_ARG", 3: "NOT_SETUP", 4: "LAUNCH", 5: "ALLOC", 6: "COMM"}
ST_FLAG_MASK, ST_BAD_TYPE = 0xff0000ff, 0x01000000

ST_MPC_MAXITER, ST_MPC_INFEAS, ST_MPC_OVERFLOW, ST_MPC_NOTSPD = 0x1, 0x2, 0x4, 0x8
ST_WBC_MAXITER, ST_WBC_INFEAS = 0x10, 0x20
FB_DEBUG_FLOATS = 324 + 18 + 18 + 216 + 12 + 12 + 12


def status_flags(status):
    """Flag bits of a status word / array (include/qrgpu.h: bits 0-7 and 24-31; 0 = converged)."""
    return np.asarray(status).astype(np.int64) & ST_FLAG_MASK


def status_iterations(status):
    """MPC active-set iteration count of a status word / array (bits 8-23)."""
    return (np.asarray(status).astype(np.int64) >> 8) & 0xffff


class QrgpuError(RuntimeError):
    pass


class MissingExtension(QrgpuError):
    """libqrgpu.so has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""


class model_desc_struct(C.Structure):
    _fields_ = [("hip_l", C.c_float), ("upper_l", C.c_float), ("lower_l", C.c_float), ("body_size", C.c_float * 3),
                ("kp_body_pos", C.c_float), ("kd_body_pos", C.c_float), ("kp_body_ori", C.c_float), ("kd_body_ori", C.c_float),
                ("kp_foot", C.c_float), ("kd_foot", C.c_float), ("weight_fb", C.c_float), ("weight_fr", C.c_float), ("mu", C.c_float)]


class vmc_desc_struct(C.Structure):
    _fields_ = [("mass", C.c_float), ("inertia", C.c_float * 9), ("acc_weight", C.c_float * 6), ("reg_weight", C.c_float),
                ("friction", C.c_float), ("fmin_ratio", C.c_float), ("fmax_ratio", C.c_float),
                ("hip_l", C.c_float), ("upper_l", C.c_float), ("lower_l", C.c_float)]


class estimator_desc_struct(C.Structure):
    _fields_ = [("hip_l", C.c_float), ("upper_l", C.c_float), ("lower_l", C.c_float), ("hip_offset", C.c_float * 12),
                ("time_step", C.c_float), ("accelerometer_variance", C.c_float), ("sensor_variance", C.c_float), ("window", C.c_int),
                ("body_height", C.c_float)]


class foothold_desc_struct(C.Structure):
    _fields_ = [("hip_offset", C.c_float * 12), ("default_hip_position", C.c_float * 12), ("hip_l", C.c_float), ("swing_kp", C.c_float * 3),
                ("foot_clearance", C.c_float)]


class gait_desc_struct(C.Structure):
    _fields_ = [("stance_duration", C.c_float * 4), ("duty_factor", C.c_float * 4), ("initial_leg_phase", C.c_float * 4),
                ("initial_leg_state", C.c_int * 4), ("contact_detection_phase_threshold", C.c_float), ("wait_time", C.c_float),
                ("advanced_trot", C.c_int)]


class walk_gait_desc_struct(C.Structure):
    _fields_ = [("stance_duration", C.c_float * 4), ("duty_factor", C.c_float * 4), ("initial_leg_phase", C.c_float * 4),
                ("initial_leg_state", C.c_int * 4), ("contact_detection_phase_threshold", C.c_float), ("n_states", C.c_int),
                ("state_switch", C.c_int * 4), ("state_ratio", C.c_float * 4)]


class swing_velocity_desc_struct(C.Structure):
    _fields_ = [("hip_position_com", C.c_float * 12), ("stance_duration", C.c_float * 4), ("swing_kp", C.c_float * 3), ("desired_height", C.c_float)]


EPILOGUE_HIP_COMP, EPILOGUE_CLIP = 1, 2
COMM_ID_BYTES = 128


def comm_unique_id():
    """A fresh ncclUniqueId blob (rank 0 makes it; the launcher hands it to the other ranks)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load_library().qrgpu_comm_unique_id(buf)
    if rc != 0:
        raise QrgpuError("qrgpu_comm_unique_id failed: %s" % ERRORS.get(rc, rc))
    return buf.raw


def lib_path():
    """The in-tree library; QRGPU_LIB names another build of it for A/B runs on one box (scratch/ab_bench.sh)."""
    return os.environ.get("QRGPU_LIB") or os.path.join(_HERE, "libqrgpu.so")


EXPORTS = ["qrgpu_model_desc_default", "qrgpu_create", "qrgpu_destroy", "qrgpu_set_stream", "qrgpu_get_stream", "qrgpu_last_error",
           "qrgpu_device_info", "qrgpu_mpc_setup", "qrgpu_wbc_setup", "qrgpu_mpc_solve_batch", "qrgpu_wbc_run_batch",
           "qrgpu_tick_batch", "qrgpu_mpc_solve1", "qrgpu_wbc_run1", "qrgpu_mpc_assemble_batch", "qrgpu_fb_debug_batch",
           "qrgpu_sync", "qrgpu_enable_timing", "qrgpu_get_timing", "qrgpu_malloc", "qrgpu_free", "qrgpu_memcpy_h2d",
           "qrgpu_memcpy_d2h", "qrgpu_mpc_frontend_batch", "qrgpu_set_lpt_schedule", "qrgpu_vmc_desc_default", "qrgpu_vmc_setup", "qrgpu_vmc_force_batch", "qrgpu_vmc_force1", "qrgpu_set_rescue_pass", "qrgpu_estimator_desc_default", "qrgpu_estimator_state_doubles",
           "qrgpu_estimator_update_batch", "qrgpu_pack_state_batch", "qrgpu_swing_targets_batch", "qrgpu_swing_velocity_batch", "qrgpu_gait_desc_default", "qrgpu_gait_update_batch",
           "qrgpu_foothold_desc_default", "qrgpu_footholds_batch", "qrgpu_ground_update_batch", "qrgpu_walk_gait_desc_default", "qrgpu_walk_gait_update_batch", "qrgpu_vmc_force_world_batch", "qrgpu_vmc_force_world1",
           "qrgpu_set_torque_epilogue", "qrgpu_comm_unique_id", "qrgpu_comm_init_rank", "qrgpu_comm_info", "qrgpu_comm_destroy",
           "qrgpu_allgather_tau", "qrgpu_allgather_tau_of_tick", "qrgpu_allgather_fence", "qrgpu_allgather_wait", "qrgpu_comm_sync", "qrgpu_set_warm_start", "qrgpu_set_planned_list",
           "qrgpu_enable_flop_count", "qrgpu_mpc_flop_counts", "qrgpu_mpc_set_hessian_mode", "qrgpu_wbc_inspect_batch", "qrgpu_host_alloc", "qrgpu_host_free",
           "qrgpu_memcpy_async", "qrgpu_memset_async", "qrgpu_mark", "qrgpu_mark_elapsed_ms", "qrgpu_set_tick_pipeline", "qrgpu_set_tick_overlap",
           "qrgpu_tick_fence", "qrgpu_tick_overlap_stats"]


def load_library():
    """dlopen libqrgpu.so; raises MissingExtension when it was never built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # a context owns more streams than the HIP runtime's default of four hardware queues; streams that share a queue serialise each other and
    # the overlapped tick is refused (qrgpu_set_tick_overlap).  Read by the runtime at its first call in the process: a caller's own value stands.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    p = lib_path()
    if not os.path.exists(p):
        raise MissingExtension("%s not found: the HIP extension must be built first (no CPU fallback exists)" % p)
    lib = C.CDLL(p)
    vp, ip, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    lib.qrgpu_create.argtypes = [ip, ip, ip, C.POINTER(vp)]
    lib.qrgpu_destroy.argtypes = [vp]; lib.qrgpu_destroy.restype = None
    lib.qrgpu_set_stream.argtypes = [vp, vp]
    lib.qrgpu_get_stream.argtypes = [vp]
    lib.qrgpu_get_stream.restype = vp
    lib.qrgpu_set_lpt_schedule.argtypes = [vp, ip]
    lib.qrgpu_set_rescue_pass.argtypes = [vp, ip]
    lib.qrgpu_set_warm_start.argtypes = [vp, ip]
    lib.qrgpu_set_planned_list.argtypes = [vp, ip, ip]
    lib.qrgpu_enable_flop_count.argtypes = [vp, ip]
    lib.qrgpu_mpc_set_hessian_mode.argtypes = [vp, ip]
    lib.qrgpu_mpc_flop_counts.argtypes = [vp, C.POINTER(C.c_double)]
    lib.qrgpu_last_error.argtypes = [vp]; lib.qrgpu_last_error.restype = C.c_char_p
    lib.qrgpu_device_info.argtypes = [vp, C.c_char_p, ip, C.POINTER(ip)]
    lib.qrgpu_model_desc_default.argtypes = [C.POINTER(model_desc_struct)]; lib.qrgpu_model_desc_default.restype = None
    lib.qrgpu_mpc_setup.argtypes = [vp, ip, C.c_float, ip, C.c_float, C.c_float, C.c_float, fp, fp, C.c_float]
    lib.qrgpu_wbc_setup.argtypes = [vp, ip, C.POINTER(model_desc_struct)]
    lib.qrgpu_mpc_solve_batch.argtypes = [vp, ip] + [vp] * 8
    lib.qrgpu_wbc_run_batch.argtypes = [vp, ip] + [vp] * 7
    lib.qrgpu_wbc_inspect_batch.argtypes = [vp, ip] + [vp] * 7
    lib.qrgpu_tick_batch.argtypes = [vp, ip] + [vp] * 11
    lib.qrgpu_set_tick_overlap.argtypes = [vp, ip]
    lib.qrgpu_tick_fence.argtypes = [vp]
    lib.qrgpu_tick_overlap_stats.argtypes = [vp, C.POINTER(ip), C.POINTER(ip)]
    lib.qrgpu_set_torque_epilogue.argtypes = [vp, ip]
    lib.qrgpu_comm_unique_id.argtypes = [C.c_char_p]
    lib.qrgpu_comm_init_rank.argtypes = [vp, C.c_char_p, ip, ip]
    lib.qrgpu_comm_info.argtypes = [vp, C.POINTER(ip), C.POINTER(ip)]
    lib.qrgpu_comm_destroy.argtypes = [vp]
    lib.qrgpu_allgather_tau.argtypes = [vp, vp, vp, ip, vp, ip]
    lib.qrgpu_allgather_tau_of_tick.argtypes = [vp, vp, vp, ip, vp, ip]
    lib.qrgpu_allgather_fence.argtypes = [vp, ip]
    lib.qrgpu_allgather_wait.argtypes = [vp, ip]
    lib.qrgpu_comm_sync.argtypes = [vp]
    lib.qrgpu_mpc_assemble_batch.argtypes = [vp, ip] + [vp] * 6
    lib.qrgpu_vmc_desc_default.argtypes = [C.POINTER(vmc_desc_struct)]; lib.qrgpu_vmc_desc_default.restype = None
    lib.qrgpu_vmc_setup.argtypes = [vp, ip, C.POINTER(vmc_desc_struct)]
    lib.qrgpu_vmc_force_batch.argtypes = [vp, ip] + [vp] * 6
    lib.qrgpu_vmc_force_world_batch.argtypes = [vp, ip] + [vp] * 7
    lib.qrgpu_estimator_desc_default.argtypes = [C.POINTER(estimator_desc_struct)]; lib.qrgpu_estimator_desc_default.restype = None
    lib.qrgpu_estimator_state_doubles.argtypes = [ip]
    lib.qrgpu_estimator_update_batch.argtypes = [vp, ip, C.POINTER(estimator_desc_struct), vp, vp, vp, vp]
    lib.qrgpu_gait_desc_default.argtypes = [C.POINTER(gait_desc_struct)]; lib.qrgpu_gait_desc_default.restype = None
    lib.qrgpu_gait_update_batch.argtypes = [vp, ip, C.POINTER(gait_desc_struct), C.c_float, ip, ip, vp, vp, vp, vp]
    lib.qrgpu_walk_gait_desc_default.argtypes = [C.POINTER(walk_gait_desc_struct)]; lib.qrgpu_walk_gait_desc_default.restype = None
    lib.qrgpu_walk_gait_update_batch.argtypes = [vp, ip, C.POINTER(walk_gait_desc_struct), C.c_float, ip, ip, vp, vp, vp, vp, vp]
    lib.qrgpu_ground_update_batch.argtypes = [vp, ip, ip, vp, vp, vp, vp]
    lib.qrgpu_swing_velocity_batch.argtypes = [vp, ip, C.POINTER(estimator_desc_struct), C.POINTER(swing_velocity_desc_struct), vp, vp]
    lib.qrgpu_swing_targets_batch.argtypes = [vp, ip, C.POINTER(estimator_desc_struct), vp, vp, vp, vp]
    lib.qrgpu_foothold_desc_default.argtypes = [C.POINTER(foothold_desc_struct)]; lib.qrgpu_foothold_desc_default.restype = None
    lib.qrgpu_footholds_batch.argtypes = [vp, ip, C.POINTER(foothold_desc_struct), vp, vp, vp, vp]
    lib.qrgpu_pack_state_batch.argtypes = [vp, ip, fp, vp, vp, vp, vp, vp]
    lib.qrgpu_vmc_force1.argtypes = [vp, ip, fp, fp, fp, fp, C.POINTER(ip)]
    lib.qrgpu_mpc_frontend_batch.argtypes = [vp, ip, ip, C.c_float, C.c_float] + [vp] * 6
    lib.qrgpu_fb_debug_batch.argtypes = [vp, ip, vp, vp, vp]
    lib.qrgpu_mpc_solve1.argtypes = [vp, ip] + [fp] * 9 + [C.POINTER(C.c_double), fp, C.POINTER(ip)]
    lib.qrgpu_wbc_run1.argtypes = [vp, ip] + [fp] * 6 + [C.POINTER(ip)]
    lib.qrgpu_sync.argtypes = [vp]
    lib.qrgpu_enable_timing.argtypes = [vp, ip]
    lib.qrgpu_get_timing.argtypes = [vp, ip, C.POINTER(C.c_double), C.POINTER(ip)]
    lib.qrgpu_malloc.argtypes = [vp, C.c_ulonglong]; lib.qrgpu_malloc.restype = vp
    lib.qrgpu_free.argtypes = [vp, vp]; lib.qrgpu_free.restype = None
    lib.qrgpu_memcpy_h2d.argtypes = [vp, vp, vp, C.c_ulonglong]
    lib.qrgpu_memcpy_d2h.argtypes = [vp, vp, vp, C.c_ulonglong]
    lib.qrgpu_host_alloc.restype = vp
    lib.qrgpu_host_alloc.argtypes = [vp, C.c_ulonglong]
    lib.qrgpu_host_free.argtypes = [vp, vp]
    lib.qrgpu_memcpy_async.argtypes = [vp, vp, vp, C.c_ulonglong, ip]
    lib.qrgpu_memset_async.argtypes = [vp, vp, ip, C.c_ulonglong]
    lib.qrgpu_mark.argtypes = [vp, ip]
    lib.qrgpu_mark_elapsed_ms.argtypes = [vp, ip, ip, C.POINTER(C.c_double)]
    _LIB = lib
    return lib


def _dp(x):
    """device pointer from an int, None, or an object with data_ptr()."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class DeviceArray:
    """Minimal hipMalloc-backed array for callers that do not use torch (tests, smoke)."""

    def __init__(self, ctx, shape, dtype=np.float32):
        self.ctx, self.shape, self.dtype = ctx, tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctx._lib.qrgpu_malloc(ctx._h, max(self.nbytes, 8))
        if not self.ptr:
            raise QrgpuError("hipMalloc of %d bytes failed" % self.nbytes)

    def data_ptr(self):
        return self.ptr

    def upload(self, host):
        host = np.ascontiguousarray(host, self.dtype)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        self.ctx._chk(self.ctx._lib.qrgpu_memcpy_h2d(self.ctx._h, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def download(self):
        out = np.empty(self.shape, self.dtype)
        self.ctx._chk(self.ctx._lib.qrgpu_memcpy_d2h(self.ctx._h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def zero(self):
        """Asynchronous byte fill with 0 on the context stream."""
        self.ctx._chk(self.ctx._lib.qrgpu_memset_async(self.ctx._h, self.ptr, 0, self.nbytes))
        return self

    def copy_from_pinned(self, pinned):
        """Asynchronous host -> device copy from a PinnedArray of the same size."""
        assert pinned.nbytes == self.nbytes
        self.ctx._chk(self.ctx._lib.qrgpu_memcpy_async(self.ctx._h, self.ptr, pinned.ptr, self.nbytes, 0))
        return self

    def copy_to_pinned(self, pinned):
        assert pinned.nbytes == self.nbytes
        self.ctx._chk(self.ctx._lib.qrgpu_memcpy_async(self.ctx._h, pinned.ptr, self.ptr, self.nbytes, 1))
        return pinned

    def row(self, r0, r1=None):
        """Device pointer (int) of rows [r0, r1) of a 2-d array, e.g. the joint angles inside fb_state."""
        return self.ptr + r0 * int(np.prod(self.shape[1:])) * self.dtype.itemsize

    def free(self):
        if self.ptr:
            self.ctx._lib.qrgpu_free(self.ctx._h, self.ptr)
            self.ptr = None


class PinnedArray:
    """hipHostMalloc-backed numpy view (qrgpu_host_alloc) for asynchronous copies."""

    def __init__(self, ctx, shape, dtype=np.float32):
        self.ctx, self.shape, self.dtype = ctx, tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctx._lib.qrgpu_host_alloc(ctx._h, max(self.nbytes, 8))
        if not self.ptr:
            raise QrgpuError("hipHostMalloc of %d bytes failed" % self.nbytes)
        self.array = np.ctypeslib.as_array((C.c_char * self.nbytes).from_address(self.ptr)).view(self.dtype).reshape(self.shape)

    def free(self):
        if self.ptr:
            self.array = None
            self.ctx._lib.qrgpu_host_free(self.ctx._h, self.ptr)
            self.ptr = None


class Context:
    """One qrgpu_ctx: a device, a stream, per-type MPC/WBC parameters."""

    def __init__(self, device_id=0, max_batch=1024, horizon_max=16):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.qrgpu_create(device_id, max_batch, horizon_max, C.byref(h))
        if rc != 0:
            raise QrgpuError("qrgpu_create failed: %s (a gfx950 device is required; there is no CPU fallback)" % ERRORS.get(rc, rc))
        self._h = h
        self.max_batch, self.horizon = max_batch, None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.qrgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise QrgpuError("%s: %s" % (ERRORS.get(rc, rc), self._lib.qrgpu_last_error(self._h).decode()))

    def last_error(self):
        return self._lib.qrgpu_last_error(self._h).decode()

    # -- setup ---------------------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._chk(self._lib.qrgpu_set_stream(self._h, C.c_void_p(int(stream_ptr) if stream_ptr else 0)))

    def get_stream(self):
        """The hipStream_t (as an integer) the context queues on: a non-blocking stream of its own unless set_stream named another."""
        return int(self._lib.qrgpu_get_stream(self._h) or 0)

    def device_info(self):
        buf = C.create_string_buffer(256); lds = C.c_int(0)
        cus = self._lib.qrgpu_device_info(self._h, buf, 256, C.byref(lds))
        return dict(name=buf.value.decode(), cus=cus, lds_per_cu=lds.value)

    def mpc_setup(self, type_id, dt, horizon, mu, fmax, mass, inertia, weights, alpha):
        inertia = np.ascontiguousarray(inertia, np.float32); weights = np.ascontiguousarray(weights, np.float32)
        self._chk(self._lib.qrgpu_mpc_setup(self._h, type_id, dt, horizon, mu, fmax, mass, _fp(inertia), _fp(weights), alpha))
        self.horizon = horizon

    def mpc_setup_packed(self, type_id, cfg20, horizon):
        """cfg20 = dt, mu, fmax, mass, inertia[3], weights[12], alpha (workload.mpc_cfg)."""
        c = np.asarray(cfg20, np.float32)
        self.mpc_setup(type_id, float(c[0]), horizon, float(c[1]), float(c[2]), float(c[3]), c[4:7], c[7:19], float(c[19]))

    def wbc_setup(self, type_id, hip_l=None, upper_l=None, lower_l=None, body_size=None, **gains):
        d = model_desc_struct()
        self._lib.qrgpu_model_desc_default(C.byref(d))
        if hip_l is not None: d.hip_l = hip_l
        if upper_l is not None: d.upper_l = upper_l
        if lower_l is not None: d.lower_l = lower_l
        if body_size is not None:
            for i in range(3): d.body_size[i] = body_size[i]
        for k, v in gains.items():
            setattr(d, k, v)
        self._chk(self._lib.qrgpu_wbc_setup(self._h, type_id, C.byref(d)))

    def wbc_setup_packed(self, type_id, model6):
        m = np.asarray(model6, np.float32)
        self.wbc_setup(type_id, float(m[0]), float(m[1]), float(m[2]), [float(x) for x in m[3:6]])

    def alloc(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def alloc_pinned(self, shape, dtype=np.float32):
        return PinnedArray(self, shape, dtype)

    def mark(self, index):
        """Record timing mark `index` on the context stream."""
        self._chk(self._lib.qrgpu_mark(self._h, int(index)))

    def mark_elapsed_ms(self, a, b):
        ms = C.c_double(0)
        self._chk(self._lib.qrgpu_mark_elapsed_ms(self._h, int(a), int(b), C.byref(ms)))
        return ms.value

    # -- batched device-pointer API ---------------------------------------------------------------
    def mpc_solve_batch(self, n, mpc_state, traj, gait, q, force, tau=None, status=None, type_id=None):
        self._chk(self._lib.qrgpu_mpc_solve_batch(self._h, n, _dp(type_id), _dp(mpc_state), _dp(traj), _dp(gait), _dp(q),
                                                  _dp(force), _dp(tau), _dp(status)))

    def wbc_run_batch(self, n, fb_state, wbc_cmd, prev_ori, tau, qdes=None, status=None, type_id=None):
        self._chk(self._lib.qrgpu_wbc_run_batch(self._h, n, _dp(type_id), _dp(fb_state), _dp(wbc_cmd), _dp(prev_ori), _dp(tau),
                                                _dp(qdes), _dp(status)))

    def wbc_inspect_batch(self, n, fb_state, wbc_cmd, prev_ori, tau, qp, status=None, type_id=None):
        """wbc_run_batch plus the relaxation QP's solution: qp [n][30] = qpz[18], optimalFr[12] (instrumented kernel)."""
        self._chk(self._lib.qrgpu_wbc_inspect_batch(self._h, n, _dp(type_id), _dp(fb_state), _dp(wbc_cmd), _dp(prev_ori), _dp(tau),
                                                    _dp(qp), _dp(status)))

    def tick_batch(self, n, mpc_state, traj, gait, fb_state, wbc_cmd, prev_ori, force, tau, status=None, type_id=None, qdes=None):
        """qdes [24][n]: desiredJPos / desiredJVel of the kinematic projection (K12); None skips that projection."""
        self._chk(self._lib.qrgpu_tick_batch(self._h, n, _dp(type_id), _dp(mpc_state), _dp(traj), _dp(gait), _dp(fb_state),
                                             _dp(wbc_cmd), _dp(prev_ori), _dp(force), _dp(tau), _dp(qdes), _dp(status)))

    def set_tick_pipeline(self, on=True):
        """WBC launch of a tick beside its MPC launches (default) or behind them."""
        self._chk(self._lib.qrgpu_set_tick_pipeline(self._h, 1 if on else 0))

    def set_tick_overlap(self, on=True, strict=True):
        """Consecutive pipelined ticks (h <= 11) that write different output arrays overlap: tick t + 1's solves start in tick t's drain
        (include/qrgpu.h: the caller's promise about inputs).  -> True when the mode is on.  The library refuses it when two of the context's
        streams share a hardware queue (GPU_MAX_HW_QUEUES): strict raises, otherwise False is returned and ticks stay as they were."""
        rc = self._lib.qrgpu_set_tick_overlap(self._h, 1 if on else 0)
        if rc == 3 and not strict:             # QRGPU_ERR_NOT_SETUP
            return False
        self._chk(rc)
        return bool(on)

    def tick_overlap_stats(self):
        """-> (ticks chained to their predecessor, overlapped-form ticks that waited for the context's stream instead)"""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self._lib.qrgpu_tick_overlap_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def tick_fence(self):
        """The next overlapped tick waits for everything queued on the context's stream so far (inputs produced there since the last tick)."""
        self._chk(self._lib.qrgpu_tick_fence(self._h))

    def set_torque_epilogue(self, hip_comp=False, clip=False):
        """K14 tail on the batched torques: +-0.9 N m abad compensation (qr_fsm_state_locomotion.cpp:141-151), +-23 N m clip (qr_safety_checker.cpp:48-66)."""
        self._chk(self._lib.qrgpu_set_torque_epilogue(self._h, (EPILOGUE_HIP_COMP if hip_comp else 0) | (EPILOGUE_CLIP if clip else 0)))

    def vmc_setup_packed(self, type_id, cfg20, geom3):
        """cfg20 = workload.vmc_cfg(): mass, inertia[9], acc_weight[6], reg_weight, friction, fmin_ratio, fmax_ratio; geom3 = hip/upper/lower length."""
        d = vmc_desc_struct()
        cfg20 = np.asarray(cfg20, np.float32)
        d.mass = float(cfg20[0])
        for i in range(9): d.inertia[i] = float(cfg20[1 + i])
        for i in range(6): d.acc_weight[i] = float(cfg20[10 + i])
        d.reg_weight, d.friction, d.fmin_ratio, d.fmax_ratio = (float(v) for v in cfg20[16:20])
        d.hip_l, d.upper_l, d.lower_l = (float(v) for v in geom3[:3])
        self._chk(self._lib.qrgpu_vmc_setup(self._h, type_id, C.byref(d)))

    def vmc_force_batch(self, n, vmc_in, q, force, tau=None, status=None, type_id=None):
        """ComputeContactForce (+ MapContactForceToJointTorques) of n robots: qr_qp_torque_optimizer.cpp:190-301."""
        self._chk(self._lib.qrgpu_vmc_force_batch(self._h, n, _dp(type_id), _dp(vmc_in), _dp(q), _dp(force), _dp(tau), _dp(status)))

    def vmc_force_world_batch(self, n, vmc_in, ratio, q, force, tau=None, status=None, type_id=None):
        """World-frame overload (qr_qp_torque_optimizer.cpp:304-398): vmc_in carries Rcb = rotMat, gvec = (0,0,9.8), normal = e_z; ratio [8][n]."""
        self._chk(self._lib.qrgpu_vmc_force_world_batch(self._h, n, _dp(type_id), _dp(vmc_in), _dp(ratio), _dp(q), _dp(force), _dp(tau), _dp(status)))

    def vmc_force1(self, vmc_in, q=None, type_id=0):
        a = np.ascontiguousarray(vmc_in, np.float32); qa = np.ascontiguousarray(q, np.float32) if q is not None else None
        f = np.zeros(12, np.float32); tau = np.zeros(12, np.float32); st = C.c_int(0)
        self._chk(self._lib.qrgpu_vmc_force1(self._h, type_id, _fp(a), _fp(qa) if qa is not None else None, _fp(f),
                                             _fp(tau) if qa is not None else None, C.byref(st)))
        return f, (tau if qa is not None else None), st.value

    def estimator_state_doubles(self, window):
        return self._lib.qrgpu_estimator_state_doubles(int(window))

    def estimator_update_batch(self, n, cfg20, est_in, tick, est_state, est_out):
        """UpdateDataFlow kinematics + qrRobotVelocityEstimator::Update of n robots.  cfg20 = workload.estimator_cfg()."""
        d = estimator_desc_struct()
        cfg20 = np.asarray(cfg20, np.float32)
        d.hip_l, d.upper_l, d.lower_l, d.time_step, d.accelerometer_variance, d.sensor_variance = (float(v) for v in cfg20[:6])
        d.window = int(cfg20[6]); d.body_height = float(cfg20[19])
        for i in range(12): d.hip_offset[i] = float(cfg20[7 + i])
        self._chk(self._lib.qrgpu_estimator_update_batch(self._h, n, C.byref(d), _dp(est_in), _dp(tick), _dp(est_state), _dp(est_out)))

    def gait_update_batch(self, n, cfg19, current_time, contact, gait_state, gait_out=None, fe_in=None, stop=False, reset=False):
        """qrOpenLoopGaitGenerator::Update of n robots (qr_openloop_gait_generator.cpp:126-249).  cfg19 = workload.gait_cfg()."""
        d = gait_desc_struct()
        cfg19 = np.asarray(cfg19, np.float32)
        for l in range(4):
            d.stance_duration[l] = float(cfg19[l]); d.duty_factor[l] = float(cfg19[4 + l]); d.initial_leg_phase[l] = float(cfg19[8 + l])
            d.initial_leg_state[l] = int(cfg19[12 + l])
        d.contact_detection_phase_threshold = float(cfg19[16]); d.wait_time = float(cfg19[17]); d.advanced_trot = int(cfg19[18])
        self._chk(self._lib.qrgpu_gait_update_batch(self._h, n, C.byref(d), float(current_time), int(bool(stop)), int(bool(reset)), _dp(contact),
                                                    _dp(gait_state), _dp(gait_out), _dp(fe_in)))

    def walk_gait_update_batch(self, n, cfg26, current_time, contact, walk_state, walk_out=None, ratio=None, vmc_in=None, stop=False, reset=0):
        """qrWalkGaitGenerator::Update of n robots + the walk branch of UpdateFRatio (qr_walk_gait_generator.cpp:202-288,
        qr_torque_stance_leg_controller.cpp:125-168).  cfg26 = workload.walk_cfg(); reset: 2 = as constructed, 1 = Reset(), 0 = carry on."""
        d = walk_gait_desc_struct()
        v = np.asarray(cfg26, np.float32)
        for l in range(4):
            d.stance_duration[l] = float(v[l]); d.duty_factor[l] = float(v[4 + l]); d.initial_leg_phase[l] = float(v[8 + l]); d.initial_leg_state[l] = int(v[12 + l])
            d.state_switch[l] = int(v[18 + l]); d.state_ratio[l] = float(v[22 + l])
        d.contact_detection_phase_threshold = float(v[16]); d.n_states = int(v[17])
        self._chk(self._lib.qrgpu_walk_gait_update_batch(self._h, n, C.byref(d), float(current_time), int(bool(stop)), int(reset), _dp(contact),
                                                         _dp(walk_state), _dp(walk_out), _dp(ratio), _dp(vmc_in)))

    def ground_update_batch(self, n, ground_in, ground_state, ground_out=None, est_in=None, reset=False):
        """qrGroundSurfaceEstimator::Update of n robots (qr_ground_surface_estimator.cpp:40-70,151-206): ground_in [23][n],
        ground_state [13][n] doubles (memory), ground_out [32][n]; est_in: rows 45-53 receive groundRMat."""
        self._chk(self._lib.qrgpu_ground_update_batch(self._h, n, int(bool(reset)), _dp(ground_in), _dp(ground_state), _dp(ground_out), _dp(est_in)))

    def footholds_batch(self, n, desc29, fh_in, swing_in, gait_state=None, gait_out=None):
        """Swing-leg selection + foothold heuristic (qr_swing_leg_controller.cpp:211-236, qr_foothold_planner.cpp:110-239).
        desc29 = workload.foothold_cfg(): hip_offset[12], default_hip_position[12], hip_l, swing_kp[3], foot_clearance."""
        d = foothold_desc_struct()
        self._lib.qrgpu_foothold_desc_default(C.byref(d))
        v = np.asarray(desc29, np.float32)
        for i in range(12):
            d.hip_offset[i] = float(v[i]); d.default_hip_position[i] = float(v[12 + i])
        d.hip_l = float(v[24]); d.foot_clearance = float(v[28])
        for i in range(3):
            d.swing_kp[i] = float(v[25 + i])
        self._chk(self._lib.qrgpu_footholds_batch(self._h, n, C.byref(d), _dp(fh_in), _dp(gait_state), _dp(gait_out), _dp(swing_in)))

    def swing_velocity_batch(self, n, cfg20, vdesc20, swing_vel_in, out):
        """Swing-leg action of the velocity mode (qr_swing_leg_controller.cpp:285-309, 408-424).  cfg20 = workload.estimator_cfg() (geometry
        part), vdesc20 = workload.swing_velocity_cfg(): hip position + com offset[12], stanceDuration[4], swingKp[3], desiredHeight - clearance."""
        d = estimator_desc_struct()
        cfg20 = np.asarray(cfg20, np.float32)
        d.hip_l, d.upper_l, d.lower_l = (float(v) for v in cfg20[:3])
        for i in range(12): d.hip_offset[i] = float(cfg20[7 + i])
        v = swing_velocity_desc_struct(); vd = np.asarray(vdesc20, np.float32)
        for i in range(12): v.hip_position_com[i] = float(vd[i])
        for i in range(4): v.stance_duration[i] = float(vd[12 + i])
        for i in range(3): v.swing_kp[i] = float(vd[16 + i])
        v.desired_height = float(vd[19])
        self._chk(self._lib.qrgpu_swing_velocity_batch(self._h, n, C.byref(d), C.byref(v), _dp(swing_vel_in), _dp(out)))

    def swing_targets_batch(self, n, cfg20, swing_in, wbc_cmd=None, foot_target_world=None, qdes=None):
        """Swing-leg targets (qr_swing_leg_controller.cpp:362-424, ADVANCED_TROT).  cfg20 = workload.estimator_cfg() (geometry part)."""
        d = estimator_desc_struct()
        cfg20 = np.asarray(cfg20, np.float32)
        d.hip_l, d.upper_l, d.lower_l = (float(v) for v in cfg20[:3])
        for i in range(12): d.hip_offset[i] = float(cfg20[7 + i])
        self._chk(self._lib.qrgpu_swing_targets_batch(self._h, n, C.byref(d), _dp(swing_in), _dp(wbc_cmd), _dp(foot_target_world), _dp(qdes)))

    def pack_state_batch(self, n, com_offset, est_in, est_out, rpy, mpc_state=None, fb_state=None):
        """mpc_state[28] / fb_state[37] from the estimator's inputs and outputs (SolveDenseMPC :385-399, UpdateModel :136-156)."""
        co = np.ascontiguousarray(com_offset, np.float32)
        self._chk(self._lib.qrgpu_pack_state_batch(self._h, n, _fp(co), _dp(est_in), _dp(est_out), _dp(rpy), _dp(mpc_state), _dp(fb_state)))

    def mpc_frontend_batch(self, n, fe_in, fe_state, traj, gait, wbc_cmd=None, mpc_updated=None, num_horizon_l=2, dt_ctrl=0.002, dt_mpc=0.06):
        """SetupCommand + Run + UpdateMPC (without the solve) of n robots: qr_mpc_stance_leg_controller.cpp:158-382."""
        self._chk(self._lib.qrgpu_mpc_frontend_batch(self._h, n, int(num_horizon_l), float(dt_ctrl), float(dt_mpc), _dp(fe_in), _dp(fe_state),
                                                     _dp(traj), _dp(gait), _dp(wbc_cmd), _dp(mpc_updated)))

    def mpc_assemble_batch(self, n, mpc_state, traj, gait, H, g, type_id=None):
        self._chk(self._lib.qrgpu_mpc_assemble_batch(self._h, n, _dp(type_id), _dp(mpc_state), _dp(traj), _dp(gait), _dp(H), _dp(g)))

    def fb_debug_batch(self, n, fb_state, out, type_id=None):
        self._chk(self._lib.qrgpu_fb_debug_batch(self._h, n, _dp(type_id), _dp(fb_state), _dp(out)))

    def set_lpt_schedule(self, on=True):
        """Longest-first workgroup dispatch from the previous call's per-robot solve time (speed only)."""
        self._chk(self._lib.qrgpu_set_lpt_schedule(self._h, 1 if on else 0))

    def set_warm_start(self, on=True):
        """Start each robot slot's active set from its previous solve (speed only; results agree with a cold start to solver tolerance)."""
        self._chk(self._lib.qrgpu_set_warm_start(self._h, 1 if on else 0))

    def set_hessian_mode(self, mode="f32"):
        """K4 arithmetic: "f32" (exact, default) or "bf16x3" (three-limb bf16 on the bf16 matrix cores: BASELINE.json configs[4])."""
        self._chk(self._lib.qrgpu_mpc_set_hessian_mode(self._h, {"f32": 0, "bf16x3": 1}[mode]))

    def set_planned_list(self, on=True, big_nls=0):
        """Robots that needed the rescue pass last call are solved beside the main launch this call (scheduling only)."""
        self._chk(self._lib.qrgpu_set_planned_list(self._h, 1 if on else 0, int(big_nls)))

    def set_rescue_pass(self, on=True):
        self._chk(self._lib.qrgpu_set_rescue_pass(self._h, 1 if on else 0))

    def sync(self):
        self._chk(self._lib.qrgpu_sync(self._h))

    def enable_flop_count(self, on=True):
        self._chk(self._lib.qrgpu_enable_flop_count(self._h, 1 if on else 0))

    def mpc_flop_counts(self):
        """Executed arithmetic of the last counted MPC call, summed over its robots: dict(fp32_vector, fp32_matrix, fp64_sweep, fp64_active_set)."""
        out = (C.c_double * 4)()
        self._chk(self._lib.qrgpu_mpc_flop_counts(self._h, out))
        return dict(fp32_vector=out[0], fp32_matrix=out[1], fp64_sweep=out[2], fp64_active_set=out[3])

    # -- multi-GPU: the all-gather of torques over RCCL (qrgpu_comm.hip) ---------------------------------------------
    def comm_init_rank(self, id_blob, nranks, rank):
        assert len(id_blob) == COMM_ID_BYTES
        self._chk(self._lib.qrgpu_comm_init_rank(self._h, bytes(id_blob), int(nranks), int(rank)))

    def comm_destroy(self):
        self._chk(self._lib.qrgpu_comm_destroy(self._h))

    def allgather_tau(self, tau, n_local, tau_all, slot=0, nccl_comm=None, of_tick=False):
        """Asynchronous on the context's communication stream; tau_all is [nranks][12][n_local] (rank-major).  of_tick: `tau` is what the
        context's last tick_batch produced and nothing queued since writes it (qrgpu_allgather_tau_of_tick: no event on the compute stream)."""
        fn = self._lib.qrgpu_allgather_tau_of_tick if of_tick else self._lib.qrgpu_allgather_tau
        self._chk(fn(self._h, _dp(nccl_comm), _dp(tau), int(n_local), _dp(tau_all), int(slot)))

    def allgather_fence(self, slot=0):
        self._chk(self._lib.qrgpu_allgather_fence(self._h, int(slot)))

    def allgather_wait(self, slot=0):
        """Make the compute stream wait for the gather of `slot` (for a consumer of tau_all queued there); the slot's fence stays due."""
        self._chk(self._lib.qrgpu_allgather_wait(self._h, int(slot)))

    def comm_sync(self):
        self._chk(self._lib.qrgpu_comm_sync(self._h))

    def enable_timing(self, on=True):
        """False: off; True: HIP events around every kernel launch; an int N > 1: around every N-th launch of a kernel; -1: pause."""
        self._chk(self._lib.qrgpu_enable_timing(self._h, int(on) if on is not True else 1))

    def get_timing(self, kernel):
        ms = C.c_double(0); cnt = C.c_int(0)
        self._chk(self._lib.qrgpu_get_timing(self._h, kernel, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    # -- single-robot host API ------------------------------------------------------------------------
    def mpc_solve1(self, p, v, quat, w, r, rpy, traj, gait, q=None, type_id=0):
        a = [np.ascontiguousarray(x, np.float32) for x in (p, v, quat, w, r, rpy, traj, gait)]
        qa = np.ascontiguousarray(q, np.float32) if q is not None else None
        f = np.zeros(12, np.float64); tau = np.zeros(12, np.float32); st = C.c_int(0)
        self._chk(self._lib.qrgpu_mpc_solve1(self._h, type_id, *[_fp(x) for x in a], _fp(qa) if qa is not None else None,
                                             f.ctypes.data_as(C.POINTER(C.c_double)), _fp(tau), C.byref(st)))
        return f, (tau if q is not None else None), st.value

    def wbc_run1(self, fb_state, wbc_cmd, prev_ori_vel, type_id=0, want_qdes=True):
        s = np.ascontiguousarray(fb_state, np.float32); c = np.ascontiguousarray(wbc_cmd, np.float32)
        prev = np.ascontiguousarray(prev_ori_vel, np.float32).copy()
        tau = np.zeros(12, np.float32); qd = np.zeros(12, np.float32); qdd = np.zeros(12, np.float32); st = C.c_int(0)
        self._chk(self._lib.qrgpu_wbc_run1(self._h, type_id, _fp(s), _fp(c), _fp(prev), _fp(tau),
                                           _fp(qd) if want_qdes else None, _fp(qdd) if want_qdes else None, C.byref(st)))
        return tau, qd, qdd, prev, st.value


class MPCInterface:
    """Drop-in shaped like the free functions of qr_mpc_interface.h (one robot, host pointers).

    The reference keeps its problem in file-static state (qr_mpc_interface.cpp:35-104); here the
    state lives in this object."""

    def __init__(self, ctx=None, type_id=0):
        self.ctx = ctx or Context(max_batch=1)
        self.type_id = type_id
        self._soln = None
        self.status = 0

    def SetupProblem(self, dt, horizon, frictionCoeff, fMax, totalMass, inertia, weight, alpha):
        self.ctx.mpc_setup(self.type_id, float(dt), int(horizon), float(frictionCoeff), float(fMax), float(totalMass),
                           inertia, weight, float(alpha))
        self.horizon = int(horizon)

    def SolveMPCKernel(self, p, v, q, w, r, rpy, state_trajectory, gait):
        """r: 3x4 (column = leg) as Eigen::Matrix<float,3,4>; q: quaternion (w,x,y,z)."""
        r = np.asarray(r, np.float32)
        r_colmajor = r.T.reshape(12) if r.shape == (3, 4) else r.reshape(12)
        f, _, st = self.ctx.mpc_solve1(p, v, q, w, r_colmajor, rpy, np.asarray(state_trajectory, np.float32)[:12 * self.horizon],
                                       np.asarray(gait, np.float32)[:4 * self.horizon], None, self.type_id)
        self._soln, self.status = f, st

    def GetMPCSolution(self, index):
        """First-step forces only (indices 0..11), as every call site of the reference uses it
        (qr_mpc_stance_leg_controller.cpp:404); 0 before the first solve (qr_mpc_interface.cpp:448)."""
        if self._soln is None:
            return 0.0
        return float(self._soln[index])


class WbcLocomotionController:
    """Shaped like qrWbcLocomotionController<float> (QI/controllers/wbc/qr_wbc_locomotion_controller.hpp:47-59).

    Run() keeps the reference's cadence: it computes on every 2nd call (iteration % 2 == 0,
    qr_wbc_locomotion_controller.cpp:111,133) and otherwise re-applies the last torques to the
    stance legs (UpdateLegCMD runs every call, :129)."""

    def __init__(self, ctx, type_id=0):
        self.ctx, self.type_id = ctx, type_id
        self.iteration = 0
        self.prev_ori_vel = np.zeros(3, np.float32)
        self.jointTorqueCmd = np.zeros(12, np.float32)
        self.desiredJPos = np.zeros(12, np.float32)
        self.desiredJVel = np.zeros(12, np.float32)
        self.status = 0

    def Run(self, fb_state, wbc_cmd, leg_cmd_tua):
        """leg_cmd_tua: array of 12 torques, stance-leg entries are overwritten in place."""
        if self.iteration % 2 == 0:
            tau, qd, qdd, prev, st = self.ctx.wbc_run1(fb_state, wbc_cmd, self.prev_ori_vel, self.type_id)
            self.jointTorqueCmd, self.desiredJPos, self.desiredJVel, self.prev_ori_vel, self.status = tau, qd, qdd, prev, st
        contact = np.asarray(wbc_cmd, np.float32)[63:67]
        for leg in range(4):
            if contact[leg] != 0:
                leg_cmd_tua[3 * leg:3 * leg + 3] = self.jointTorqueCmd[3 * leg:3 * leg + 3]
        self.iteration += 1
        return leg_cmd_tua
