"""Tick logs: the on-disk record of what the MPC+WBC path read and produced, tick by tick.
Same bytes as include/qrgpu_ticklog.h (header-only C for the reference side); format description there.

    w = TickLogWriter("run.qrtl", n_robots, horizon, mpc_cfg20, model15, "a1");  w.append(batch, force, tau, status);  w.close()
    r = TickLogReader("run.qrtl");  r.ticks, r.n_robots, r.horizon;  t = r.tick(k)  ->  dict of [n_robots, width] arrays
"""
import numpy as np

MAGIC = b"QRTICK01"
HEADER_BYTES = 256
FIELDS = ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel", "force", "tau", "status")
MODEL_DEFAULT_GAINS = (100.0, 10.0, 100.0, 10.0, 500.0, 10.0, 0.1, 1.0, 0.4)     # qrgpu_model_desc_default: kp/kd body pos, ori, foot; weights; mu


def field_widths(horizon):
    return dict(mpc_state=28, traj=12 * horizon, gait=4 * horizon, fb_state=37, wbc_cmd=67, prev_ori_vel=3, force=12, tau=12, status=1)


def words_per_robot(horizon):
    return sum(field_widths(horizon).values())          # 160 + 16 h


def model15(model6, gains=MODEL_DEFAULT_GAINS):
    """qrgpu_model_desc as 15 floats from the 6 YAML-dependent ones (workload.model_desc) and the controller gains."""
    return np.concatenate([np.asarray(model6, np.float32)[:6], np.asarray(gains, np.float32)])


def _header(n_robots, horizon, ticks, mpc_cfg, model, robot):
    h = bytearray(HEADER_BYTES)
    h[0:8] = MAGIC
    h[8:24] = np.array([HEADER_BYTES, n_robots, horizon, ticks], "<u4").tobytes()
    h[24:104] = np.asarray(mpc_cfg, "<f4").reshape(20).tobytes()
    h[104:164] = np.asarray(model, "<f4").reshape(15).tobytes()
    name = robot.encode()[:15]
    h[164:164 + len(name)] = name
    return bytes(h)


class TickLogWriter:
    def __init__(self, path, n_robots, horizon, mpc_cfg, model, robot=""):
        if n_robots <= 0 or horizon <= 0:
            raise ValueError("n_robots and horizon must be positive")
        self.n_robots, self.horizon, self.ticks = int(n_robots), int(horizon), 0
        self._cfg, self._model, self._robot = np.asarray(mpc_cfg, np.float32), np.asarray(model, np.float32), robot
        if self._cfg.size != 20 or self._model.size != 15:
            raise ValueError("mpc_cfg must have 20 floats and model 15 (ticklog.model15)")
        self._f = open(path, "wb")
        self._f.write(_header(self.n_robots, self.horizon, 0, self._cfg, self._model, robot))

    def append(self, inputs, force, tau, status):
        """inputs: dict with mpc_state, traj, gait, fb_state, wbc_cmd, prev_ori_vel as [n_robots, width] arrays (workload.make_batch keys)."""
        w = field_widths(self.horizon)
        src = dict(inputs, force=force, tau=tau)
        for k in FIELDS[:-1]:
            a = np.ascontiguousarray(src[k], "<f4")
            if a.shape != (self.n_robots, w[k]):
                raise ValueError("%s: expected shape %s, got %s" % (k, (self.n_robots, w[k]), a.shape))
            self._f.write(a.tobytes())
        s = np.ascontiguousarray(status, "<i4").reshape(self.n_robots, 1)
        self._f.write(s.tobytes())
        self.ticks += 1

    def close(self):
        if self._f:
            self._f.seek(0)
            self._f.write(_header(self.n_robots, self.horizon, self.ticks, self._cfg, self._model, self._robot))
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class TickLogReader:
    def __init__(self, path):
        with open(path, "rb") as f:
            h = f.read(HEADER_BYTES)
        if len(h) != HEADER_BYTES or h[:8] != MAGIC:
            raise ValueError("%s: not a tick log" % path)
        hb, n, hz, ticks = (int(x) for x in np.frombuffer(h, "<u4", 4, 8))
        if hb != HEADER_BYTES or n == 0 or hz == 0:
            raise ValueError("%s: bad header" % path)
        self.n_robots, self.horizon = n, hz
        self.mpc_cfg = np.frombuffer(h, "<f4", 20, 24).copy()
        self.model = np.frombuffer(h, "<f4", 15, 104).copy()
        self.robot = h[164:180].split(b"\0")[0].decode()
        self._w = field_widths(hz)
        self._rec_words = n * words_per_robot(hz)
        self._data = np.memmap(path, "<u4", mode="r", offset=HEADER_BYTES)
        whole = self._data.size // self._rec_words
        self.ticks = whole if ticks == 0 or ticks > whole else ticks        # a log whose writer never closed is sized by its length

    def __len__(self):
        return self.ticks

    def tick(self, k):
        if not 0 <= k < self.ticks:
            raise IndexError(k)
        rec = self._data[k * self._rec_words:(k + 1) * self._rec_words]
        out, off = {}, 0
        for name in FIELDS:
            words = self.n_robots * self._w[name]
            a = np.asarray(rec[off:off + words]).reshape(self.n_robots, self._w[name])
            out[name] = a.view("<i4").reshape(self.n_robots).copy() if name == "status" else a.view("<f4").copy()
            off += words
        out["n"], out["horizon"] = self.n_robots, self.horizon
        return out
