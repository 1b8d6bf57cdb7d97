"""CPU: the tick-log format: Python writer/reader round trip, the header-only C writer/reader of
include/qrgpu_ticklog.h against the Python side, and the committed golden log against the CPU oracle."""
import os
import subprocess
import sys

import numpy as np

import gpu_helpers as G
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ticklog_a1_h10.qrtl")


def _rand_tick(rng, n, h, widths):
    t = {k: rng.standard_normal((n, w)).astype(np.float32) for k, w in widths.items() if k != "status"}
    t["status"] = rng.integers(0, 1 << 20, n).astype(np.int32)
    return t


def test_roundtrip_and_unclosed_log(pkg, tmp_path):
    tl = pkg.ticklog
    n, h = 3, 5
    rng = np.random.default_rng(1)
    w = tl.field_widths(h)
    assert tl.words_per_robot(h) == 160 + 16 * h
    ticks = [_rand_tick(rng, n, h, w) for _ in range(4)]
    p = str(tmp_path / "a.qrtl")
    cfg, model = pkg.mpc_cfg("lite3"), tl.model15(pkg.model_desc("lite3"))
    wr = tl.TickLogWriter(p, n, h, cfg, model, "lite3")
    for t in ticks:
        wr.append(t, t["force"], t["tau"], t["status"])
    # before close(): header says 0 ticks, the reader sizes the log from the file
    wr._f.flush()
    r0 = tl.TickLogReader(p)
    assert r0.ticks == 4
    wr.close()
    r = tl.TickLogReader(p)
    assert (r.ticks, r.n_robots, r.horizon, r.robot) == (4, n, h, "lite3")
    assert np.array_equal(r.mpc_cfg, cfg) and np.array_equal(r.model, model)
    assert os.path.getsize(p) == 256 + 4 * 4 * n * (160 + 16 * h)
    for k, t in enumerate(ticks):
        got = r.tick(k)
        for name in tl.FIELDS:
            assert np.array_equal(got[name], t[name]), name
    with pytest.raises(IndexError):
        r.tick(4)
    with pytest.raises(ValueError):
        wr2 = tl.TickLogWriter(str(tmp_path / "b.qrtl"), n, h, cfg, model)
        wr2.append(dict(ticks[0], traj=ticks[0]["traj"][:, :-1]), ticks[0]["force"], ticks[0]["tau"], ticks[0]["status"])
    open(str(tmp_path / "c.qrtl"), "wb").write(b"not a log")
    with pytest.raises(ValueError):
        tl.TickLogReader(str(tmp_path / "c.qrtl"))


C_PROG = r"""
#include "qrgpu_ticklog.h"
int main(int argc, char **argv)
{
    /* argv[1]: log to copy, argv[2]: copy written through the C API */
    qrtl_file in, out;
    if (qrtl_open(&in, argv[1]) != 0) return 2;
    if (qrtl_create(&out, argv[2], in.n_robots, in.horizon, in.mpc_cfg, in.model, in.robot) != 0) return 3;
    void *buf[QRTL_NFIELDS];
    for (int k = 0; k < QRTL_NFIELDS; ++k) buf[k] = malloc((size_t)4 * in.n_robots * qrtl_field_width(in.horizon, k));
    for (uint32_t t = 0; t < in.ticks; ++t) {
        for (int k = 0; k < QRTL_NFIELDS; ++k) if (qrtl_read(&in, t, k, buf[k]) != 0) return 4;
        if (qrtl_append(&out, (float *)buf[0], (float *)buf[1], (float *)buf[2], (float *)buf[3], (float *)buf[4], (float *)buf[5],
                        (float *)buf[6], (float *)buf[7], (int32_t *)buf[8]) != 0) return 5;
    }
    printf("%u %u %u %s\n", in.ticks, in.n_robots, in.horizon, in.robot);
    if (qrtl_close(&out) != 0) return 6;
    qrtl_close(&in);
    return 0;
}
"""


def test_c_header_reads_and_writes_the_same_bytes(pkg, tmp_path):
    src = tmp_path / "copy.c"; src.write_text(C_PROG)
    exe = str(tmp_path / "copy")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    out = str(tmp_path / "copy.qrtl")
    txt = subprocess.check_output([exe, GOLDEN, out]).decode().split()
    assert txt == ["12", "2", "10", "a1"]
    assert open(out, "rb").read() == open(GOLDEN, "rb").read()
    # and as C++ (the reference side is C++)
    subprocess.check_call(["g++", "-O1", "-Wall", "-std=c++17", "-x", "c++", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe + "pp"])
    subprocess.check_call([exe + "pp", GOLDEN, out + "2"])
    assert open(out + "2", "rb").read() == open(GOLDEN, "rb").read()


def test_golden_log_is_what_the_oracle_computes(pkg, oracle):
    """The committed log was written by tests/golden/make_ticklog.py; the oracle must still reproduce it bit for bit (stateful sequence)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_ticklog
    r = pkg.ticklog.TickLogReader(GOLDEN)
    seq = make_ticklog.oracle_sequence(pkg, r.n_robots, r.ticks, r.horizon, r.robot, seed=0x71C, excite=0.8)
    assert np.array_equal(r.mpc_cfg, pkg.mpc_cfg("a1")) and np.array_equal(r.model[:6], pkg.model_desc("a1"))
    moved = 0.0
    for k, (b, f, tau, st) in enumerate(seq):
        t = r.tick(k)
        for name in ("mpc_state", "traj", "gait", "fb_state", "wbc_cmd", "prev_ori_vel"):
            assert np.array_equal(t[name], b[name]), (k, name)
        assert np.array_equal(t["status"], st) and np.all(G.flags(st) == 0)
        assert np.array_equal(t["force"], f) and np.array_equal(t["tau"], tau), k
        moved = max(moved, float(np.abs(t["prev_ori_vel"]).max()))
    assert moved > 0          # the WBC memory is really carried from tick to tick
