"""CPU: the torch-free rendezvous of the multi-GPU launcher (quadruped-robot_amd/rendezvous.py) with real rank processes: barrier, max-reduce,
broadcast of the 128-byte communicator id, all-gather of a torque block; a stale port file of an earlier run; a rank that never arrives."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RDV = os.path.join(ROOT, "quadruped-robot_amd", "rendezvous.py")

_RANK = r"""
import importlib.util, os, sys
import numpy as np
spec = importlib.util.spec_from_file_location("rdv", %(rdv)r); rdv = importlib.util.module_from_spec(spec); spec.loader.exec_module(rdv)
g = rdv.Group(timeout=float(os.environ.get("RDV_TIMEOUT", "60")))
assert "torch" not in sys.modules
g.barrier()
assert g.allreduce_max(10.0 * (g.rank + 1)) == 10.0 * g.world
blob = rdv.exchange_comm_id(g, lambda: bytes(range(128)))
assert blob == bytes(range(128))
tau = np.full((12, 5), float(g.rank), np.float32)
parts = g.allgather_bytes(tau.tobytes())
allt = np.stack([np.frombuffer(p, np.float32).reshape(12, 5) for p in parts])
assert allt.shape == (g.world, 12, 5) and all((allt[r] == r).all() for r in range(g.world))
for k in range(50):
    assert g.allreduce_max(float(k + g.rank)) == float(k + g.world - 1)
g.barrier(); g.close()
print("RANK_OK", g.rank)
"""


def _env(rank, world, port):
    return dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_group_collectives(world):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c", _RANK % dict(rdv=RDV)], env=_env(r, world, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()[-2000:]
        assert ("RANK_OK %d" % r) in out.decode()


def test_a_stale_port_file_of_an_earlier_run_is_survived():
    """Rank 0 of an earlier run with the same MASTER_PORT died before it could remove its file: the file names a port nobody listens on.  The
    late rank keeps polling; rank 0 of THIS run replaces the file."""
    port = _free_port()
    stale = os.path.join(tempfile.gettempdir(), "qrgpu_rdv_127.0.0.1_%d.port" % port)
    with open(stale, "w") as f:
        f.write("%d dead-nonce\n" % _free_port())
    p1 = subprocess.Popen([sys.executable, "-c", _RANK % dict(rdv=RDV)], env=_env(1, 2, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    import time
    time.sleep(1.0)                                        # rank 1 is already chewing on the stale file when rank 0 comes up
    p0 = subprocess.Popen([sys.executable, "-c", _RANK % dict(rdv=RDV)], env=_env(0, 2, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for p in (p0, p1):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()[-2000:]
    assert not os.path.exists(stale)


def test_a_rank_that_never_arrives_fails_the_others_at_the_deadline():
    port = _free_port()
    p0 = subprocess.Popen([sys.executable, "-c", _RANK % dict(rdv=RDV)], env=dict(_env(0, 2, port), RDV_TIMEOUT="2"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out, err = p0.communicate(timeout=60)
    assert p0.returncode != 0 and b"1 of 2 ranks arrived" in err
