"""CPU: bench.py's own launcher.  `python bench.py --gpus N` without WORLD_SIZE must start N rank processes (the parent never touches a GPU),
print rank 0's single JSON line, and fail -- never fall back to one GPU -- when a rank fails.  QRGPU_BENCH_DRY makes the ranks stop after the
rendezvous (a localhost socket: quadruped-robot_amd/rendezvous.py) and one max-reduce, so this runs without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, dry, extra_env=None):
    env = dict(os.environ, QRGPU_BENCH_DRY=dry)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_self_launch_two_ranks():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1"], "1")
    assert r.returncode == 0, r.stderr.decode()
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1                                  # ONE JSON line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] == 2.0       # the max-reduce over both ranks went through


def test_self_launch_fails_when_a_rank_fails():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1"], "fail1")
    assert r.returncode != 0
    assert not any('"n_gpus": 1' in l for l in r.stdout.decode().splitlines())     # never continues as one GPU
    assert "rank process failed" in r.stderr.decode()


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4"], "1", dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode == 2 and b"WORLD_SIZE" in r.stderr


def test_single_process_dry_run_does_not_import_torch():
    r = _run(["--gpus", "1"], "1", dict(PYTHONVERBOSE=""))
    assert r.returncode == 0 and json.loads(r.stdout.decode().strip())["n_gpus"] == 1
    code = "import sys, runpy; sys.argv = ['bench.py', '--gpus', '1']; runpy.run_path(%r, run_name='__main__'); assert 'torch' not in sys.modules" % BENCH
    env = dict(os.environ, QRGPU_BENCH_DRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()


def test_no_rank_imports_torch_at_n_2():
    """The launcher's plumbing at N > 1 is a localhost socket: no rank imports torch (each rank checks its own sys.modules behind the rendezvous)."""
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1"], "1", dict(QRGPU_BENCH_DRY_ASSERT_NO_TORCH="1"))
    assert r.returncode == 0, r.stderr.decode()
    assert json.loads(r.stdout.decode().strip())["value"] == 2.0


def test_launched_by_torch_distributed_run():
    """The driver's own launch line: torch.distributed.run keeps MASTER_PORT for its agent's store, so the ranks must meet elsewhere (an ephemeral
    port published in a file named after MASTER_ADDR / MASTER_PORT)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, QRGPU_BENCH_DRY="1", QRGPU_BENCH_DRY_ASSERT_NO_TORCH="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        BENCH, "--gpus", "2", "--steps", "4", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["value"] == 2.0 and json.loads(lines[0])["n_gpus"] == 2


def test_launcher_deadline_kills_ranks_that_hang():
    """All ranks hung (a collective that never completes): the launcher's overall deadline kills the children and the run fails."""
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1"], "hang", dict(QRGPU_BENCH_DEADLINE_S="3"))
    assert r.returncode == 124
    assert "deadline" in r.stderr.decode()


def test_the_json_line_is_alone_on_stdout_even_when_a_library_writes_there():
    """RCCL prints a five-line banner on the C-level standard output when its first communicator comes up -- in front of the one JSON line the
    driver parses.  bench.py points file descriptor 1 at standard error and writes its line to the real standard output; the dry run imitates
    the banner (QRGPU_BENCH_DRY_NOISE) on every rank."""
    for args in (["--gpus", "1"], ["--gpus", "2", "--steps", "4", "--warmup", "1"]):
        r = _run(args, "1", dict(QRGPU_BENCH_DRY_NOISE="1"))
        assert r.returncode == 0, r.stderr.decode()
        lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1 and json.loads(lines[0])["dry"] is True, lines
        assert b"banner on file descriptor 1" in r.stderr
