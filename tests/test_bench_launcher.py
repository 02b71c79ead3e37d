"""`python3 bench.py --gpus N` starts its own ranks (bench.self_launch): the branch the driver's multi-GPU run enters first.
CPU tests -- `--launch-check` makes the ranks meet over gloo instead of touching a GPU.  Reference: the reference forks its
nodes from one command line before any device call (M/libmasa/libmasa.cpp:540-642)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=180):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout, cwd="/tmp")


def test_self_launch_starts_the_ranks_and_relays_rank_0s_line():
    p = _run(["--gpus", "3", "--launch-check"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # ONE line on stdout, everything else went to stderr
    rec = json.loads(lines[0])
    assert rec["launch_check"] and rec["world"] == 3 and rec["gpus"] == 3 and rec["sum_of_ranks"] == 6
    assert rec["launcher"] == "self" and rec["master"].startswith("127.0.0.1:")
    assert b"torch.distributed.run" in p.stderr and b"--nproc-per-node=3" in p.stderr


def test_self_launch_returns_the_ranks_exit_code():
    # no GPU here: every rank of the real bench stops with the engine's "no CPU fallback" message, and the parent reports failure
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert p.stdout.decode().strip() == ""
    assert b"needs an MI355X" in p.stderr


def test_self_launch_time_out_ends_the_ranks():
    t0 = time.time()
    p = _run(["--gpus", "2", "--launch-check", "--launch-check-sleep", "120"], {"MI355SW_BENCH_TIMEOUT_S": "15"})
    assert p.returncode == 124
    assert time.time() - t0 < 100
    assert b"terminating" in p.stderr
    # nobody of the child's process group is left behind (the rank processes carry the sleep argument on their command line)
    time.sleep(1.0)
    ps = subprocess.run(["ps", "-eo", "args"], stdout=subprocess.PIPE).stdout.decode()
    assert "--launch-check-sleep 120" not in ps


def test_a_rank_environment_is_respected():
    # under an external torchrun (WORLD_SIZE set) bench.py must NOT start ranks of its own
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--launch-check"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, **env), timeout=120, cwd="/tmp")
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    rec = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert rec["world"] == 1 and rec["launcher"] == "external"
    assert b"starting" not in p.stderr
