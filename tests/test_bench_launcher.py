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


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_check_joins_the_real_runs_backend_when_the_devices_are_there():
    """--launch-check initialises "nccl" (= RCCL, the call bench.main makes) as soon as there is one device per rank and stays
    on gloo otherwise (this container: no GPU; a one-GPU box under --gpus 2)."""
    b = _bench_module()
    assert b.launch_check_backend(8, 8) == "nccl" and b.launch_check_backend(4, 8) == "nccl"
    assert b.launch_check_backend(2, 1) == "gloo" and b.launch_check_backend(2, 0) == "gloo"
    assert b.launch_check_backend(1, 8) == "gloo"                   # one rank: no group at all
    assert b.launch_check_backend(2, 0, forced="nccl") == "nccl" and b.launch_check_backend(8, 8, forced="gloo") == "gloo"
    p = _run(["--gpus", "2", "--launch-check"])
    rec = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert rec["backend"] == "gloo" and rec["devices_visible"] == 0 and rec["sum_of_ranks"] == 3


def test_the_gpus_n_line_runs_baselines_own_multi_gpu_configuration():
    """--gpus 4 -> BASELINE config 4 (59 M x 64 M unrelated SW, 4 bands), --gpus 8 -> config 5 (249 M x 228 M related NW with
    pruning, 8 bands), --gpus 2 -> C4's first two bands; other N none; a rehearsal runs them at 1/16 of the linear size."""
    b = _bench_module()
    c4, c5, c2 = b.full_config_for(4), b.full_config_for(8), b.full_config_for(2)
    assert (c4["key"], c4["m"], c4["n"], c4["related"], c4["nw"]) == ("c4_full", 59000000, 64000000, False, False)
    assert (c5["key"], c5["m"], c5["n"], c5["related"], c5["nw"]) == ("c5_full", 249000000, 228000000, True, True)
    assert (c2["key"], c2["m"], c2["n"]) == ("c4_half", 59000000, 32000000)
    assert c4["expect"]["score"] == 26 and c5["expect"]["score"] == 134862766
    assert "59000000x64000000" in c4["workload"] and "249000000x228000000" in c5["workload"]
    assert b.full_config_for(1) is None and b.full_config_for(3) is None and b.full_config_for(16) is None
    r = b.full_config_for(8, rehearse=True)
    assert (r["m"], r["n"], r["expect"]) == (249000000 // 16, 228000000 // 16, None)
    # the recorded values are the ones under profiles/
    rec4 = json.load(open(os.path.join(ROOT, "profiles", "r02_scale_c4_chain_59Mx64M_4bands.json")))
    assert (rec4["best"]["i"], rec4["best"]["j"], rec4["best"]["score"]) == (c4["expect"]["i"], c4["expect"]["j"], c4["expect"]["score"])
    rec5 = json.load(open(os.path.join(ROOT, "profiles", "r05_nw_c5_249Mx228M_one_gpu_2048rows.json")))
    assert rec5["h_last_cell"] == c5["expect"]["score"]
