"""CPU, world_size 2 (gloo): bench.run_full_config -- BASELINE's multi-GPU configuration run once behind the timed steps of
`bench.py --gpus N` (c4_half / c4_full / c5_full) -- with the oracle-based stand-in engine of test_bands_gloo.py in place of
MI355Aligner: the record rank 0 gets, and that a rank which fails (before the chain starts, or in the middle of it) costs an
error INSIDE the record, within the wait budget, never a hang.  Reference: the reference runs its real job on N devices from one
command line (--fork, M/libmasa/libmasa.cpp:540-642)."""
import os
import sys
import time
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_bands_gloo import OracleStreamEngine, _free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _worker(rank, world, port, m, n, comm, fault, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    real = graft.load_package()
    oracle = graft.load_oracle()
    bench = _bench_module()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MI355SW_BENCH_EXTRA_WAIT_S"] = "4"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p2p_group = dist.new_group(backend="gloo")

    class FakeAligner(OracleStreamEngine):
        """MI355Aligner's constructor and bookkeeping around the stand-in's streaming surface"""

        def __init__(self, device=-1, rows_per_lane=0, waves=0, wait_seconds=0.0, **kw):
            if fault == "setup" and rank == 1:
                raise RuntimeError("no GPU for rank 1 (test)")
            self._wait, self.polls = wait_seconds, 0

        def setSequences(self, s0, s1):
            OracleStreamEngine.__init__(self, oracle, s0, s1, seg=256)
            self._opts["wait_seconds"] = self._wait

        def _segment_done(self, r1):
            # (the stand-in computes every segment it can in one poll: the fault sits between two segments)
            if fault == "run" and rank == 0 and r1 >= 1024:
                raise real.engine.AlignerError("streamPoll: EHIP the device fell off the bus (test)")
            return OracleStreamEngine._segment_done(self, r1)

        def getStatistics(self):
            return {"kernel_ms": 1.0, "wait_ms": 0.0, "pruned_cells": 0, "seed_ms": 0.0, "strip_rows": 256, "kernel": "stand-in",
                    "kernel_launches": 1}

        def close(self):
            if hasattr(self, "in_shm"):
                self.portClose()

    class _Dist:
        def send(self, t, dst):
            dist.send(t, dst=dst, group=p2p_group)

        def recv(self, t, src):
            dist.recv(t, src=src, group=p2p_group)

        def all_gather(self, out, t):
            dist.all_gather(out, t)

        def new_group(self, ranks, backend="gloo"):
            return dist.new_group(ranks, backend=backend)

        def all_reduce(self, t, op=None, group=None):
            dist.all_reduce(t, op=op, group=group)

        ReduceOp = dist.ReduceOp

    pkg = types.SimpleNamespace(seqgen=real.seqgen, MI355Aligner=FakeAligner, engine=real.engine)
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(synchronize=lambda: None))
    fc = dict(key="c4_test", m=m, n=n, related=False, nw=False, cfg=5, est_s=1, expect=None,
              workload="a small C4: %dx%d unrelated, local SW, %d column bands" % (m, n, world))
    args = types.SimpleNamespace(rows_per_lane=4)
    t0 = time.time()
    try:
        out = bench.run_full_config(fc, pkg, fake_torch, dist, _Dist, p2p_group, world, rank, rank, comm, True, 0, torch.device("cpu"), args)
        q.put((rank, out, time.time() - t0))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(m, n, comm, fault):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, n, comm, fault, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (o, dt)) for r, o, dt in [q.get(timeout=200) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(300)
@pytest.mark.parametrize("comm", ["host", "p2p"])
def test_the_extra_runs_the_chain_and_checks_it(pkg, oracle, comm):
    m, n = 2600, 2400
    res = _run(m, n, comm, None)
    out, _ = res[0]
    assert res[1][0] is None                                 # only rank 0 gets the record
    assert "error" not in out, out
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=5)
    ref = oracle.stage1(s0, s1)
    assert (out["best"]["i"], out["best"]["j"], out["best"]["score"]) == tuple(ref["best"])
    assert out["check"]["equals_one_band_over_all_columns"] and out["check"]["oracle_window_600x600_ending_at_the_cell"] and out["check"]["ok"]
    assert out["bands"] == 2 and out["comm"] == comm and len(out["ranks"]) == 2 and out["value"] > 0
    assert [r["band_columns"] for r in out["ranks"]] == [[0, 1200], [1200, 2400]]


@pytest.mark.timeout(300)
def test_a_rank_without_its_engine_is_reported_not_waited_for(pkg):
    res = _run(2600, 2400, "host", "setup")
    out, dt = res[0]
    assert "error" in out and "rank 1" in out["error"] and "no GPU for rank 1" in out["error"]
    assert "value" not in out and dt < 60


@pytest.mark.timeout(300)
def test_a_band_that_dies_in_mid_run_costs_the_wait_budget_not_a_hang(pkg):
    """band 0 fails after its first segments: band 1 waits for boundary rows that never come, gives up when the wait budget
    (MI355SW_BENCH_EXTRA_WAIT_S, 4 s here) is spent, and rank 0's record names both"""
    res = _run(2600, 2400, "p2p", "run")
    out, dt = res[0]
    assert "error" in out and "rank 0" in out["error"] and "fell off the bus" in out["error"]
    assert "value" not in out and dt < 90
