"""GPU (-m gpu): two ranks (gloo rendezvous, both on cuda:0 -- the box has one GPU) run two column bands
with the real HIP engine; the boundary column is streamed while both strip kernels are running."""
import os
import sys

import pytest
import torch.multiprocessing as mp

from test_bands_gloo import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, m, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=43)
        lim = band_limits(n, [1] * world)
        al = pkg.MI355Aligner(device=0, rows_per_lane=4)
        al.setSequences(s0, s1)
        runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=512)
        best = runner.run(m, lim[rank], lim[rank + 1])
        gbest = runner.reduce_best(best)
        al.close()
        q.put((rank, tuple(best), tuple(gbest)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_bands_two_processes_one_gpu(pkg, oracle):
    m, n, world = 6000, 7000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=43)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    for rank, best, gbest in res:
        assert gbest == want, (rank, best, gbest, want)
