"""Chain of column bands in separate processes on ONE GPU, related pair, block pruning against the shared best:
    python tools/chain_prune_probe.py M N WORLD [R] [transport] [waves]
prints per band: kernel ms, pruned fraction, best -- with and without the shared best (MI355SW_NO_SHARED_BEST)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, m, n, R, transport, waves, q):
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=2)
        lim = band_limits(n, [1] * world)
        out = {}
        for shared in (True, False):
            if shared:
                os.environ.pop("MI355SW_NO_SHARED_BEST", None)
            else:
                os.environ["MI355SW_NO_SHARED_BEST"] = "1"
            al = pkg.MI355Aligner(device=0, rows_per_lane=R, waves=waves)
            al.setSequences(s0, s1)
            runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=1 << 15, transport=transport, prune_blocks=True)
            dist.barrier()
            t0 = time.time()
            best = runner.run(m, lim[rank], lim[rank + 1], n_total=n)
            dt = time.time() - t0
            st = al.getStatistics()
            out[shared] = dict(best=tuple(runner.reduce_best(best)), own=tuple(best), kernel_ms=st["kernel_ms"], wall=dt,
                               pruned=st["pruned_cells"] / max(1, st["cells"]), wait_ms=st["wait_ms"], hints=runner.hints)
            dist.barrier()
            al.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    from test_bands_gloo import _free_port
    m, n, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    R = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    transport = sys.argv[5] if len(sys.argv) > 5 else "p2p"
    waves = int(sys.argv[6]) if len(sys.argv) > 6 else 1024 // world // 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, m, n, R, transport, waves, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for shared in (True, False):
        print("shared best" if shared else "every band on its own")
        for r in range(world):
            o = res[r][shared]
            print("  band %d: kernel %.0f ms wall %.2f s pruned %.3f wait %.1f ms own best %s chain best %s hints %d" % (
                r, o["kernel_ms"], o["wall"], o["pruned"], o["wait_ms"], o["own"], o["best"], o["hints"]))
