#!/usr/bin/env python3
"""Stage-1 GCUPS bench (BASELINE.json metric) for the MI355X strip-wavefront engine.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete Stage-1 pass (score + canonical position) over the synthetic pair with both
sequences already resident in HBM.
  N = 1 : BASELINE config C2, 3,000,000 x 3,000,000 unrelated random ACGT, local SW, score-only.
  N > 1 : weak scaling, per-GPU work fixed at 9e12 cells: (3,000,000*N) x 3,000,000, seq1 cut into N
          column bands, boundary column streamed rank g -> g+1 while all strip kernels run (bands.py): through
          pinned zero-copy host columns + gloo by default, RCCL send/recv of device tensors with
          MI355SW_BENCH_COMM=nccl; barrier, best-score all_gather and timing all_reduce go over RCCL.
Rank 0 prints ONE JSON line.  GCUPS convention of the reference: cells = m*n (sw_stage1.cpp:440-448).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue measured on this chip (tools/micro_valu.hip): one wave64 instruction (int32 or packed 2x16)
# per 4 cycles per SIMD, i.e. 0.59 wave-instructions/ns/SIMD at the 2.38 GHz the chip holds under this load
VALU_PEAK_WAVE_INSTR = 256 * 4 * 0.59e9
# HBM-side traffic of one launch from the PMC passes (profiles/r01_pk16_hbm_pmc.json: FETCH_SIZE x2 gfx950
# correction + WRITE_SIZE), keyed by (kernel, m, n, strip_rows); other configurations report null
PMC_TRAFFIC_BYTES = {("pk16", 3000000, 3000000, 1536): 112130775680.0}
# VALU instructions per wave-step (SQ_INSTS_VALU / wave-steps; one step = strip_rows cells).  pk16/1536 is
# measured (profiles/r01_pk16_sq_pmc.json: 128.5); the other packed heights scale its 9.65 per packed row
# pair + 12 per step; int32 figures are from profiles/r01_int32_sq_pmc.json
VALU_PER_STEP = {("int32", 256): 47.0, ("int32", 512): 86.3, ("int32", 1024): 165.0,
                 ("pk16", 256): 31.3, ("pk16", 512): 50.6, ("pk16", 768): 69.9, ("pk16", 1024): 89.2,
                 ("pk16", 1536): 128.5, ("pk16", 2048): 166.4}


def cpu_baseline(pkg, seconds_budget=20.0):
    """MASA-Core's own CPU aligner path (oracle/_ref/ref_driver = reference sources compiled as-is) timed
    on this host, 1 thread, on a bounded sample of the same workload; falls back to the C restatement."""
    oracle = graft.load_oracle()
    side = 60000
    s0, s1 = pkg.seqgen.unrelated_pair(side, side, cfg=1)
    if oracle.have_ref():
        tmp = tempfile.mkdtemp(prefix="bench_ref_")
        try:
            t0 = time.time()
            ref = oracle.run_ref(s0, s1, ["--stage-1", "--no-flush"], workdir=tmp, timeout=600)
            dt = time.time() - t0
            # use the reference's own ALIGN timer when present (excludes FASTA parsing)
            ms = None
            try:
                for ln in open(os.path.join(tmp, "work", "statistics_01.00")):
                    if ln.strip().startswith("ALIGN:"):
                        ms = float(ln.split()[1])
            except OSError:
                pass
            if ms:
                dt = ms / 1000.0
            return {"value": side * side / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "reference",
                    "sample": "%dx%d unrelated SW stage-1, MASA-Core CPUBlockProcessor path (oracle/_ref), best=%s"
                              % (side, side, list(ref["best"]))}
        finally:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    t0 = time.time()
    r = oracle.stage1(s0, s1)
    dt = time.time() - t0
    return {"value": side * side / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
            "sample": "%dx%d unrelated SW stage-1, oracle/sw_oracle.c, best=%s" % (side, side, list(r["best"]))}


def cpu_baseline_mt(pkg):
    """the same recurrence on every host core: the repo's C restatement of CPUBlockProcessor (oracle/sw_oracle.c,
    1024 x 1024 blocks on an anti-diagonal wavefront of threads) -- extra information next to the single-thread
    reference figure, not a replacement for it"""
    oracle = graft.load_oracle()
    cores = min(64, os.cpu_count() or 1)
    side = 120000
    s0, s1 = pkg.seqgen.unrelated_pair(side, side, cfg=1)
    t0 = time.time()
    r = oracle.stage1(s0, s1, threads=cores)
    dt = time.time() - t0
    return {"value": side * side / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": "%dx%d unrelated SW stage-1, oracle/sw_oracle.c on %d threads, best=%s" % (side, side, cores, list(r["best"]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=3000000, help="n (and m per GPU) of the synthetic pair")
    ap.add_argument("--rows-per-lane", type=int, default=int(os.environ.get("MI355SW_R", "0")))
    ap.add_argument("--waves", type=int, default=int(os.environ.get("MI355SW_WAVES", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the engine has no CPU fallback")
    # MI355SW_BENCH_REHEARSAL=1: run the N>1 path on a ONE-GPU box (every rank on cuda:0, gloo instead of RCCL for
    # the collectives, --waves small enough for all ranks' strip kernels to be resident together).  It exercises
    # the band driver, the column transport and the result line; its numbers mean nothing.
    rehearse = os.environ.get("MI355SW_BENCH_REHEARSAL") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = torch.device("cpu") if rehearse else device
    # boundary-column transport between bands: "host" = pinned zero-copy columns + gloo between the
    # rank processes (no GPU queue involved while the persistent kernels run; default), "nccl" = RCCL
    # send/recv of device tensors over xGMI (needs free CU resources next to the strip kernel)
    comm = os.environ.get("MI355SW_BENCH_COMM", "host")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            comm = "host"
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        p2p_group = dist.new_group(backend="gloo") if comm != "nccl" else None

    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits

    n = args.size
    m = args.size * world
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=2)
    waves = args.waves
    if rehearse and waves == 0:
        waves = 1024 // world // 2        # all ranks' strip kernels must be resident on the one GPU together
    if waves == 0 and world > 1 and comm == "nccl":
        # every strip wavefront owns a whole SIMD; RCCL's send/recv kernels need SIMDs of
        # their own, so leave 32 CUs' worth unclaimed (experimental transport -- "host" is the default)
        waves = (256 - 32) * 4
    # N > 1: strip height from the chain model (bands.rows_per_lane_for_bands)
    lim = band_limits(n, [1] * world)
    rows_per_lane = args.rows_per_lane
    if rows_per_lane == 0 and world > 1:
        from masa_cudalign_amd.bands import rows_per_lane_for_bands
        rows_per_lane = rows_per_lane_for_bands(m, lim[1] - lim[0], world, waves or 1024)
    al = pkg.MI355Aligner(device=local_rank, rows_per_lane=rows_per_lane, waves=waves)
    al.setSequences(s0, s1)            # H2D once, outside the timed region
    j0, j1 = lim[rank], lim[rank + 1]

    class _Dist:                       # boundary-column transport: RCCL p2p (device tensors) or gloo (host)
        def __init__(self):
            self.group = p2p_group if world > 1 else None

        def send(self, t, dst):
            dist.send(t, dst=dst, group=self.group)

        def recv(self, t, src):
            dist.recv(t, src=src, group=self.group)

        def all_gather(self, out, t):
            dist.all_gather(out, t)

    # block pruning is left off: C2 is an unrelated pair, on which the reference's (default-on) pruning
    # prunes nothing either, and the engine's kernel without the skip path is the faster one
    runner = BandRunner(al, dist=_Dist() if world > 1 else None, rank=rank, world=world,
                        device=(device if (world > 1 and comm == "nccl") else None), segment_rows=1 << 15)
    if world > 1 and comm != "nccl":
        runner.reduce_best = lambda b, _r=runner: _reduce_cpu(dist, b, world, coll_device)

    def one_step():
        best = runner.run(m, j0, j1)
        if world > 1:
            best = runner.reduce_best(best)
        return best, al.getStatistics()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    best = None
    fence()                            # all bands start together: a band's kernel waits a bounded time for its left neighbour
    for _ in range(args.warmup):
        best, _st = one_step()
    fence()
    t0 = time.time()
    kernel_ms = []
    for _ in range(args.steps):
        best, st = one_step()
        kernel_ms.append(st["kernel_ms"])
    fence()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        cells = float(m) * float(n)
        gcups = cells * args.steps / dt / 1e9
        k_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = st["algorithmic_bytes"]
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        band_cells = float(m) * float(j1 - j0)
        out = {
            "metric": "GCUPS (DP cells/sec) Stage-1", "value": gcups, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i16x2 (packed, exact; int32 fallback)" if st["profile_kernel"] == 2 else "int32",
            "data": "synthetic",
            "config": {"workload": ("C2: %dx%d unrelated random ACGT, local SW, score-only" % (m, n)) if world == 1 else
                       ("weak scaling of C2: (%d*%d)x%d, %d column bands of %d columns, boundary column streamed rank g -> g+1"
                        % (args.size, world, n, world, n // world)),
                       "m": m, "n": n, "bands": world, "strip_rows": st["strip_rows"], "waves_per_gpu": st["waves"],
                       "kernel": {2: "pk16", 1: "int32-profile", 0: "int32-generic"}[st["profile_kernel"]],
                       "comm": comm if world > 1 else "none"},
            "best": {"i": best[0] + 1, "j": best[1] + 1, "score": best[2]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (PMC_TRAFFIC_BYTES.get(("pk16" if st["profile_kernel"] == 2 else "int32", m, n,
                                                            st["strip_rows"])) if world == 1 else None),
                         "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r01_pk16_hbm_pmc.json)",
                         "kernel": "sw_strip_kernel_pk16" if st["profile_kernel"] == 2 else "sw_strip_kernel",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "scan kernel: 17 B per column per %d-row strip; the binding unit is VALU issue" % st["strip_rows"]},
            "valu_roofline": _valu(st, band_cells, k_ms),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg)
            out["cpu_baseline_all_cores"] = cpu_baseline_mt(pkg)
        print(json.dumps(out), flush=True)
    al.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _valu(st, band_cells, k_ms):
    kind = "pk16" if st["profile_kernel"] == 2 else "int32"
    per_step = VALU_PER_STEP.get((kind, st["strip_rows"]), 0.0832 * st["strip_rows"])
    achieved = band_cells / st["strip_rows"] * per_step / (k_ms * 1e-3)
    return {"valu_instr_per_step": per_step, "cells_per_step": st["strip_rows"], "achieved_wave_instr_per_s": achieved,
            "peak_wave_instr_per_s": VALU_PEAK_WAVE_INSTR, "frac": achieved / VALU_PEAK_WAVE_INSTR}


def _reduce_cpu(dist, best, world, device):
    import torch
    from masa_cudalign_amd.bands import canonical_best
    t = torch.tensor(list(best), dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return canonical_best([tuple(int(x) for x in o.tolist()) for o in out])


if __name__ == "__main__":
    main()
