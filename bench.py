#!/usr/bin/env python3
"""Stage-1 GCUPS bench (BASELINE.json metric) for the MI355X strip-wavefront engine.

    python bench.py --gpus N --steps K --warmup W        (any N: for N > 1 it starts its own ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete Stage-1 pass (score + canonical position) over the synthetic pair with both
sequences already resident in HBM.
  N = 1 : BASELINE config C2, 3,000,000 x 3,000,000 unrelated random ACGT, local SW, score-only.
  N > 1 : weak scaling, per-GPU work fixed at 7.2e13 cells: (24,000,000*N) x 3,000,000 (tall like BASELINE's C4/C5:
          the start-up of a chain of column bands is N-1 band sweeps, whatever the height), seq1 cut into N column
          bands; the boundary column goes GPU to GPU through column ports (bands.py transport "p2p": band g's strip
          kernel stores its last column into band g+1's HBM over xGMI and publishes the row count, band g+1's kernel
          polls it); barrier, best-score all_gather and timing all_reduce go over RCCL, the 80-byte port handles
          over a gloo side group.  Before anything is timed the ranks check the ports: every rank maps its neighbour's
          (bands.probe_p2p), then a 1 Mi-row chain runs once through the ports and once through the host and every
          band compares the column it received and the best cell it found (bands.verify_p2p, a minute's budget); if
          any rank fails either, ALL ranks use the pinned-host + gloo transport and the line says so ("comm",
          "comm_note").  MI355SW_BENCH_COMM=host selects that transport outright.
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
# VALU issue of the instruction class the packed kernel is made of (VOP3P, v_perm, DPP: "slow" class, 4 cycles per
# wave64 instruction per SIMD whatever the occupancy), measured with controlled placement: v_pk_max_i16 at 4
# wavefronts per SIMD = 0.536 wave-instructions/ns/SIMD (profiles/r02_valu_issue_rate.md, r02_micro_valu3_survey.txt)
VALU_PEAK_WAVE_INSTR = 256 * 4 * 0.536e9
# PMC-derived figures (HBM traffic per launch, VALU instructions per wave-step) are only quoted when they were
# measured on THIS build of the kernels: profiles/pmc_index.json is keyed by kernel_build_id() (tools/pmc_collect.py
# writes it from rocprofv3 --pmc passes); anything else is reported as null, never as a stale constant
PMC_INDEX = os.path.join(ROOT, "profiles", "pmc_index.json")


def kernel_build_id():
    """identity of the device code of the LOADED library (mi355sw_build_id: the hash of csrc/ compiled into it).  PMC
    figures are quoted only when library id = id of the sources in the tree = key of profiles/pmc_index.json."""
    graft.load_package()
    from masa_cudalign_amd import engine
    return engine.library_build_id()


def build_identity():
    graft.load_package()
    from masa_cudalign_amd import engine
    lib, src = engine.library_build_id(), engine.source_build_id()
    return {"library": lib, "sources": src, "stale_library": lib != src}


def pmc_lookup(kernel, m, n, strip_rows):
    try:
        idx = json.load(open(PMC_INDEX))
    except (OSError, ValueError):
        return None
    ident = build_identity()
    if ident["stale_library"]:          # the .so was not built from the sources next to it: nothing measured applies
        return None
    return idx.get(ident["library"], {}).get("%s:%dx%d:%d" % (kernel, m, n, strip_rows))


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


def self_launch(argv, gpus):
    """`python3 bench.py --gpus N` with N > 1 and no rank environment: start the ranks ourselves.

    The reference forks its nodes BEFORE any device call from one command line (M/libmasa/libmasa.cpp:540-642, device
    selection per node X/cuda_util.cpp:191-257); here the parent -- which has imported nothing that touches the GPU --
    starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a CHILD process
    (never an exec), relays rank 0's JSON line to its own stdout and everything else to stderr, and returns the child's
    exit code.  MI355SW_BENCH_TIMEOUT_S (default 3300) bounds the child: on a time-out its whole process group is
    terminated and the exit code is 124."""
    import signal
    import socket
    import threading
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:       # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env["MI355SW_BENCH_LAUNCHER"] = "self"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the column ports between the rank processes need it
    env.setdefault("OMP_NUM_THREADS", "1")
    timeout = float(os.environ.get("MI355SW_BENCH_TIMEOUT_S", "3300"))
    sys.stderr.write("[bench] starting %d ranks: %s\n" % (gpus, " ".join(cmd)))
    sys.stderr.flush()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, start_new_session=True)
    lines = []

    def relay():
        for raw in child.stdout:
            text = raw.decode(errors="replace")
            is_line = False
            if text.lstrip().startswith("{"):
                try:
                    is_line = "metric" in json.loads(text) or "launch_check" in json.loads(text)
                except ValueError:
                    pass
            if is_line:
                lines.append(text)
                sys.stdout.write(text)
                sys.stdout.flush()
            else:
                sys.stderr.write(text)
                sys.stderr.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    try:
        rc = child.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        sys.stderr.write("[bench] the ranks did not finish within %.0f s: terminating them\n" % timeout)
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
            try:
                os.killpg(child.pid, sig)           # the child leads its own process group (start_new_session)
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        th.join(5.0)
        return 124
    th.join(30.0)
    if rc == 0 and not lines:
        sys.stderr.write("[bench] the ranks ended without a result line\n")
        return 1
    return rc


def launch_check_backend(world, devices_visible, forced=None):
    """which torch.distributed backend --launch-check joins: the real run's "nccl" (RCCL) as soon as there is one device per
    rank, gloo otherwise; MI355SW_LAUNCH_CHECK_BACKEND forces one"""
    if forced in ("nccl", "gloo"):
        return forced
    return "nccl" if (world > 1 and devices_visible >= world) else "gloo"


# BASELINE.json's multi-GPU configurations as one-shot extras of the `--gpus N` line (like c3_full at N = 1): run ONCE after the
# timed weak-scaling steps, on all ranks, through the same band driver and transport the headline settled on.
#   N = 4: C4 -- 59 M x 64 M unrelated, local SW, 4 column bands
#   N = 8: C5 -- 249 M x 228 M related, global NW, block pruning on (the chain's bound starts from the seed of the whole matrix)
#   N = 2: the first two bands of C4 as a matrix of their own (59 M x 32 M)
# `expect`: what a one-GPU run of the same pair recorded (profiles/), 1-based DP cell.  MI355SW_BENCH_REHEARSAL=1 runs them at
# 1/16 of the linear size (a one-GPU box; nothing recorded to compare with: the chain must equal one band over all columns).
FULL_CONFIGS = {
    2: dict(key="c4_half", m=59000000, n=32000000, related=False, nw=False, cfg=5, est_s=200,
            workload="BASELINE config 4's first two bands as a matrix of their own: %dx%d unrelated random ACGT, local SW, 2 column bands",
            expect=None),
    4: dict(key="c4_full", m=59000000, n=64000000, related=False, nw=False, cfg=5, est_s=260,
            workload="BASELINE config 4 at full size: %dx%d unrelated random ACGT, local SW, score + canonical position, 4 column bands",
            expect={"i": 30491425, "j": 3008203, "score": 26, "source": "profiles/r02_scale_c4_chain_59Mx64M_4bands.json (4 bands, one GPU)"}),
    8: dict(key="c5_full", m=249000000, n=228000000, related=True, nw=True, cfg=5, est_s=600,
            workload="BASELINE config 5 at full size: %dx%d related synthetic pair, global NW, gap-initialised borders, block pruning on, 8 column bands",
            expect={"i": 249000000, "j": 228000000, "score": 134862766, "source": "profiles/r05_nw_c5_249Mx228M_one_gpu_2048rows.json (one GPU, 1550 s)"}),
}


def full_config_for(world, rehearse=False):
    """the BASELINE configuration `--gpus world` runs once behind its timed steps, or None (N = 1: c3_full; other N: none)"""
    c = FULL_CONFIGS.get(world)
    if c is None:
        return None
    c = dict(c)
    if rehearse:
        c["m"], c["n"] = c["m"] // 16, c["n"] // 16
        c["expect"] = None
        c["est_s"] = 30
    c["workload"] = c["workload"] % (c["m"], c["n"])
    return c


def launch_check(args):
    """--launch-check: what the launcher needs to work, without a GPU -- every rank joins a gloo group over the rendezvous it
    was handed, the ranks add up their numbers, rank 0 prints one line.  (CPU test of the self-launch branch.)"""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # The collectives backend of the real run: "nccl" (= RCCL) when every rank has a device of its own, gloo otherwise (no GPU
    # here, or fewer devices than ranks: a one-GPU box).  torch.cuda.device_count() does not initialise the GPU.
    backend = launch_check_backend(world, torch.cuda.device_count(), os.environ.get("MI355SW_LAUNCH_CHECK_BACKEND"))
    device = torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    # (a forced "nccl" with ONE rank still makes the group: RCCL initialised and one all_reduce through it on a one-GPU box)
    grouped = world > 1 or os.environ.get("MI355SW_LAUNCH_CHECK_BACKEND") == "nccl"
    if grouped:
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)     # the call main() makes
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1], dtype=torch.int64, device=device)
    if grouped:
        dist.all_reduce(t)
    if args.launch_check_sleep > 0:
        time.sleep(args.launch_check_sleep)
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "gpus": args.gpus, "sum_of_ranks": int(t.item()),
                          "backend": backend if grouped else "none", "devices_visible": torch.cuda.device_count(),
                          "launcher": os.environ.get("MI355SW_BENCH_LAUNCHER", "external"),
                          "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return 0


T_BENCH_START = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=3000000, help="n (and m per GPU) of the synthetic pair")
    ap.add_argument("--rows-per-lane", type=int, default=int(os.environ.get("MI355SW_R", "0")))
    ap.add_argument("--waves", type=int, default=int(os.environ.get("MI355SW_WAVES", "0")))
    ap.add_argument("--reserve-cus", type=int, default=int(os.environ.get("MI355SW_RESERVE_CUS", "0")),
                    help="leave this many compute units of every GPU without a strip wavefront (waves = 4 * (CUs - K)): room for "
                         "somebody else's kernels -- an RCCL send/recv beside the persistent strip kernel, which otherwise owns every "
                         "SIMD; measured cost: profiles/r04_reserved_cus.json")
    ap.add_argument("--tall", type=int, default=8, help="N > 1: rows per GPU = tall * size (weak scaling)")
    ap.add_argument("--related", action="store_true",
                    help="a RELATED synthetic pair (2 %% substitutions, indels, one inversion) with block pruning on in every "
                         "band against the chain-wide best score: GCUPS in the reference's m*n convention plus the pruned fraction")
    ap.add_argument("--nw", action="store_true",
                    help="GLOBAL alignment (Needleman-Wunsch, gap-initialised borders: BASELINE config 5's recurrence) instead of a "
                         "local one; with --related, block pruning against a running lower bound of H[m][n] shared along the chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shapes", action="store_true", help="N = 1: skip the C3- and C5-shaped checks (c3_shape, c5_shape)")
    ap.add_argument("--no-target-shape", action="store_true", help="N = 1: skip the 228 M-row north-star-height step")
    ap.add_argument("--no-c3-full", action="store_true", help="N = 1: skip BASELINE config 3's stage 1 at full size (c3_full, about two minutes)")
    ap.add_argument("--no-single-reference", action="store_true",
                    help="N > 1: skip rank 0's untimed run of ONE GPU's share of the cells (tall*size x size) alone")
    ap.add_argument("--no-full-config", action="store_true",
                    help="N = 2, 4, 8: skip BASELINE's multi-GPU configuration run once behind the timed steps (c4_half / c4_full / c5_full)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only check the launcher: the ranks meet over gloo, rank 0 prints one line (runs without a GPU)")
    ap.add_argument("--launch-check-sleep", type=float, default=0.0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    # N > 1 from a plain `python3 bench.py --gpus N`: the ranks are started here, as a child process, before anything that
    # initialises the GPU has been imported (self_launch)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(sys.argv[1:], args.gpus)
    if args.launch_check:
        return launch_check(args)

    import torch
    import torch.distributed as dist

    if os.environ.get("MI355SW_BENCH_STACKS"):      # debugging aid: every rank dumps its Python stacks every N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MI355SW_BENCH_STACKS"]), repeat=True, file=sys.stderr)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: one rank per GPU (python3 bench.py --gpus N starts them itself)" % (args.gpus, world))
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
    # boundary-column transport between bands: "p2p" = column ports, GPU to GPU (default); "host" = pinned
    # zero-copy columns + gloo between the rank processes
    comm = os.environ.get("MI355SW_BENCH_COMM", "p2p")
    if comm not in ("p2p", "host"):
        raise SystemExit("MI355SW_BENCH_COMM must be p2p or host")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        p2p_group = dist.new_group(backend="gloo")     # host-side messages between neighbours (port handles / segments)

    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits

    n = args.size
    m = args.size * (1 if world == 1 else args.tall * world)
    s0, s1 = (pkg.seqgen.related_pair if args.related else pkg.seqgen.unrelated_pair)(m, n, cfg=2)
    waves = args.waves
    if waves == 0 and args.reserve_cus > 0:
        waves = 4 * max(1, torch.cuda.get_device_properties(local_rank).multi_processor_count - args.reserve_cus)
    if rehearse and waves == 0:
        waves = 1024 // world // 2        # all ranks' strip kernels must be resident on the one GPU together
    # N > 1: strip height from the chain model (bands.rows_per_lane_for_bands)
    lim = band_limits(n, [1] * world)
    rows_per_lane = args.rows_per_lane
    if rows_per_lane == 0 and world > 1:
        from masa_cudalign_amd.bands import rows_per_lane_for_bands
        rows_per_lane = rows_per_lane_for_bands(m, lim[1] - lim[0], world, waves or 1024)
    al = pkg.MI355Aligner(device=local_rank, rows_per_lane=rows_per_lane, waves=waves)
    al.setSequences(s0, s1)            # H2D once, outside the timed region
    j0, j1 = lim[rank], lim[rank + 1]

    class _Dist:                       # host-side neighbour messages (gloo side group) + collectives (RCCL)
        def __init__(self):
            self.group = p2p_group if world > 1 else None

        def send(self, t, dst):
            dist.send(t, dst=dst, group=self.group)

        def recv(self, t, src):
            dist.recv(t, src=src, group=self.group)

        def all_gather(self, out, t):
            dist.all_gather(out, t)

        def new_group(self, ranks, backend="gloo"):
            return dist.new_group(ranks, backend=backend)

        def all_reduce(self, t, op=None, group=None):
            dist.all_reduce(t, op=op, group=group)

        ReduceOp = dist.ReduceOp

    # block pruning is left off: C2 is an unrelated pair, on which the reference's (default-on) pruning
    # prunes nothing either, and the engine's kernel without the skip path is the faster one (block pruning)
    # (--related: pruning on, every band against the best of the whole chain -- bands.py / include/mi355sw.h share_best)
    runner = BandRunner(al, dist=_Dist() if world > 1 else None, rank=rank, world=world, device=None,
                        segment_rows=1 << 15, transport=comm, prune_blocks=args.related)
    attach_chain = None                # bands.InProcessChain on rank 0 when the ranks' hipIpc ports fail their check
    if world > 1:
        runner.reduce_best = lambda b, _r=runner: _reduce_cpu(dist, b, world, coll_device)
    comm_note = None
    if world > 1 and comm == "p2p":
        # every rank creates its column port and maps its neighbour's once, before anything is timed; if any rank
        # cannot (no peer access between two of the GPUs, IPC refused), ALL ranks use the host transport instead
        ok = torch.tensor([1 if runner.probe_p2p(m) else 0], dtype=torch.int32, device=coll_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        usable = int(ok.item()) == 1
        if usable and os.environ.get("MI355SW_BENCH_VERIFY_P2P", "1") != "0":
            # ... and a short chain (1 Mi rows) runs once through the ports and once through the host, untimed: every
            # band must come through within a minute and receive the same column and find the same best cell both
            # ways (bands.verify_p2p) -- a mapping that opens but does not deliver must not reach the measurement
            def all_min(v):
                t = torch.tensor([int(v)], dtype=torch.int32, device=coll_device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return int(t.item())
            usable = runner.verify_p2p(min(m, 1 << 20), j0, j1, all_min, budget_s=60.0)
        if os.environ.get("MI355SW_BENCH_FAIL_IPC") == "1":      # rehearsal of the fall-backs: as if the hipIpc check had failed
            usable = False
            runner.p2p_error = "MI355SW_BENCH_FAIL_IPC=1"
        if not usable:
            # Second device-side transport before anybody touches host memory: rank 0 drives ALL bands from its one process,
            # ports attached with hipDeviceEnablePeerAccess (mi355sw_port_attach, bands.InProcessChain) -- no hipIpc handle, no
            # second process.  Checked like the first: a 1 Mi-row chain must report what one band over all columns reports.
            al.portClose()
            # why: the rank that failed knows, the rank that writes the line may not -- collect every rank's reason (gloo side group)
            whys = [None] * world
            dist.all_gather_object(whys, runner.p2p_error, group=p2p_group)
            whys = ["rank %d: %s" % (r, w) for r, w in enumerate(whys) if w]
            ipc_error = "; ".join(whys[:3]) if whys else "the check failed with no reason given"
            flag = torch.zeros(1, dtype=torch.int32, device=coll_device)
            attach_error = None
            if rank == 0 and os.environ.get("MI355SW_BENCH_NO_ATTACH") != "1":
                try:
                    from masa_cudalign_amd.bands import InProcessChain
                    devs = [0] * world if rehearse else list(range(world))
                    if not rehearse and pkg.engine.load_library().mi355sw_device_count() < world:
                        raise RuntimeError("rank 0 sees fewer than %d GPUs" % world)
                    chain_als = [pkg.MI355Aligner(device=d, rows_per_lane=rows_per_lane, waves=waves) for d in devs]
                    for a2 in chain_als:
                        a2.setSequences(s0, s1)
                    attach_chain = InProcessChain(chain_als, prune_blocks=args.related)
                    mm = min(m, 1 << 20)
                    got, _ = attach_chain.run(mm, lim, **_chain_kw(args))
                    want = _single_band(pkg, al, mm, n, args)
                    if tuple(got) != tuple(want):
                        raise RuntimeError("attached chain reports %r, one band %r" % (tuple(got), tuple(want)))
                    flag += 1
                except Exception as e:                   # noqa: BLE001
                    attach_error = "%s: %s" % (type(e).__name__, e)
                    attach_chain = None
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) == 1:
                comm = "p2p-attach"
                comm_note = "hipIpc ports unavailable (%s): rank 0 drives all %d bands, ports attached with peer access inside its process" % (ipc_error, world)
            else:
                notes = [attach_error] if rank == 0 else [None]
                dist.broadcast_object_list(notes, src=0, group=p2p_group)
                comm_note = ("device-to-device ports unavailable -- hipIpc between the ranks: %s; peer access inside rank 0: %s -- "
                             "pinned host columns + gloo instead" % (ipc_error, notes[0]))
                comm = "host"
                runner.transport = "host"

    # N > 1: what ONE GPU does with the same number of cells as its share of the chain -- (tall*size) x size, one band,
    # no neighbour -- so that the scaling figure compares like with like (the N = 1 line of this bench is C2, a
    # 2-round shape that is ~10 % slower per cell than a tall one).  Rank 0, untimed, before the chain.
    single_ref = None
    if world > 1 and not args.no_single_reference:
        if rank == 0:
            # its own engine, default configuration (the strip height one GPU would pick for this shape, not the chain's),
            # the same recurrence and pruning as the chain; one untimed run first (buffers, seed pass), then the timed one
            mm = args.size * args.tall
            al1 = pkg.MI355Aligner(device=local_rank, waves=waves)
            try:
                al1.setSequences(s0, s1)
                _single_band(pkg, al1, mm, n, args)
                t0s = time.time()
                b1 = _single_band(pkg, al1, mm, n, args)
                dts = time.time() - t0s
                st1 = al1.getStatistics()
            finally:
                al1.close()
            single_ref = {"workload": "%dx%d on rank 0 alone (one GPU's share of the chain's cells), same recurrence and pruning, the "
                                      "engine's own strip height, second of two runs" % (mm, n),
                          "gcups": float(mm) * n / dts / 1e9, "kernel_gcups": float(mm) * n / st1["kernel_ms"] / 1e6, "seconds": dts,
                          "kernel_ms": st1["kernel_ms"], "strip_rows": st1["strip_rows"], "kernel": st1["kernel"],
                          "pruned_fraction": st1["pruned_cells"] / float(mm) / n,
                          "best": {"i": b1[0] + 1, "j": b1[1] + 1, "score": b1[2]}}
        dist.barrier()

    def one_step():
        if comm == "p2p-attach":
            # rank 0 drives every band (the other ranks wait at the fences); the statistics are band 0's, the slowest
            # kernel's time in place of its own
            if rank != 0:
                return (-1, -1, -INF_), {}
            best, sts = attach_chain.run(m, lim, **_chain_kw(args))
            st = dict(sts[0])
            st["kernel_ms"] = max(s["kernel_ms"] for s in sts)
            st["pruned_cells"] = sum(s["pruned_cells"] for s in sts)
            st["bands"] = sts
            return best, st
        if args.nw:
            # global alignment: gap-initialised borders, nothing tracked, the answer is the last band's last cell
            from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, INIT_WITH_GAPS, INF
            got, last = {}, rank == world - 1
            runner.run(m, j0, j1, n_total=n, recurrence=NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=INIT_WITH_GAPS,
                       first_col_init_type=INIT_WITH_GAPS, want_last_row=last,
                       before_end=(lambda eng: got.update(h=int(eng.streamReadLastRow(col=j1 - j0 - 1, length=1)[0, 0]))) if last else None)
            best = (m - 1, n - 1, got["h"]) if last else (-1, -1, -INF)
        else:
            best = runner.run(m, j0, j1, n_total=n)
        if world > 1:
            best = runner.reduce_best(best)
        return best, al.getStatistics()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    INF_ = 999999999
    best = None
    fence()                            # all bands start together: a band's kernel waits a bounded time for its left neighbour
    for _ in range(args.warmup):
        best, _st = one_step()
    fence()
    t0 = time.time()
    kernel_ms, wait_ms, pruned = [], [], []
    for _ in range(args.steps):
        best, st = one_step()
        kernel_ms.append(st.get("kernel_ms", 0.0))
        wait_ms.append(st.get("wait_ms", 0.0))
        pruned.append(st.get("pruned_cells", 0) if comm != "p2p-attach" or rank != 0 else st["bands"][0]["pruned_cells"])
    fence()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ranks = None
    if world > 1:
        # who ran what where: one record per rank, gathered over the gloo side group
        prop = torch.cuda.get_device_properties(local_rank)
        if comm == "p2p-attach" and rank == 0:
            band_stats = st["bands"]
        mine = {"rank": rank, "device": local_rank, "name": prop.name,
                "pci_bus_id": getattr(prop, "pci_bus_id", None), "pci_device_id": getattr(prop, "pci_device_id", None),
                "band_columns": [j0, j1], "kernel_ms": sum(kernel_ms) / len(kernel_ms),
                "seed_ms": st.get("seed_ms", 0.0),            # the pass that gave the chain's pruning bound its first value (band 0 runs it), outside kernel_ms
                "wait_for_left_neighbour_ms": sum(wait_ms) / len(wait_ms),      # per wavefront, in claim_strip_common
                "pruned_cells": sum(pruned) / len(pruned), "restarts": runner.restarts, "p2p_error": runner.p2p_error}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine, group=p2p_group)
        if comm == "p2p-attach" and rank == 0:        # every band ran in rank 0's process: its per-band figures
            for k, r in enumerate(ranks):
                r.update(kernel_ms=band_stats[k]["kernel_ms"], wait_for_left_neighbour_ms=band_stats[k].get("wait_ms", 0.0),
                         pruned_cells=band_stats[k]["pruned_cells"], restarts=attach_chain.restarts, driven_by_rank=0)
    # BASELINE's own multi-GPU configuration for this N, once, behind the timed steps (collective: every rank takes part; whatever
    # goes wrong in it is reported inside the line, it never costs the headline)
    full_cfg, full_out = None, None
    if world > 1 and not args.no_full_config:
        full_cfg = full_config_for(world, rehearse)
    if full_cfg is not None:
        budget = float(os.environ.get("MI355SW_BENCH_EXTRAS_MAX_S", "1800"))
        verdict = [None]
        if rank == 0 and (time.time() - T_BENCH_START) + full_cfg["est_s"] > budget:
            verdict[0] = "skipped: %.0f s of the run gone, about %d s more would pass MI355SW_BENCH_EXTRAS_MAX_S = %.0f" % (
                time.time() - T_BENCH_START, full_cfg["est_s"], budget)
        dist.broadcast_object_list(verdict, src=0, group=p2p_group)
        if verdict[0] is not None:
            full_out = {"workload": full_cfg["workload"], "error": verdict[0]}
        else:
            al.close()                 # the headline's buffers and ports go back first
            if attach_chain is not None:
                for a2 in attach_chain.aligners:
                    a2.close()
            full_out = run_full_config(full_cfg, pkg, torch, dist, _Dist, p2p_group, world, rank, local_rank, comm, rehearse, waves, coll_device, args)
    if rank == 0:
        cells = float(m) * float(n)
        gcups = cells * args.steps / dt / 1e9
        k_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = st["algorithmic_bytes"]
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        band_cells = float(m) * float(j1 - j0)
        kname = "pk16" if st["profile_kernel"] == 2 else "int32"
        # two strip heights in one launch (sw_strip_kernel_pk16_mixed): the library says so itself (mi355sw_stats, ABI 5)
        mixed = st["strip_rows_second"] > 0
        pmc = pmc_lookup(kname, m, j1 - j0, st["strip_rows"]) if world == 1 else None
        out = {
            "metric": "GCUPS (DP cells/sec) Stage-1", "value": gcups, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i16x2 (packed, exact; int32 fallback)" if st["profile_kernel"] == 2 else "int32",
            "data": "synthetic",
            "config": {"workload": (("C2: %dx%d unrelated random ACGT, local SW, score-only" % (m, n)) if not (args.related or args.nw) else
                                    ("%dx%d %s synthetic pair, %s, score-only%s" % (m, n, "RELATED" if args.related else "unrelated",
                                                                                 "global NW (gap-initialised borders)" if args.nw else "local SW",
                                                                                 ", block pruning on" if args.related else ""))) if world == 1 else
                       ("weak scaling, %d x C2's cells per GPU: (%d*%d)x%d %s, %s, score-only; "
                        "%d column bands of %d columns, boundary column GPU to GPU (%s)"
                        % (args.tall, args.size * args.tall, world, n,
                           ("RELATED synthetic pair, block pruning on against the chain-wide %s" % ("lower bound of H[m][n]" if args.nw else "best"))
                           if args.related else "unrelated random ACGT",
                           "global NW (gap-initialised borders)" if args.nw else "local SW", world, n // world, comm)),
                       "m": m, "n": n, "bands": world, "strip_rows": st["strip_rows"], "waves_per_gpu": st["waves"],
                       "reserved_cus": args.reserve_cus,
                       "kernel": {2: "pk16", 1: "int32-profile", 0: "int32-generic"}[st["profile_kernel"]],
                       "comm": comm if world > 1 else "none", "comm_note": comm_note,
                       # who started the ranks: "self" = python3 bench.py --gpus N (self_launch), "external" = a torchrun around it
                       "launcher": (os.environ.get("MI355SW_BENCH_LAUNCHER", "external") if world > 1 else "none"),
                       # does the boundary column cross a GPU-to-GPU link?  Only with column ports between different
                       # devices; the host transport (pinned columns + gloo over loopback) and a rehearsal do not
                       "xgmi": bool(world > 1 and comm in ("p2p", "p2p-attach") and not rehearse),
                       "collectives": ({"backend": dist.get_backend(), "world": dist.get_world_size()} if world > 1 else None),
                       "ranks": ranks, "same_shape_single_gpu": single_ref,
                       "related_pair": bool(args.related), "recurrence": "NW" if args.nw else "SW",
                       "kernel_name": st["kernel"], "strips": st["strips"], "strips_first": st["strips_first"],
                       "strip_rows_second": st["strip_rows_second"], "restarts": runner.restarts,
                       "pruned_fraction": ((sum(r["pruned_cells"] for r in ranks) if ranks else sum(pruned) / len(pruned)) / (float(m) * n)), "kernel_build_id": kernel_build_id(), "build_identity": build_identity()},
            "best": {"i": best[0] + 1, "j": best[1] + 1, "score": best[2]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc["traffic_bytes"] if pmc else None,
                         "traffic_source": (pmc["source"] + " (rocprofv3 --pmc passes on this kernel build, bytes per launch)") if pmc
                                           else "not measured on this kernel build (profiles/pmc_index.json has no entry)",
                         "kernel": st["kernel"],
                         "strips": st["strips"],
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": ("scan kernel: 17 B per column per strip (%d strips of %s rows); the binding unit is VALU issue"
                                  % (st["strips"], ("%d and %d" % (st["strip_rows"], st["strip_rows_second"])) if mixed else str(st["strip_rows"])))},
            "valu_roofline": _valu(st, band_cells, k_ms, pmc),
        }
        # the extras must never cost the headline line: whatever goes wrong in them is reported inside the JSON
        if world == 1 and not args.no_target_shape:
            try:
                out["target_shape"] = target_shape(pkg, local_rank, check=not args.no_cpu_baseline)
            except Exception as e:                       # noqa: BLE001
                out["target_shape"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_shapes:
            for key, fn in (("c3_shape", c3_shape), ("c5_shape", c5_shape)):
                try:
                    out[key] = fn(pkg, local_rank)
                except Exception as e:                   # noqa: BLE001
                    out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_c3_full:
            try:
                out["c3_full"] = c3_full(pkg, local_rank)
            except Exception as e:                       # noqa: BLE001
                out["c3_full"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if full_cfg is not None:
            out[full_cfg["key"]] = full_out
        if world == 1 and not args.no_cpu_baseline:
            for key, fn in (("cpu_baseline", cpu_baseline), ("cpu_baseline_all_cores", cpu_baseline_mt)):
                try:
                    out[key] = fn(pkg)
                except Exception as e:                   # noqa: BLE001
                    out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out), flush=True)
    al.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_full_config(fc, pkg, torch, dist, make_dist, p2p_group, world, rank, local_rank, comm, rehearse, waves, coll_device, args):
    """One of BASELINE's multi-GPU configurations (FULL_CONFIGS) through the band chain, once: the reference runs its real job
    on N devices from one command line (--fork, M/libmasa/libmasa.cpp:540-642; block pruning switched off when it forks,
    :1318-1321 -- here the bands share the running bound through their ports).  Every rank calls this; rank 0 gets the record.
    A rank that fails says so to the others at the end (its neighbours' kernels give up after the wait budget), nobody hangs."""
    import numpy as np
    from masa_cudalign_amd.bands import BandRunner, InProcessChain, band_limits, rows_per_lane_for_bands, check_chain_bound
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, INIT_WITH_GAPS, INF
    m, n, nw, related = fc["m"], fc["n"], fc["nw"], fc["related"]
    out = {"workload": fc["workload"], "bands": world, "comm": comm}
    err, best, al2, chain = None, (-1, -1, -INF), None, None
    mine = {"rank": rank}
    lim = band_limits(n, [1] * world)
    j0, j1 = lim[rank], lim[rank + 1]
    nwargs = argparse.Namespace(nw=nw, related=related)
    try:
        t0 = time.time()
        s0, s1 = (pkg.seqgen.related_pair if related else pkg.seqgen.unrelated_pair)(m, n, cfg=fc["cfg"])
        mine["generate_s"] = time.time() - t0
        R = args.rows_per_lane or rows_per_lane_for_bands(m, lim[1] - lim[0], world, waves or 1024)
        # (how long a band's kernel waits for rows of its boundary column, and its driver for progress, before it gives up: band 7 of
        #  C5 sees its first rows after the seed and seven first-strip sweeps, ~90 s; a neighbour that died is found out after this)
        wait_s = float(os.environ.get("MI355SW_BENCH_EXTRA_WAIT_S", "600"))
        if comm == "p2p-attach":
            if rank == 0:
                devs = [0] * world if rehearse else list(range(world))
                als = [pkg.MI355Aligner(device=d, rows_per_lane=R, waves=waves, wait_seconds=wait_s) for d in devs]
                for a2 in als:
                    a2.setSequences(s0, s1)
                chain = InProcessChain(als, prune_blocks=related)
        else:
            al2 = pkg.MI355Aligner(device=local_rank, rows_per_lane=R, waves=waves, wait_seconds=wait_s)
            al2.setSequences(s0, s1)
            runner = BandRunner(al2, dist=make_dist(), rank=rank, world=world, device=None, segment_rows=1 << 15, transport=comm, prune_blocks=related)
            runner.stall_abort_s = wait_s
    except Exception as e:                                   # noqa: BLE001
        err = "set-up: %s: %s" % (type(e).__name__, e)
    # everybody ready?  (a rank without its engine must not leave the others waiting in the chain)
    ready = [None] * world
    dist.all_gather_object(ready, err, group=p2p_group)
    if any(ready):
        out["error"] = "; ".join("rank %d: %s" % (r, w) for r, w in enumerate(ready) if w)
        return out if rank == 0 else None
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    try:
        if comm == "p2p-attach":
            if rank == 0:
                best, sts = chain.run(m, lim, **_chain_kw(nwargs))
                mine["bands"] = [{"kernel_ms": st["kernel_ms"], "wait_for_left_neighbour_ms": st.get("wait_ms", 0.0), "pruned_cells": st["pruned_cells"],
                                  "seed_ms": st.get("seed_ms", 0.0), "strip_rows": st["strip_rows"], "kernel": st["kernel"]} for st in sts]
                mine["restarts"], mine["initial_bound"] = chain.restarts, chain.initial_bound
        else:
            if nw:
                got, last = {}, rank == world - 1
                runner.run(m, j0, j1, n_total=n, recurrence=NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=INIT_WITH_GAPS,
                           first_col_init_type=INIT_WITH_GAPS, want_last_row=last,
                           before_end=(lambda eng: got.update(h=int(eng.streamReadLastRow(col=j1 - j0 - 1, length=1)[0, 0]))) if last else None)
                best = (m - 1, n - 1, got["h"]) if last else (-1, -1, -INF)
            else:
                best = runner.run(m, j0, j1, n_total=n)
            st = al2.getStatistics()
            mine.update(kernel_ms=st["kernel_ms"], wait_for_left_neighbour_ms=st.get("wait_ms", 0.0), pruned_cells=st["pruned_cells"],
                        seed_ms=st.get("seed_ms", 0.0), strip_rows=st["strip_rows"], kernel=st["kernel"], kernel_launches=st["kernel_launches"],
                        band_columns=[j0, j1], band_best=[int(x) for x in best], restarts=runner.restarts, initial_bound=runner.initial_bound)
    except Exception as e:                                   # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    mine["seconds"] = time.time() - t0
    mine["error"] = err
    torch.cuda.synchronize()
    recs = [None] * world
    dist.all_gather_object(recs, mine, group=p2p_group)      # (also the fence behind the chain: every band is through)
    dt = max(r["seconds"] for r in recs)
    failed = [r for r in recs if r.get("error")]
    if comm != "p2p-attach":
        bests = [tuple(r.get("band_best", (-1, -1, -INF))) for r in recs]
        from masa_cudalign_amd.bands import canonical_best
        best = canonical_best(bests)
    if rank == 0:
        out["ranks"] = recs if comm != "p2p-attach" else recs[0].get("bands")
        if failed:
            out["error"] = "; ".join("rank %d: %s" % (r["rank"], r["error"]) for r in failed)
        else:
            bound = recs[0].get("initial_bound")
            pruned = sum(r.get("pruned_cells", 0) for r in (recs if comm != "p2p-attach" else recs[0]["bands"]))
            out.update({"value": float(m) * n / dt / 1e9, "unit": "GCUPS (m*n, one pass, seed included)", "seconds": dt,
                        "best": {"i": best[0] + 1, "j": best[1] + 1, "score": best[2]}, "pruned_fraction": pruned / float(m) / n,
                        "initial_bound": bound, "generate_s": recs[0].get("generate_s"),
                        "xgmi": bool(comm in ("p2p", "p2p-attach") and not rehearse),
                        "collectives": {"backend": dist.get_backend(), "world": dist.get_world_size()}})
            check = {}
            try:
                if related:
                    check_chain_bound(best[2], bound)
                    check["not_below_the_bound_the_pruning_started_from"] = True
            except Exception as e:                           # noqa: BLE001
                check["not_below_the_bound_the_pruning_started_from"] = False
                out["bound_error"] = str(e)
            exp = fc.get("expect")
            if exp is not None and getattr(pkg.seqgen, "GENERATOR_VERSION", None) != RECORDED_FOR_GENERATOR:
                exp = None                                   # recorded for another generator: compare with one band instead
            if exp is not None:
                check["equals_the_recorded_one_gpu_run"] = (best[0] + 1, best[1] + 1, best[2]) == (exp["i"], exp["j"], exp["score"])
                out["expect"] = exp
            else:
                # nothing recorded at this size (rehearsal): the chain must report what ONE band over all columns reports
                try:
                    al1 = pkg.MI355Aligner(device=local_rank, waves=waves)
                    try:
                        al1.setSequences(s0, s1)
                        want = _single_band(pkg, al1, m, n, nwargs)
                    finally:
                        al1.close()
                    check["equals_one_band_over_all_columns"] = tuple(int(x) for x in want) == tuple(int(x) for x in best)
                except Exception as e:                       # noqa: BLE001
                    check["equals_one_band_over_all_columns"] = False
                    out["single_band_error"] = "%s: %s" % (type(e).__name__, e)
            if not nw and best[1] >= 0:
                # the reported cell under the oracle: the 600 x 600 window that ends at it
                try:
                    oracle = graft.load_oracle()
                    i, j = best[0] + 1, best[1] + 1
                    i0, jj0 = max(0, i - 600), max(0, j - 600)
                    ref = oracle.stage1(s0[i0:i], s1[jj0:j], want_last_row=True)
                    check["oracle_window_600x600_ending_at_the_cell"] = bool(ref["best"][2] == best[2] and int(ref["last_row"][-1][0]) == best[2])
                except Exception as e:                       # noqa: BLE001
                    out["oracle_error"] = "%s: %s" % (type(e).__name__, e)
            check["ok"] = all(v for v in check.values() if isinstance(v, bool)) and len(check) > 0
            out["check"] = check
    for a2 in ([al2] if al2 is not None else []) + (chain.aligners if chain is not None else []):
        try:
            a2.close()
        except Exception:                                    # noqa: BLE001
            pass
    return out if rank == 0 else None


def _chain_kw(args):
    """recurrence and borders of the chain for bands.InProcessChain.run"""
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, SMITH_WATERMAN, INIT_WITH_GAPS, INIT_WITH_ZEROES
    if args.nw:
        return dict(recurrence=NEEDLEMAN_WUNSCH, first_row_init_type=INIT_WITH_GAPS, first_col_init_type=INIT_WITH_GAPS)
    return dict(recurrence=SMITH_WATERMAN, first_row_init_type=INIT_WITH_ZEROES, first_col_init_type=INIT_WITH_ZEROES)


def _single_band(pkg, al, m, n, args):
    """one band over all n columns on one engine, same recurrence / pruning as the chain: (i, j, score) 0-based"""
    from masa_cudalign_amd.bands import BandRunner
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, INIT_WITH_GAPS
    br = BandRunner(al, prune_blocks=args.related)
    if not args.nw:
        return br.run(m, 0, n)
    got = {}
    br.run(m, 0, n, recurrence=NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=INIT_WITH_GAPS, first_col_init_type=INIT_WITH_GAPS,
           want_last_row=True, before_end=lambda eng: got.update(h=int(eng.streamReadLastRow(col=n - 1, length=1)[0, 0])))
    return (m - 1, n - 1, got["h"])


def target_shape(pkg, device, check=True):
    """One step at the north star's HEIGHT: 228,000,000 x 1,000,000 unrelated ACGT, local SW, one launch (111 329
    strips of 2048 rows, two-phase best).  The full 228 M x 228 M matrix is 228 such bands (~2.3 h); this is the part of
    it a bench run can afford.  The reported cell is checked by the oracle on the 600 x 600 window that ends at it."""
    from masa_cudalign_amd.bands import BandRunner
    m, n = 228000000, 1000000
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=5)
    al = pkg.MI355Aligner(device=device)
    try:
        al.setSequences(s0, s1)
        t0 = time.time()
        best = BandRunner(al).run(m, 0, n)
        dt = time.time() - t0
        st = al.getStatistics()
    finally:
        al.close()
    out = {"workload": "north-star height: %dx%d unrelated random ACGT, local SW, score + canonical position, one launch" % (m, n),
           "value": float(m) * n / dt / 1e9, "unit": "GCUPS", "seconds": dt, "kernel_ms": st["kernel_ms"],
           "strip_rows": st["strip_rows"], "strips": st["strips"], "kernel_launches": st["kernel_launches"],
           "best": {"i": best[0] + 1, "j": best[1] + 1, "score": best[2]}}
    if check:
        oracle = graft.load_oracle()
        i, j = best[0] + 1, best[1] + 1
        i0, j0 = max(0, i - 600), max(0, j - 600)
        ref = oracle.stage1(s0[i0:i], s1[j0:j], want_last_row=True)
        out["check"] = {"oracle_window": "600x600 ending at the reported cell",
                        "ok": bool(ref["best"][2] == best[2] and int(ref["last_row"][-1][0]) == best[2])}
    return out


def _sha(a):
    import hashlib
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).hexdigest()


def c3_shape(pkg, device):
    """BASELINE config 3's kind of work in the driver-run line: the first 1 M columns of a 48 M x 46 M RELATED pair (C3's
    height; the band's pruning bound looks at the whole 46 M columns like band 0 of a chain would), local SW with special
    rows, once without and once with block pruning.  The pruned run must report the same best cell, and its special rows
    must be lower bounds of the unpruned ones (sha256 of those recorded) with the row maximum intact above the best cell."""
    import numpy as np
    from masa_cudalign_amd.bands import BandRunner
    m, n, n_total = 48000000, 1000000, 46000000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=3)
    out = {"workload": "C3-shaped: %dx%d band of a related pair (bound against %d columns), local SW, special rows every 8 Mi rows, "
                       "without and with block pruning" % (m, n, n_total)}
    res = {}
    al = pkg.MI355Aligner(device=device)
    try:
        al.setSequences(s0, s1)
        for prune in (False, True, "own"):
            # ("own": the band as a matrix in its own right -- the bound sees only its 1 M columns, nearly everything below
            #  the alignment goes: runs of skipped slabs and whole skipped strips at C3's height)
            rows = {}
            t0 = time.time()
            br = BandRunner(al, prune_blocks=bool(prune))
            best = br.run(m, 0, n, special_row_interval=8 << 20, n_total=n_total if prune != "own" else n,
                          special_row_sink=lambda dp, c0, cells: rows.__setitem__(dp, cells.copy()))
            dt = time.time() - t0
            st = al.getStatistics()
            st["restarts"] = br.restarts
            res[prune] = (best, rows)
            out[{False: "plain", True: "pruned", "own": "pruned_own_extent"}[prune]] = {"value": float(m) * n / dt / 1e9, "unit": "GCUPS (m*n)", "seconds": dt, "kernel_ms": st["kernel_ms"],
                                                   "kernel": st["kernel"], "strip_rows": st["strip_rows"], "restarts": st["restarts"],
                                                   "pruned_fraction": st["pruned_cells"] / float(m) / n,
                                                   "best": {"i": best[0] + 1, "j": best[1] + 1, "score": best[2]}}
    finally:
        al.close()
    (b0, r0), (b1, r1), (b2, r2) = res[False], res[True], res["own"]
    out["special_rows"] = {str(dp): _sha(r0[dp]) for dp in sorted(r0)}
    lower = all(bool(np.all(r1[dp] <= r0[dp])) and bool(np.all(r2[dp] <= r0[dp])) for dp in r0)
    maxima = all(int(r1[dp][:, 0].max()) == int(r0[dp][:, 0].max()) == int(r2[dp][:, 0].max()) for dp in r0 if dp <= b0[0])
    out["check"] = {"same_best_cell": tuple(b0) == tuple(b1) == tuple(b2), "same_rows": sorted(r0) == sorted(r1) == sorted(r2) and len(r0) >= 4,
                    "pruned_rows_are_lower_bounds": lower, "row_maxima_above_the_best_cell_intact": maxima}
    out["check"]["ok"] = all(out["check"].values())
    return out


# C3's stage 1 without pruning, swept once on this engine family (builder-side, 6 minutes of GPU: tools/scale_run.py c3 ->
# profiles/r04_scale_c3_48Mx46M.json "unpruned"): the best cell a pruned run of the same pair must report (0-based i, j)
# (recorded for seqgen.GENERATOR_VERSION 1: with another generator the comparison is skipped, not failed -- ADVICE round 5)
C3_RECORDED_UNPRUNED_BEST = (45999788, 45999999, 35906671)
RECORDED_FOR_GENERATOR = 1


def c3_full(pkg, device):
    """BASELINE config 3's stage 1 AT FULL SIZE under the driver's clock: 48 M x 46 M related pair, local SW, block pruning on
    (the bound starts from the diagonal seed pass), special rows every 2 Mi rows.  GCUPS in the reference's m*n convention
    (sw_stage1.cpp:440-448) over the whole call, seed included.  Checks that do not take the run's own word: the first 3 M
    columns of the same matrix swept WITHOUT pruning (a band with a zero first column is a matrix in its own right: exact
    cells) -- every special row of the pruned run is a lower bound of it there, and where the alignment crosses a row inside
    the band the row's maximum is the same, at the same column; the rows' maxima rise strictly down to the best cell; the
    best cell is the one the unpruned sweep of the whole matrix reported (recorded)."""
    import numpy as np
    from masa_cudalign_amd.bands import BandRunner
    m, n, band = 48000000, 46000000, 3000000
    interval = 2 << 20
    t0 = time.time()
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=3)
    out = {"workload": "C3 stage 1 at full size: %dx%d related synthetic pair, local SW, block pruning on behind the diagonal seed, "
                       "special rows every %d rows" % (m, n, interval), "generate_s": time.time() - t0}
    al = pkg.MI355Aligner(device=device, max_special_bytes=64 << 30)
    try:
        al.setSequences(s0, s1)
        kept, maxima = {}, {}

        def sink(dp, c0, cells):
            kept[dp] = cells[:band].copy()
            h = cells[:, 0]
            maxima[dp] = (int(h.max()), int(h.argmax()))
        t0 = time.time()
        br = BandRunner(al, prune_blocks=True)
        best = br.run(m, 0, n, special_row_interval=interval, special_row_sink=sink)
        dt = time.time() - t0
        st = al.getStatistics()
        out.update({"value": float(m) * n / dt / 1e9, "unit": "GCUPS (m*n, seed included)", "seconds": dt, "kernel_ms": st["kernel_ms"],
                    "seed_ms": st["seed_ms"], "kernel": st["kernel"], "strip_rows": st["strip_rows"], "restarts": br.restarts,
                    "pruned_fraction": st["pruned_cells"] / float(m) / n, "computed_cells_gcups": st["processed_cells"] / st["kernel_ms"] / 1e6,
                    "special_rows": len(kept), "best": {"i": best[0] + 1, "j": best[1] + 1, "score": best[2]}})
        # the same rows of the first `band` columns, every cell computed
        exact = {}
        t0 = time.time()
        BandRunner(al).run(m, 0, band, special_row_interval=interval, special_row_sink=lambda dp, c0, cells: exact.__setitem__(dp, cells.copy()))
        out["unpruned_band"] = {"columns": band, "seconds": time.time() - t0, "kernel": al.getStatistics()["kernel"]}
    finally:
        al.close()
    lower = sorted(kept) == sorted(exact) and all(bool(np.all(kept[dp] <= exact[dp])) for dp in kept)
    crossed = [dp for dp in sorted(exact) if int(exact[dp][:, 0].argmax()) < band - 4096 and int(exact[dp][:, 0].max()) > 1000]
    same_max = all(int(kept[dp][:, 0].max()) == int(exact[dp][:, 0].max()) and int(kept[dp][:, 0].argmax()) == int(exact[dp][:, 0].argmax()) for dp in crossed)
    # a row above the best cell: the optimal path crosses it, gains at most one per row from there on, and nothing beats the best
    above = [(dp, maxima[dp][0]) for dp in sorted(maxima) if dp <= best[0]]
    within = all(best[2] - (best[0] + 1 - dp) <= mx <= best[2] for dp, mx in above)
    out["check"] = {"pruned_rows_are_lower_bounds_of_the_unpruned_band": bool(lower), "rows_crossed_inside_the_band": len(crossed),
                    "row_maxima_equal_where_the_alignment_crosses_the_band": bool(same_max and len(crossed) >= 1),
                    "row_maxima_above_the_best_cell_within_reach_of_it": bool(len(above) >= 10 and within),
                    "best_cell_is_the_recorded_unpruned_sweeps": (tuple(best) == C3_RECORDED_UNPRUNED_BEST)
                    if getattr(pkg.seqgen, "GENERATOR_VERSION", None) == RECORDED_FOR_GENERATOR else "skipped: another sequence generator",
                    "no_restart": br.restarts == 0}
    out["check"]["ok"] = all(v for v in out["check"].values() if isinstance(v, bool))
    return out


def c5_shape(pkg, device):
    """BASELINE config 5's kind of work: 249 M rows (C5's height) x 256 k columns, GLOBAL NW with gap-initialised
    borders.  The packed kernel's H[m][n] and last row (sha256) must be the int32 kernel's; then the same band as band 0
    of C5 (the bound looks at all 228 M columns) with block pruning: lower bounds of the unpruned last row."""
    import numpy as np
    from masa_cudalign_amd.bands import BandRunner
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, INIT_WITH_GAPS
    m, n, n_total = 249000000, 262144, 228000000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
    out = {"workload": "C5-shaped: %dx%d global NW, gap-initialised borders; packed kernel against the int32 kernel, then with block "
                       "pruning as band 0 of %d columns" % (m, n, n_total)}
    rows = {}
    for key, flags, prune in (("packed", 0, False), ("int32", 2, False), ("packed_pruned", 0, True)):
        al = pkg.MI355Aligner(device=device, flags=flags)
        try:
            al.setSequences(s0, s1)
            got = {}
            t0 = time.time()
            br = BandRunner(al, prune_blocks=prune)
            br.run(m, 0, n, recurrence=NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=INIT_WITH_GAPS,
                   first_col_init_type=INIT_WITH_GAPS, want_last_row=True, n_total=n_total if prune else None,
                   before_end=lambda eng: got.update(row=eng.streamReadLastRow()))
            dt = time.time() - t0
            st = al.getStatistics()
            st["restarts"] = br.restarts
        finally:
            al.close()
        rows[key] = got["row"]
        out[key] = {"value": float(m) * n / dt / 1e9, "unit": "GCUPS (m*n)", "seconds": dt, "kernel_ms": st["kernel_ms"], "kernel": st["kernel"],
                    "strip_rows": st["strip_rows"], "restarts": st["restarts"], "pruned_fraction": st["pruned_cells"] / float(m) / n,
                    "h_last_cell": int(got["row"][-1, 0]), "last_row_sha256": _sha(got["row"])}
    out["check"] = {"packed_equals_int32": bool(np.array_equal(rows["packed"], rows["int32"])),
                    "pruned_last_row_is_a_lower_bound": bool(np.all(rows["packed_pruned"] <= rows["packed"])),
                    "no_restart": out["packed"]["restarts"] == 0 and out["packed_pruned"]["restarts"] == 0}
    out["check"]["ok"] = all(out["check"].values())
    return out


def _valu(st, band_cells, k_ms, pmc):
    """the unit that binds this kernel: vector instructions issued per second (SQ_INSTS_VALU of this build's launch, per
    wavefront) against the SIMDs' issue rate for the packed instruction class"""
    if not pmc or not pmc.get("valu_per_launch"):
        return {"valu_instr_per_launch": None, "frac": None, "peak_wave_instr_per_s": VALU_PEAK_WAVE_INSTR,
                "note": "SQ_INSTS_VALU not measured on this kernel build (profiles/pmc_index.json has no entry)"}
    per_launch = pmc["valu_per_launch"]
    achieved = per_launch / (k_ms * 1e-3)
    return {"valu_instr_per_launch": per_launch, "valu_instr_per_cell": per_launch / band_cells,
            "achieved_wave_instr_per_s": achieved, "peak_wave_instr_per_s": VALU_PEAK_WAVE_INSTR, "frac": achieved / VALU_PEAK_WAVE_INSTR,
            "frac_of_one_instr_per_4_cycles": achieved / (256 * 4 * 2.38e9 / 4),
            "source": pmc["source"], "peak_source": "profiles/r02_valu_issue_rate.md (v_pk_max_i16, 4 wavefronts per SIMD: 0.536 / ns / SIMD; "
                                                  "a lone wavefront issues one per 4 cycles at 2.38 GHz = 0.595 / ns / SIMD)"}


def _reduce_cpu(dist, best, world, device):
    import torch
    from masa_cudalign_amd.bands import canonical_best
    t = torch.tensor(list(best), dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return canonical_best([tuple(int(x) for x in o.tolist()) for o in out])


if __name__ == "__main__":
    sys.exit(main() or 0)
