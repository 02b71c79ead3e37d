"""Two FASTA files in, alignment.00.txt out: the native pipeline (masa-cudalign_amd/pipeline.py, stages 1-6 on one
MI355X) behind the handful of MASA-CUDAlign options that concern the path -- not a re-build of its command line.

    python tools/align_fasta.py [options] seq0.fasta seq1.fasta
      --work-dir=DIR            work directory in MASA-Core's layout (default ./work.tmp); a killed stage 1 resumes from it
      --disk-size=N[KMG]        Special Rows Area budget on disk (default: 8192-row spacing, at most 4 GiB; the traceback needs special rows)
      --ram-size=N[KMG]         part of the special rows kept in memory instead (they alternate with the rows on disk)
      --trim=I0,I1,J0,J1  --reverse=1|2|both  --complement=1|2|both  --reverse-complement=1|2|both  --clear-n
      --alignment-edges=XY      X start, Y end: * anywhere (local), 1 / 2 on that sequence's edge, 3 on either, + on both (global)
      --max-alignments=N        trace back up to N different alignments (alignment.00.txt .. alignment.NN.txt)
      --no-block-pruning  --prune-global (block pruning for a global alignment too)  --gpu=ID  --stage-1 (best score only)

Prints one JSON line (best score, crosspoints per stage, seconds per stage) and leaves alignment.00.txt in the work
directory.  The engine has no CPU fallback: without an MI355X this fails with MI355SW_ENOGPU."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402


def _size(v):
    mult = {"K": 1024, "M": 1024 ** 2, "G": 1024 ** 3}.get(v[-1:].upper())
    return int(float(v[:-1]) * mult) if mult else int(float(v))


def _flags(v):
    return [v in ("1", "both"), v in ("2", "both")]


def main(argv):
    pkg = g.load_package()
    from masa_cudalign_amd import fasta, pipeline, stage1
    edge = {"*": pkg.AT_ANYWHERE, "1": pkg.AT_SEQUENCE_1, "2": pkg.AT_SEQUENCE_2, "3": pkg.AT_SEQUENCE_1_OR_2,
            "+": pkg.AT_SEQUENCE_1_AND_2}
    work, limit, device, prune, only1, edges, count, ram = "./work.tmp", None, 0, True, False, "**", 1, 0
    trim, rev, comp, clear_n = [0, 0, 0, 0], [False, False], [False, False], False
    prune_global = False
    files = []
    for a in argv:
        if a.startswith("--work-dir="):
            work = a[11:]
        elif a.startswith("--disk-size="):
            limit = _size(a[12:])
        elif a.startswith("--ram-size="):
            ram = _size(a[11:])
        elif a.startswith("--trim="):
            trim = [int(x) for x in a[7:].split(",")]
        elif a.startswith("--reverse="):
            rev = _flags(a[10:])
        elif a.startswith("--complement="):
            comp = _flags(a[13:])
        elif a.startswith("--reverse-complement="):
            rev = comp = _flags(a[21:])
        elif a == "--clear-n":
            clear_n = True
        elif a.startswith("--alignment-edges="):
            edges = a[18:]
        elif a.startswith("--max-alignments="):
            count = int(a[17:])
        elif a == "--no-block-pruning":
            prune = False
        elif a == "--prune-global":
            prune_global = True
        elif a.startswith("--gpu="):
            device = int(a[6:])
        elif a == "--stage-1":
            only1 = True
        elif a.startswith("-"):
            raise SystemExit("unknown option %s\n\n%s" % (a, __doc__))
        else:
            files.append(a)
    if len(files) != 2 or len(edges) != 2 or edges[0] not in edge or edges[1] not in edge or len(trim) != 4:
        raise SystemExit(__doc__)
    seqs = [fasta.load(files[k], fasta.SequenceModifiers(clear_n=clear_n, reverse=rev[k], complement=comp[k],
                                                         trim_start=trim[2 * k], trim_end=trim[2 * k + 1])) for k in (0, 1)]
    if limit is None:
        limit = min((len(seqs[0]) // 8192 + 2) * (len(seqs[1]) + 1) * 8, 4 << 30)
    al = pkg.MI355Aligner(device=device)
    try:
        if only1:
            bounds = (seqs[0].offset0 - 1, seqs[1].offset0 - 1, seqs[0].offset1, seqs[1].offset1)
            r = stage1.stage1(al, seqs[0].data(), seqs[1].data(), work, alignment_start=edge[edges[0]],
                              alignment_end=edge[edges[1]], sra_limit=limit, block_pruning=prune, bounds=bounds,
                              progress=sys.stderr, max_alignments=count, ram_limit=ram, prune_global=prune_global)
            res = {"best": list(r["best"]), "seconds": {"1": r["seconds"]}, "gcups": r["gcups"]}
        else:
            out = pipeline.align(al, seqs[0], seqs[1], work, alignment_start=edge[edges[0]], alignment_end=edge[edges[1]],
                                 sra_limit=limit, block_pruning=prune, progress=sys.stderr, max_alignments=count, ram_limit=ram,
                                 prune_global=prune_global)
            res = {"best": list(out["best"]), "seconds": {str(k): v for k, v in out["seconds"].items()},
                   "crosspoints": {str(k): v for k, v in out["crosspoints"].items()},
                   "alignments": [os.path.join(work, "alignment.%02d.txt" % k) for k in range(len(out["alignments"]))]}
    finally:
        al.close()
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
