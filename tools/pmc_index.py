#!/usr/bin/env python3
"""tools/pmc_index.py TAG: refresh profiles/pmc_index.json from profiles/TAG_pk16_{hbm,sq}_pmc.json and
profiles/TAG_bench_line.json (written by tools/pmc_collect.sh on the GPU box and copied into profiles/).
The index is keyed by bench.kernel_build_id(); bench.py only quotes PMC-derived figures for the build they were
measured on.  HBM traffic = FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, in bytes
(the counters are in KiB); VALU instructions per launch = SQ_INSTS_VALU of the dominant kernel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main(tag):
    prof = os.path.join(ROOT, "profiles")
    line = json.loads(open(os.path.join(prof, "%s_bench_line.json" % tag)).read().strip().splitlines()[-1])
    cfg = line["config"]
    if cfg["kernel_build_id"] != bench.kernel_build_id():
        raise SystemExit("profiles/%s_bench_line.json was measured on build %s, the tree is %s: re-run tools/pmc_collect.sh"
                         % (tag, cfg["kernel_build_id"], bench.kernel_build_id()))
    hbm = json.load(open(os.path.join(prof, "%s_pk16_hbm_pmc.json" % tag)))
    sq = json.load(open(os.path.join(prof, "%s_pk16_sq_pmc.json" % tag)))
    # the kernel of the bench line (the profiler prints template arguments with a space after every comma); the profiled command
    # may have run other kernels too (fills, the seed pass)
    want = cfg["kernel_name"].replace(" ", "")
    named = [k for k in hbm["counters"] if want in k.replace(" ", "")]
    dom = named[0] if named else max(hbm["counters"], key=lambda k: hbm["counters"][k].get("WRITE_SIZE", 0))
    fetch, write = hbm["counters"][dom]["FETCH_SIZE"], hbm["counters"][dom]["WRITE_SIZE"]
    traffic = (2.0 * fetch + write) * 1024.0
    valu = sq["counters"][dom]["SQ_INSTS_VALU"]
    path = os.path.join(prof, "pmc_index.json")
    try:
        idx = json.load(open(path))
    except (OSError, ValueError):
        idx = {}
    key = "%s:%dx%d:%d" % (cfg["kernel"], cfg["m"], cfg["n"], cfg["strip_rows"])
    idx.setdefault(cfg["kernel_build_id"], {})[key] = {
        "traffic_bytes": traffic, "fetch_size_kib": fetch, "write_size_kib": write, "valu_per_launch": valu,
        "kernel": dom, "source": "profiles/%s_pk16_hbm_pmc.json, profiles/%s_pk16_sq_pmc.json" % (tag, tag)}
    json.dump(idx, open(path, "w"), indent=1, sort_keys=True)
    print(key, idx[cfg["kernel_build_id"]][key])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
