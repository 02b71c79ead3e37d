"""rocprofv3 (ROCm 7.2) writes rocpd sqlite databases by default.  This turns them into the small text summaries
kept under profiles/:
  python tools/rocpd_summary.py stats  <results.db> out_kernel_stats.csv out_dispatches.csv
  python tools/rocpd_summary.py pmc    out.json "<command note>" <results.db> [more.db ...]
"""
import collections, csv, json, sqlite3, sys

def stats(db, out_stats, out_disp):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, duration, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size from kernels order by start"))
    by = collections.defaultdict(list)
    for r in rows:
        by[r[0]].append(r[3])
    tot = sum(sum(v) for v in by.values())
    with open(out_stats, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)])
    with open(out_disp, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "StartNs", "DurationNs", "Grid", "Workgroup", "VGPR", "AGPR", "SGPR", "LDS", "Scratch"])
        for r in rows:
            if r[3] > 1000000:
                w.writerow([r[0], r[1], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]])
    print(open(out_stats).read())

def pmc(out, note, dbs):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    regs = {}
    for db in dbs:
        c = sqlite3.connect(db)
        for name, disp, cname, val, vg, ag, sg, lds, scr in c.execute(
                "select kernel_name, dispatch_id, counter_name, value, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, scratch_size from counters_collection"):
            k = name.split("(")[0]
            acc[k][cname][(db, disp)] += val
            regs[k] = {"vgpr": vg, "agpr": ag, "sgpr": sg, "lds": lds, "scratch": scr}
    res = {"command": note, "counters": {}, "registers": regs}
    for k, cs in acc.items():
        res["counters"][k] = {cn: sum(v.values()) / len(v) for cn, v in cs.items()}
        res["counters"][k]["dispatches"] = max(len(v) for v in cs.values())
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res["counters"], indent=1, sort_keys=True))

if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:5])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4:])
