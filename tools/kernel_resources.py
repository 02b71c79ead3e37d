#!/usr/bin/env python3
"""tools/kernel_resources.py [OUT]: what every assembled kernel of the library occupies, from the code objects' own
metadata (llvm-readelf --notes, kept as csrc/_obj/*.notes.txt by hipcc_aligned.sh): registers, scratch, LDS.
rocprofv3's dispatch table shows the granulated allocation ("VGPR 256, AGPR 0" for a kernel that claims a255); these are
the counts the assembler recorded.  Writes a table (default: stdout)."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "masa-cudalign_amd", "csrc", "_obj")
sys.path.insert(0, os.path.join(ROOT, "masa-cudalign_amd", "csrc"))
KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size",
        ".vgpr_spill_count", ".sgpr_spill_count", ".wavefront_size", ".max_flat_workgroup_size")


def kernels(path):
    cur = None
    for ln in open(path):
        m = re.match(r"\s*-?\s*(\.[a-z_]+):\s*(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == ".agpr_count" or (k == ".args" and cur is None):
            pass
        if ln.lstrip().startswith("- .") and cur is not None and ".name" in cur and k in (".agpr_count", ".args"):
            yield cur
            cur = None
        if cur is None:
            cur = {}
        cur[k] = v
    if cur and ".name" in cur:
        yield cur


def demangle(name):
    import subprocess
    try:
        return subprocess.run(["c++filt", name.strip()], stdout=subprocess.PIPE).stdout.decode().strip()
    except OSError:
        return name


def main(out):
    import build_id
    rows = []
    for fn in sorted(glob.glob(os.path.join(OBJ, "*.notes.txt"))):
        txt = open(fn).read()
        # one YAML list item per kernel under amdhsa.kernels; items start with "  - .agpr_count" or "  - .args"
        body = txt.split("amdhsa.kernels:")[1].split("amdhsa.target:")[0] if "amdhsa.kernels:" in txt else ""
        for item in re.split(r"\n  - ", "\n" + body)[1:]:
            rec = {}
            for ln in item.splitlines():
                m = re.match(r"\s*(\.[a-z_]+):\s*(\S.*)$", ln)
                if m and m.group(1) in KEYS + (".name",) and m.group(1) not in rec:
                    rec[m.group(1)] = m.group(2).strip().strip("'")
            if ".name" in rec and ".vgpr_count" in rec:
                rows.append((os.path.basename(fn).replace(".notes.txt", ""), rec))
    w = sys.stdout if out is None else open(out, "w")
    w.write("# kernel resources of device build %s (llvm-readelf --notes of the assembled code objects)\n" % build_id.kernel_build_id())
    w.write("# unit | kernel | vgpr | agpr | sgpr | scratch B/lane | LDS B | vgpr spills | sgpr spills\n")
    for unit, r in rows:
        w.write("%s | %s | %s | %s | %s | %s | %s | %s | %s\n" % (
            unit, demangle(r[".name"]), r.get(".vgpr_count"), r.get(".agpr_count"), r.get(".sgpr_count"),
            r.get(".private_segment_fixed_size"), r.get(".group_segment_fixed_size"), r.get(".vgpr_spill_count"), r.get(".sgpr_spill_count")))
    if out is not None:
        w.close()
        print("wrote %s (%d kernels)" % (out, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
