import sys, os, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle()
from helpers import load_golden, make_pair
from oracle.binding import _write_fasta
case = [c for c in load_golden()["cases"] if c["name"] == "full_pipeline_3000x2700"][0]
s0, s1 = make_pair(pkg, case["seq"])
tmp = tempfile.mkdtemp()
_write_fasta(tmp + "/s0.fasta", s0, "s0"); _write_fasta(tmp + "/s1.fasta", s1, "s1")
which = sys.argv[1]
p = subprocess.run([os.path.join(g.ROOT, "oracle/_ref", which), "--work-dir=" + tmp + "/work", "--disk-size=200K"] + sys.argv[2:] + [tmp + "/s0.fasta", tmp + "/s1.fasta"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=tmp)
out = os.path.join(g.ROOT, "gpurun_out", "dropin_" + which)
shutil.rmtree(out, ignore_errors=True); os.makedirs(out)
for f in ["alignment.00.txt", "crosspoints/crosspoint_02.00", "crosspoints/crosspoint_03.00", "crosspoints/crosspoint_04.00", "statistics_01.00", "statistics_02.00"]:
    if os.path.exists(tmp + "/work/" + f): shutil.copy(tmp + "/work/" + f, out + "/" + f.replace("/", "_"))
open(out + "/log.txt", "wb").write(p.stdout)
print(which, p.returncode)
