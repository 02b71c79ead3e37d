import sys, os, subprocess, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle()
from helpers import load_golden, make_pair
from oracle.binding import _write_fasta
case = [c for c in load_golden()["cases"] if c["name"] == "sw_special_rows_20000x9000"][0]
s0, s1 = make_pair(pkg, case["seq"])
tmp = tempfile.mkdtemp()
_write_fasta(tmp + "/s0.fasta", s0, "s0"); _write_fasta(tmp + "/s1.fasta", s1, "s1")
p = subprocess.run([os.path.join(g.ROOT, "oracle/_ref/masa_mi355"), "--work-dir=" + tmp + "/work", "--stage-1", "--disk-size=200K", "--no-block-pruning", tmp + "/s0.fasta", tmp + "/s1.fasta"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=tmp)
print(p.stdout.decode()[-1500:])
for r, d, f in os.walk(tmp + "/work/special_rows"):
    print(r, sorted(f))
