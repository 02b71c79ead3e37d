"""Identity of the device code: sha256 over everything the kernels are built from (this directory's .hip, .inc, .h, .py,
.sh and Makefile).  `python3 build_id.py header` prints the C header the runtime compiles in (mi355sw_build_id());
bench.py and tools/pmc_index.py import kernel_build_id() and only quote PMC figures measured on the SAME id."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_build_id(src=HERE):
    h = hashlib.sha256()
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".inc", ".h", ".py", ".sh")) or fn == "Makefile":
            h.update(fn.encode())
            h.update(open(os.path.join(src, fn), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "header":
        print('#define MI355SW_BUILD_ID "%s"' % kernel_build_id())
    else:
        print(kernel_build_id())
