"""Identity of the library: sha256 over everything it is built from -- this directory's .hip, .inc, .h, .cpp, .py, .sh and
Makefile, and the public header.  (Round 4: the host runtime is part of it.  Which kernel runs, with which strip heights
and how many wavefronts, is decided in runtime.cpp -- a policy edit changes what a measurement measured as surely as a
kernel edit does.)  `python3 build_id.py header` prints the C header the runtime compiles in (mi355sw_build_id());
bench.py and tools/pmc_index.py import kernel_build_id() and only quote PMC figures measured on the SAME id."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_build_id(src=HERE):
    h = hashlib.sha256()
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".inc", ".h", ".cpp", ".py", ".sh")) or fn == "Makefile":
            h.update(fn.encode())
            h.update(open(os.path.join(src, fn), "rb").read())
    api = os.path.join(src, "..", "..", "include", "mi355sw.h")
    if os.path.exists(api):
        h.update(b"mi355sw.h")
        h.update(open(api, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "header":
        print('#define MI355SW_BUILD_ID "%s"' % kernel_build_id())
    else:
        print(kernel_build_id())
