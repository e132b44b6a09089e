#!/usr/bin/env python
"""SHA-256 of the kernel sources that the planner's sweep tables were measured with: the 3x3 kernel families' headers with comments
and whitespace removed (a comment-only edit does not age a sweep).  tools/engine_sweep.py records it in the table's first line;
tests/test_planner_choice.py refuses a table recorded with other sources than the tree's (round-5 review, item 6).
usage: kernel_hash.py            -> prints the hash"""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["lce_kernels.h", "lce_kernels_mfma.h", "lce_kernels_stream.h", "lce_kernels_wstream.h", "lce_device_intrinsics.h", "lce_kernel_args.h"]


def kernel_sources_hash() -> str:
    h = hashlib.sha256()
    for f in FILES:
        text = open(os.path.join(ROOT, "compute-engine_amd", "csrc", f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(f.encode())
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_hash())
