#!/usr/bin/env python3
"""Registers / spills / scratch of every kernel of a BUILT translation unit, from the code object's metadata (no recompile).
usage: co_resources.py compute-engine_amd/csrc/obj/lce_tu_stream_f32.o [name filter]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin/"
obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    # (an explicit output file: without one llvm-objcopy rewrites its INPUT in place, and make would see a newer object)
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows, cur = [], {}
for line in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "agpr_count" and cur.get("name"):
        rows.append(cur)
        cur = {}
    cur[k] = v
if cur.get("name"):
    rows.append(cur)
seen = set()
for r in rows:
    nm = r.get("name", "")
    if not nm or nm in seen:
        continue
    seen.add(nm)
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if flt and flt not in dem:
        continue
    print("%-64s VGPR %4s AGPR %4s SGPR %4s scratch %5s sgpr-spill %3s vgpr-spill %3s" % (
        dem, r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"), r.get("private_segment_fixed_size"),
        r.get("sgpr_spill_count"), r.get("vgpr_spill_count")))
