#!/usr/bin/env python3
"""Prints VGPR/AGPR/SGPR/scratch/occupancy for every kernel of the product library."""
import re
import subprocess
import sys

# one translation unit per kernel family (csrc/Makefile); only the streaming kernel's is built with -amdgpu-mfma-vgpr-form
#   usage: kernel_resources.py [name filter] [tu ...]     (default: every translation unit)
VF = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
TUS = {"lce_tu_valu": [], "lce_tu_mfma_ws": [], "lce_tu_mfma_direct": [], "lce_tu_mfma_2d": [], "lce_tu_pointwise": [],
       **{"lce_tu_stream_" + p: VF for p in ("f32", "f32_clamp", "i8", "i8_floor", "bitpacked")},
       **{"lce_tu_wstream_" + p: [] for p in ("f32", "i8", "i8_floor", "bitpacked")}}
txt = ""
for tu in (sys.argv[2:] or TUS):
    cmd = ["hipcc", "-DLCE_PRODUCT_BUILD", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", *TUS[tu], "-Icompute-engine_amd/csrc",
           "-Rpass-analysis=kernel-resource-usage", "-c", "compute-engine_amd/csrc/%s.hip" % tu, "-o", "/dev/null"]
    txt += subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-R", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if flt and flt not in name:
        continue
    print("%-46s VGPR %4d AGPR %4d SGPR %4d scratch %5d occ %d" % (
        name, v.get("VGPRs", 0), v.get("AGPRs", 0), v.get("TotalSGPRs", 0), v.get("ScratchSize", 0), v.get("Occupancy", 0)) + " sgpr-spill %d" % v.get("SGPRs Spill", 0))
