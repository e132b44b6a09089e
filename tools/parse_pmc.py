#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (tools/gpu_pmc.sh) per kernel: mean counter value per
launch, plus HBM bytes per launch with the gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE counts
64 B per 128-B request for wide coalesced streams -> calibrated on the LceQuantize stream,
whose algorithmic bytes are exact)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
    per_dispatch = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("void ", "") + " grid=" + r["Grid_Size"]
    for (disp, ctr), v in per_dispatch.items():
        vals[names[disp]][ctr].append(v)
out = {}
for k, ctrs in vals.items():
    short = k
    if short.startswith("at::native") or "rocclr" in short:
        continue
    out[short] = {c: sum(v) / len(v) for c, v in ctrs.items()}
    out[short]["launches"] = max(len(v) for v in ctrs.values())
print(json.dumps(out, indent=1))
