#!/usr/bin/env python
"""Per-dispatch durations from a rocprofv3 --kernel-trace CSV, summarised so that the figure can be compared with a HIP-event
timing of the same process: tools/run_one.py (and bench.py) run an untimed clock spin-up first -- the first ~50 launches of a
process run 10-35 % slow -- so the average over ALL dispatches of a kernel is not the steady-state launch duration (round 3's
l0_f32 stats averaged 195.1 us where the timed launches took 185.4).
usage: steady_stats.py <kernel_trace.csv> <N>     -> per kernel: all dispatches, and the LAST N (the timed ones)"""
import csv
import statistics
import sys
from collections import defaultdict

path, n_last = sys.argv[1], int(sys.argv[2])
rows = defaultdict(list)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "at::native" in name or "rocclr" in name or "elementwise" in name:
        continue
    rows[name].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for name, d in rows.items():
    d.sort()
    dur = [x[1] for x in d]
    last = dur[-n_last:] if len(dur) >= n_last else dur
    print("%-72s all %5d: mean %8.2f us | last %4d (timed): mean %8.2f  median %8.2f  min %8.2f  max %8.2f us" % (
        name[:72], len(dur), statistics.mean(dur), len(last), statistics.mean(last), statistics.median(last), min(last), max(last)))
