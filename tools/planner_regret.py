#!/usr/bin/env python
"""The planner's auto choice against a measured sweep table (tools/engine_sweep.py), on the HOST (no GPU needed): per row the
kernel `auto` picks now, its recorded time, the best recorded time, the regret.  LCE_PLAN_DEBUG=1 prints the estimates too.
usage: planner_regret.py TABLE.jsonl [-v]
       planner_regret.py --remeasure OUT.jsonl [engine_sweep.py arguments]   (needs the GPU: measures every candidate again with
                          tools/engine_sweep.py -- the constants of csrc/lce_plan_cost.cpp age with ROCm, firmware and kernels -- and
                          then prints the regret table of the fresh measurement: the round's evidence script runs this, nothing is
                          inherited from an earlier round)"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")


def choice(row):
    layer = SL.Layer(row["batch"], row["hw"], row["hw"], row["cin"], 3, 3, row["cout"], stride=row["stride"],
                     padding=SL.PADDING_SAME, pad_values=1)
    dst = {"f32": amd.F32, "i8": amd.I8, "bp": amd.BITPACKED}[row["dst"]]
    plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3))
    return plan.kernel_name()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--remeasure":
        import subprocess
        out = sys.argv[2]
        if os.path.exists(out):
            os.remove(out)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "engine_sweep.py"), out] + sys.argv[3:], check=True,
                       stdout=subprocess.DEVNULL)
        sys.argv = [sys.argv[0], out]
    lines = [json.loads(l) for l in open(sys.argv[1])]
    meta = next((l["meta"] for l in lines if "meta" in l), {})
    rows = [l for l in lines if "meta" not in l]
    if meta:
        import kernel_hash
        same = meta.get("kernel_sources_sha256") == kernel_hash.kernel_sources_hash()
        print("# table %s: %s, %s CUs, kernel sources %s (%s the tree's)" % (os.path.basename(sys.argv[1]), meta.get("device"), meta.get("compute_units"),
                                                                            str(meta.get("kernel_sources_sha256"))[:12], "=" if same else "NOT"))
    holdout = {tuple(l) for l in meta.get("holdout", [])}
    verbose = "-v" in sys.argv
    worst, bad, unknown = 0.0, 0, 0
    h_rows = h_bad = 0
    h_worst = 0.0
    for r in rows:
        name = choice(r)
        by_kernel = {}
        for c, k in r["kernel"].items():
            if c != "auto":
                by_kernel.setdefault(k, r["us"][c])
        best = min(v for c, v in r["us"].items() if c != "auto")
        us = by_kernel.get(name)
        tag = "%3dx%3dx%3d s%d b%-3d %-3s" % (r["hw"], r["cin"], r["cout"], r["stride"], r["batch"], r["dst"])
        if us is None:
            unknown += 1
            print("%s  %-48s NOT IN THE TABLE (best %.1f)" % (tag, name, best))
            continue
        regret = us / best - 1.0
        worst = max(worst, regret)
        if regret > 0.05:
            bad += 1
        if (r["hw"], r["cin"], r["cout"], r["stride"]) in holdout:
            h_rows += 1
            h_bad += regret > 0.05
            h_worst = max(h_worst, regret)
        if verbose or regret > 0.05:
            print("%s  %-48s %7.1f us  best %7.1f  regret %5.1f %%" % (tag, name, us, best, 100 * regret))
    print("rows %d: regret > 5 %% on %d, choice not in the table on %d, worst regret %.1f %%" % (len(rows), bad, unknown, 100 * worst))
    if h_rows:
        print("of which held-out layers (never used to fit the constants) %d rows: regret > 5 %% on %d, worst %.1f %%" % (h_rows, h_bad, 100 * h_worst))


if __name__ == "__main__":
    main()
