#!/usr/bin/env python
"""Re-fit the constants of the planner's cost estimate (compute-engine_amd/csrc/lce_plan_cost.cpp, LCE_COST_TABLE) to measured sweeps.

The estimate prices every candidate kernel of a 3x3 layer in microseconds; its constants are fitted quantities that age with ROCm,
firmware and the kernels (round-5 review, item 6).  This tool is the procedure, so that a re-fit is one command and not archaeology:

    python tools/fit_cost.py profiles/r06/engine_sweep_r06_box*.jsonl [--fit NAME,NAME,...] [--iters N] [--holdout]

It runs on the HOST: the sweeps hold the measured time of every candidate (tools/engine_sweep.py), the host-simulation library
(tests/hostsim, built with -DLCE_COST_TUNABLE) holds the REAL planner with its constants as variables.  For every (layer, batch, output
type, candidate) it asks that planner for the candidate's estimate and compares it with the mean measured time; coordinate descent
moves the chosen constants (multiplicative steps) to minimise

    loss = mean over candidates of log(estimate / measured)^2  +  4 * mean over rows of regret(row)^2

(the second term is what matters in the end: the time of the candidate the estimate ranks first against the best measured one).
Rows of the held-out layers (engine_sweep.HOLDOUT) never enter the loss; they are reported.  The result is printed as the lines to
paste into LCE_COST_TABLE; nothing is written to the sources."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hostsim_lib as H  # noqa: E402  (ctypes front end of the host simulation: test tooling, not the oracle)
import oracle_lib as O  # noqa: E402  (only its ConvSpec, to fill the C descriptor)

ENGINE = {"direct": (3, 0, -1), "mfma": (2, 0, -1), "stream": (5, 0, -1), "wstream": (6, 0, -1)}
for r in (2, 4, 7, 8, 14):
    ENGINE["stream_il%d" % r] = (5, r, 1)
DST = {"f32": O.DST_F32, "i8": O.DST_I8, "bp": O.DST_BITPACKED}


def load(paths):
    acc, holdout = {}, set()
    for path in paths:
        for line in open(path):
            r = json.loads(line)
            if "meta" in r:
                holdout |= {tuple(l) for l in r["meta"].get("holdout", [])}
                continue
            key = (r["hw"], r["cin"], r["cout"], r["stride"], r["batch"], r["dst"])
            for cand, name in r["kernel"].items():
                if cand != "auto":
                    acc.setdefault(key, {}).setdefault((cand, name), []).append(r["us"][cand])
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}, holdout


class Planner:
    def __init__(self):
        self.lib = H.lib()
        self.lib.hostsim_cost_name.restype = C.c_char_p
        self.lib.hostsim_cost_get.restype = C.c_double
        self.lib.hostsim_plan_estimate.restype = C.c_double
        self.n = self.lib.hostsim_cost_count()
        self.names = [self.lib.hostsim_cost_name(i).decode() for i in range(self.n)]

    def get(self, name):
        return self.lib.hostsim_cost_get(self.names.index(name))

    def set(self, name, v):
        self.lib.hostsim_cost_set(self.names.index(name), C.c_double(v))

    def estimate(self, key, cand):
        hw, cin, cout, stride, batch, dst = key
        eng, rows, il = ENGINE[cand]
        spec = O.ConvSpec(batch, hw, hw, cin, 3, 3, cout, 1, stride, stride, 1, 1, O.PADDING_SAME, 1, O.ACT_NONE, O.SEM_OPTIMIZED)
        desc = H.make_desc(spec, DST[dst], 0.125, 3)
        name = C.create_string_buffer(128)
        us = self.lib.hostsim_plan_estimate(C.byref(desc), eng, rows, il, 256, name, 128)
        return us, name.value.decode()


def evaluate(P, table, holdout, verbose=False):
    """(fit loss, rows, rows over 5 %, worst regret) on the fitting rows, the same on the held-out rows"""
    out = {}
    for part in ("fit", "holdout"):
        errs, regrets = [], []
        for key, cands in table.items():
            if ((key[:4] in holdout) != (part == "holdout")):
                continue
            priced = []
            for (cand, name), t in cands.items():
                if cand not in ENGINE:
                    continue
                e, got = P.estimate(key, cand)
                if e <= 0 or got != name:       # (the forced option resolves to another kernel on the host than it did on the box)
                    continue
                errs.append((e / t, key, cand))
                priced.append((e, t, name))
            if len(priced) >= 2:
                pick = min(priced)[1]
                regrets.append((pick / min(t for _, t, _ in priced) - 1.0, key, min(priced)[2]))
        import math
        fit = sum(math.log(r) ** 2 for r, _, _ in errs) / max(1, len(errs))
        reg = sum(r * r for r, _, _ in regrets) / max(1, len(regrets))
        out[part] = dict(loss=fit + 4.0 * reg, candidates=len(errs), rows=len(regrets), over=sum(r > 0.05 for r, _, _ in regrets),
                         worst=max([r for r, _, _ in regrets] or [0.0]), rms_log=fit ** 0.5)
        if verbose:
            for r, key, name in sorted(regrets, reverse=True)[:12]:
                if r > 0.03:
                    print("   %-5s %s -> %s: regret %.1f %%" % (part, key, name, 100 * r))
    return out


def main():
    args = sys.argv[1:]
    paths = [a for a in args if not a.startswith("--")]
    opt = {a.split("=")[0]: (a.split("=") + [""])[1] for a in args if a.startswith("--")}
    table, holdout = load(paths)
    P = Planner()
    base = {n: P.get(n) for n in P.names}
    fit_names = [n for n in opt.get("--fit", "").split(",") if n] or [n for n in P.names if n.startswith(("kWs", "kSt", "kGemm"))]
    iters = int(opt.get("--iters", "3"))
    e0 = evaluate(P, table, holdout, verbose=True)
    print("before: fit rows %(rows)d (%(candidates)d candidates): rms log error %(rms_log).3f, rows over 5 %%: %(over)d, worst %(worst).3f" % e0["fit"])
    print("        held-out rows %(rows)d: rows over 5 %%: %(over)d, worst %(worst).3f" % e0["holdout"])
    best = e0["fit"]["loss"]
    for it in range(iters):
        improved = False
        for n in fit_names:
            v0 = P.get(n)
            for f in (1.25, 0.8, 1.1, 0.91, 1.04, 0.96):
                P.set(n, v0 * f)
                l = evaluate(P, table, holdout)["fit"]["loss"]
                if l < best * 0.999:
                    best, v0, improved = l, v0 * f, True
                P.set(n, v0)
        print("iteration %d: loss %.5f" % (it + 1, best), flush=True)
        if not improved:
            break
    e1 = evaluate(P, table, holdout, verbose=True)
    print("after:  fit rows %(rows)d: rms log error %(rms_log).3f, rows over 5 %%: %(over)d, worst %(worst).3f" % e1["fit"])
    print("        held-out rows %(rows)d: rows over 5 %%: %(over)d, worst %(worst).3f" % e1["holdout"])
    print("constants that moved (paste into LCE_COST_TABLE, csrc/lce_plan_cost.cpp):")
    for n in P.names:
        if abs(P.get(n) / base[n] - 1.0) > 1e-9:
            print("  %-24s %12.6g  ->  %12.6g" % (n, base[n], P.get(n)))


if __name__ == "__main__":
    main()
