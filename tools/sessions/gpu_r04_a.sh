#!/bin/bash
# Round 4, GPU call A: smoke, the whole GPU suite, the default bench line.   bash tools/gpu_r04_a.sh [tag]
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
echo "== pytest -m gpu"; ( time timeout 1500 python -m pytest tests -m gpu --maxfail=20 -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
echo "== bench"; ( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
LCE_TAG=$TAG python - <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/" + __import__("os").environ.get("LCE_TAG", "r04a") + "/bench.json") if l.startswith("{")][0])
except Exception as e:
    print("no bench line", e); sys.exit(0)
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["kernel"])
for k,v in d.get("extra",{}).items():
    if isinstance(v,dict):
        print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("ms","hbm_frac","kernel","timed_from","ms_one_operand_set","convolutions_only_ms","device_resident_chain_ms","error")})
PY
