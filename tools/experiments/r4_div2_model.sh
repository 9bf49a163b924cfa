#!/bin/bash
# round 4: HISPMV_PLAN_RESIDENT_DIV=2,40 on the model workload, layer by layer
out=gpurun_out/r4ak; mkdir -p $out
for v in base div; do
  if [ $v = div ]; then export HISPMV_PLAN_RESIDENT_DIV=2,40; else unset HISPMV_PLAN_RESIDENT_DIV; fi
  python3 bench.py --workload model --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 10 --details $out/model_$v.json > $out/model_$v.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/model_$v.json"))
print("$v", [(r["name"], r["us"], r["us_back_to_back"], r["plan"]) for r in d["per_matrix"]], [l for l in open("$out/model_$v.log") if l.startswith("{")][-1][:0])
import re
print("$v step:", re.findall(r'"ms_per_step": [0-9.]*', open("$out/model_$v.log").read())[:1])
PY
done
