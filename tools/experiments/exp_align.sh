#!/bin/bash
# experiment: slices per workgroup (stride between workgroups) vs time, PFlow_742 alone
O=gpurun_out/exp_align; mkdir -p $O
M="--matrices PFlow_742 --launch streams --streams 1 --no-cpu-baseline --steps 10 --warmup 2 --per-matrix-reps 20"
for cfg in "1 256" "1 250" "0 256" "0 251" "0 240"; do
  set -- $cfg
  HISPMV_ROW_ALIGN=$1 HISPMV_PLAN_CUS=$2 python3 bench.py $M --details $O/r.json > $O/r.log 2>&1 || { tail -3 $O/r.log; exit 1; }
  python3 - $1 $2 <<'PY'
import json,sys
d=json.load(open("gpurun_out/exp_align/r.json"))
print("align", sys.argv[1], "plan cus", sys.argv[2], [(r["name"][:6], r["us"], r["plan"]) for r in d["per_matrix"]])
PY
done
