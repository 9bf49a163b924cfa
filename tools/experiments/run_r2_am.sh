#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2am; mkdir -p $O
for cus in 256 128; do
  HISPMV_PLAN_CUS=$cus timeout -k 10 300 python3 bench.py --matrices mouse_gene --steps 30 --warmup 5 --no-cpu-baseline --no-extras --details $O/d_$cus.json > $O/b_$cus.log 2>&1
  python3 - $cus <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2am/d_{sys.argv[1]}.json"))
r=d["per_matrix"][0]
print("plan cus", sys.argv[1], "step", d["summary"]["ms_per_step"], r["us"], r.get("us_back_to_back"), r["plan"])
PY
done
