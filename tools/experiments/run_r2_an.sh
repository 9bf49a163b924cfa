#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2an; mkdir -p $O
M=TSOPF_RS_b2383,Si41Ge41H72,crankseg_2
for p in default 2; do
  if [ $p = default ]; then unset HISPMV_PLAN; else export HISPMV_PLAN=$p; fi
  timeout -k 10 300 python3 bench.py --matrices $M --steps 50 --warmup 5 --no-cpu-baseline --no-extras --details $O/d_$p.json > $O/b_$p.log 2>&1
  python3 - $p <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2an/d_{sys.argv[1]}.json"))
print("plan", sys.argv[1], "step", d["summary"]["ms_per_step"], [(r["name"][:6], r["us"], r["plan"]) for r in d["per_matrix"]])
PY
done
