#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2y; mkdir -p $O
M=PFlow_742,TSOPF_RS_b2383,Si41Ge41H72,crankseg_2,nd6k,thread,mouse_gene
for p in default 0 1 2 3 4 5; do
  if [ $p = default ]; then unset HISPMV_PLAN; else export HISPMV_PLAN=$p; fi
  timeout -k 10 300 python3 bench.py --matrices $M --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/d_$p.json > $O/b_$p.log 2>&1; echo "plan $p rc=$?"
  python3 - $p <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2y/d_{sys.argv[1]}.json"))
print(" step", d["summary"]["ms_per_step"])
for r in d["per_matrix"]: print(f'  {r["name"]:16s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
done
