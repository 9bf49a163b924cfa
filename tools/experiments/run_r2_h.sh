#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details.json > $O/bench.log 2>&1; echo "bench rc=$?"
grep -o '"ms_per_step": [0-9.]*' $O/bench.log
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2h/details.json"))
for r in d["per_matrix"]:
    if "0s/" in r["plan"]: print(f'{r["name"]:16s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
./tools/run_trace.sh h > $O/trace.log 2>&1; tail -5 $O/trace.log
