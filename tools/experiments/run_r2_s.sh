#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
for fam in uniform structured; do
timeout -k 10 300 python3 bench.py --standin $fam --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details_$fam.json > $O/bench_$fam.log 2>&1; echo "$fam rc=$?"
python3 - $fam <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2s/details_{sys.argv[1]}.json"))
print(sys.argv[1], d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"][:8]: print(f'{r["name"]:16s} {r["source"]:20s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
done
