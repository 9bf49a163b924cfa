#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2r; mkdir -p $O
timeout -k 10 300 python3 bench.py --standin uniform --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details_uniform.json > $O/bench_uniform.log 2>&1; echo "uniform rc=$?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2r/details_uniform.json"))
print(d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"][:10]: print(f'{r["name"]:16s} {r["source"]:20s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
HISPMV_BATCH_STREAMS=1 ./tools/run_trace.sh r --standin uniform > $O/trace.log 2>&1; tail -5 $O/trace.log
