#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2n; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -4 $O/pytest_all.log
timeout -k 10 300 python3 bench.py --workload powerlaw --no-cpu-baseline --details $O/details_powerlaw.json > $O/bench_powerlaw.log 2>&1; echo "powerlaw rc=$?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2n/details_powerlaw.json"))
print(d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"]: print(f'{r["name"]:28s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
