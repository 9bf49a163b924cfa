#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_prep.py tests/test_log_schema.py -m gpu -x -q -s > $O/pytest_prep.log 2>&1; echo "pytest prep rc=$?"
tail -12 $O/pytest_prep.log
python3 examples/spmv_host.py tests/golden/syn_1138.mtx --exec_ms 20 > $O/spmv_host_sample.log 2>&1; echo "spmv_host rc=$?"; tail -25 $O/spmv_host_sample.log
timeout -k 10 300 python3 bench.py --workload powerlaw --no-cpu-baseline --details $O/details_powerlaw.json > $O/bench_powerlaw.log 2>&1; echo "powerlaw rc=$?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2f/details_powerlaw.json"))
print(d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"]: print(f'{r["name"]:28s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
timeout -k 10 900 python3 -m pytest tests/ -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -5 $O/pytest_all.log
