#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2ah; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_set.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for w in dense model; do
timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --no-extras --details $O/details_$w.json > $O/bench_$w.log 2>&1
python3 - $w <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2ah/details_{sys.argv[1]}.json"))
print(sys.argv[1], d["summary"]["ms_per_step"], d["summary"]["value"], d["summary"]["roofline"]["frac"])
PY
done
