#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_tts.py -m gpu -x -q > $O/pytest_tts.log 2>&1; echo "pytest tts rc=$?"
tail -25 $O/pytest_tts.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details.json > $O/bench.log 2>&1; echo "bench rc=$?"
grep -o '"ms_per_step": [0-9.]*' $O/bench.log
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2g/details.json"))
for r in d["per_matrix"]: print(f'{r["name"]:16s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
./tools/run_trace.sh g > $O/trace.log 2>&1; tail -6 $O/trace.log
