#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_tts.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log
for i in 1 2; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details.json > $O/bench$i.log 2>&1; echo "bench rc=$?"
grep -o '"ms_per_step": [0-9.]*' $O/bench$i.log
done
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2l/details.json"))
for r in d["per_matrix"][:8]: print(f'{r["name"]:16s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
HISPMV_BATCH_STREAMS=1 ./tools/run_trace.sh l > $O/trace.log 2>&1; tail -5 $O/trace.log
