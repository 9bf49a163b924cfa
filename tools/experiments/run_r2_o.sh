#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2o; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_tts.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for ns in 2 1; do
HISPMV_BATCH_STREAMS=$ns timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details_s$ns.json > $O/bench_s$ns.log 2>&1; echo "bench streams=$ns rc=$?"
grep -o '"ms_per_step": [0-9.]*' $O/bench_s$ns.log
done
HISPMV_BATCH_STREAMS=1 ./tools/run_trace.sh o > $O/trace.log 2>&1; tail -5 $O/trace.log
