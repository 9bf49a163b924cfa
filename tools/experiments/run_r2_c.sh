#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for pin in 1 0; do
  if [ $pin = 0 ]; then export HISPMV_NO_XCD_PIN=1; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --per-matrix-reps 0 > $O/bench_pin$pin.log 2>&1; echo "bench pin=$pin rc=$?"
  grep -o '"ms_per_step": [0-9.]*' $O/bench_pin$pin.log
done
unset HISPMV_NO_XCD_PIN
./tools/run_trace.sh pin > $O/trace_pin.log 2>&1; tail -14 $O/trace_pin.log
