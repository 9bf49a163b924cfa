#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
for ns in 3 2 1; do
  HISPMV_BATCH_STREAMS=$ns timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/bench_s$ns.log 2>&1; echo "streams=$ns rc=$?"
  grep -o '"ms_per_step": [0-9.]*' $O/bench_s$ns.log
done
