#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2ap; mkdir -p $O
for rep in 1 2; do for s in 2 3 1; do
  HISPMV_BATCH_STREAMS=$s timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b_$s.log 2>&1
  echo "streams=$s $(grep -o '"ms_per_step": [0-9.]*' $O/b_$s.log | head -1)"
done; done
