#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2ao; mkdir -p $O
for cus in 128 192 256 320 512; do
  HISPMV_PLAN_CUS=$cus timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b_$cus.log 2>&1
  echo "plan_cus=$cus $(grep -o '"ms_per_step": [0-9.]*' $O/b_$cus.log | head -1)"
done
