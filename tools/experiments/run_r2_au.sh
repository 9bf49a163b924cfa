#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2au; mkdir -p $O
for cfg in "20 3" "20 50" "20 200" "100 10" "500 20" "500 200"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b_$1_$2.log 2>&1
  echo "steps=$1 warmup=$2 $(grep -o '"ms_per_step": [0-9.]*' $O/b_$1_$2.log | head -1) $(grep -o '"frac": [0-9.]*' $O/b_$1_$2.log | head -1)"
done
