#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2ac; mkdir -p $O
M=TSOPF_RS_b2383,Si41Ge41H72,crankseg_2,nd6k,thread
for p in default 2 0; do
  if [ $p = default ]; then unset HISPMV_PLAN; else export HISPMV_PLAN=$p; fi
  timeout -k 10 300 python3 bench.py --matrices $M --steps 50 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b_$p.log 2>&1
  echo "plan $p $(grep -o '"ms_per_step": [0-9.]*' $O/b_$p.log | head -1)"
done
