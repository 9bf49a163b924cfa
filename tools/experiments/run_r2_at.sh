#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2at; mkdir -p $O
for rep in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b_$rep.log 2>&1
  echo "rep=$rep $(grep -o '"ms_per_step": [0-9.]*' $O/b_$rep.log | head -1) $(grep -o '"frac": [0-9.]*' $O/b_$rep.log | head -1)"
done
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
