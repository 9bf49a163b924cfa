#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2aq; mkdir -p $O
for rep in 1 2; do for g in 0 1; do
  if [ $g = 1 ]; then export HISPMV_BATCH_GRAPH=1; else unset HISPMV_BATCH_GRAPH; fi
  timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b_$g.log 2>&1
  echo "graph=$g rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/b_$g.log | head -1)"
done; done
export HISPMV_BATCH_GRAPH=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_bench_set.py -m gpu -x -q -k "batch or set" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
