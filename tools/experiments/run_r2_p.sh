#!/bin/bash
# A/B: tile-stream size threshold x batch streams, three alternating rounds
set -u
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
for round in 1 2 3; do
  for cfg in "4194304 1" "4194304 2" "65536 1" "65536 2" "65536 3" "1000000 2"; do
    set -- $cfg
    HISPMV_TTS_MIN_NNZ=$1 HISPMV_BATCH_STREAMS=$2 timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b.log 2>&1
    echo "round $round min_nnz=$1 streams=$2 $(grep -o '"ms_per_step": [0-9.]*' $O/b.log)"
  done
done
