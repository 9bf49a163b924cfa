#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2aa; mkdir -p $O
for thr in 1000000 1400000 2500000 3000000; do
for fam in structured; do
HISPMV_TTS_MIN_NNZ=$thr timeout -k 10 300 python3 bench.py --standin $fam --steps 30 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b_${thr}_$fam.log 2>&1
echo "thr=$thr $fam $(grep -o '"ms_per_step": [0-9.]*' $O/b_${thr}_$fam.log | head -1)"
done; done
HISPMV_TTS_MIN_NNZ=3000000 HISPMV_BATCH_STREAMS=1 ./tools/run_trace.sh aa3 | tail -6
