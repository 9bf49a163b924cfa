#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2w; mkdir -p $O
for f in auto slices; do for s in 1 2; do
HISPMV_FORMAT=$f HISPMV_BATCH_STREAMS=$s timeout -k 10 300 python3 bench.py --workload model --steps 50 --warmup 5 --no-cpu-baseline --per-matrix-reps 0 > $O/model_${f}_$s.log 2>&1
echo "model format=$f streams=$s $(grep -o '"ms_per_step": [0-9.]*' $O/model_${f}_$s.log | head -1)"
done; done
for s in 1 2; do
HISPMV_BATCH_STREAMS=$s timeout -k 10 300 python3 bench.py --workload dense --steps 50 --warmup 5 --no-cpu-baseline --per-matrix-reps 0 > $O/dense_$s.log 2>&1
echo "dense streams=$s $(grep -o '"ms_per_step": [0-9.]*' $O/dense_$s.log | head -1)"
HISPMV_BATCH_STREAMS=$s timeout -k 10 300 python3 bench.py --workload powerlaw --steps 20 --warmup 3 --no-cpu-baseline --per-matrix-reps 0 > $O/pl_$s.log 2>&1
echo "powerlaw streams=$s $(grep -o '"ms_per_step": [0-9.]*' $O/pl_$s.log | head -1)"
done
