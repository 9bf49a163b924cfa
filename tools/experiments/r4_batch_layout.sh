#!/bin/bash
# round 4: the batch layout of short-group matrices (HISPMV_BATCH_LAYOUT=1, default) against without, interleaved on one box
out=gpurun_out/r4al; mkdir -p $out
run() { tag=$1; shift; env "$1" python3 bench.py --no-cpu-baseline --no-extras --per-matrix-reps 0 --steps 300 --warmup 100 "${@:2}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
for rep in 1 2 3; do
  run off_$rep HISPMV_BATCH_LAYOUT=0
  run on_$rep HISPMV_BATCH_LAYOUT=1
done
run uniform_off HISPMV_BATCH_LAYOUT=0 --standin uniform
run uniform_on HISPMV_BATCH_LAYOUT=1 --standin uniform
run model_off HISPMV_BATCH_LAYOUT=0 --workload model
run model_on HISPMV_BATCH_LAYOUT=1 --workload model
run powerlaw_on HISPMV_BATCH_LAYOUT=1 --workload powerlaw
