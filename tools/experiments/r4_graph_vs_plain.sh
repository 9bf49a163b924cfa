#!/bin/bash
# round 4: the two-lane step replayed as a HIP graph against plain launches (HISPMV_BATCH_GRAPH=0), interleaved repeats on one box
out=gpurun_out/r4ag; mkdir -p $out
run() { tag=$1; shift; env "$1" python3 bench.py --no-cpu-baseline --no-extras --per-matrix-reps 0 "${@:2}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
for rep in 1 2 3; do
  run graph_$rep HISPMV_BATCH_GRAPH=1 --steps 300 --warmup 100
  run plain_$rep HISPMV_BATCH_GRAPH=0 --steps 300 --warmup 100
done
for rep in 1 2; do
  run driver_graph_$rep HISPMV_BATCH_GRAPH=1 --gpus 1 --steps 20 --warmup 5
  run driver_plain_$rep HISPMV_BATCH_GRAPH=0 --gpus 1 --steps 20 --warmup 5
done
run uniform_graph HISPMV_BATCH_GRAPH=1 --steps 300 --warmup 100 --standin uniform
run uniform_plain HISPMV_BATCH_GRAPH=0 --steps 300 --warmup 100 --standin uniform
run powerlaw_graph HISPMV_BATCH_GRAPH=1 --steps 300 --warmup 100 --workload powerlaw
run powerlaw_plain HISPMV_BATCH_GRAPH=0 --steps 300 --warmup 100 --workload powerlaw
run default_500_graph HISPMV_BATCH_GRAPH=1
run default_500_plain HISPMV_BATCH_GRAPH=0
