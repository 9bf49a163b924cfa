#!/bin/bash
# round 4: one tail launch per lane, on the lane's own stream (HISPMV_LANE_TAILS=1), against the one tail behind the join
out=gpurun_out/r4af; mkdir -p $out
run() { tag=$1; shift; env "$1" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 "${@:2}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*\|"y_max_backward_error": [0-9.e-]*' $out/$tag.log | tr '\n' ' ')"; }
for rep in 1 2 3; do
  run base_$rep X=1
  run lane_tails_$rep HISPMV_LANE_TAILS=1
done
run uniform_base X=1 --standin uniform
run uniform_lane_tails HISPMV_LANE_TAILS=1 --standin uniform
run plain_base HISPMV_BATCH_GRAPH=0
HISPMV_BATCH_GRAPH=0 HISPMV_LANE_TAILS=1 python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 > $out/plain_lane_tails.log 2>&1
echo "plain_lane_tails: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*' $out/plain_lane_tails.log | tr '\n' ' ')"
