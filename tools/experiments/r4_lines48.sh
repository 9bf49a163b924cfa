#!/bin/bash
# round 4: HISPMV_TTS_MAX_LINES (lines of x a 64-element gather of a tile stream may touch: 32 -> 48) on the other workloads
out=gpurun_out/r4s; mkdir -p $out
run() { tag=$1; shift; env "$1" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 "${@:2}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
for L in 32 48 40 56; do
run uniform_$L HISPMV_TTS_MAX_LINES=$L --standin uniform
run set_$L HISPMV_TTS_MAX_LINES=$L
done
run powerlaw_32 HISPMV_TTS_MAX_LINES=32 --workload powerlaw
run powerlaw_48 HISPMV_TTS_MAX_LINES=48 --workload powerlaw
run model_32 HISPMV_TTS_MAX_LINES=32 --workload model
run model_48 HISPMV_TTS_MAX_LINES=48 --workload model
