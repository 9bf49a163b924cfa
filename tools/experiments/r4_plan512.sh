#!/bin/bash
# round 4: 512-thread two-per-CU plans for the slice matrices with small windows (HISPMV_PLAN_CORESIDE_KIB, the knob of the selective
# co-residency experiment) WITHOUT paired tiles: two workgroups per CU overlap one's window staging with the other's streaming.
# Interleaved repeats on one box.
out=gpurun_out/r4ae; mkdir -p $out
run() { tag=$1; shift; env "$1" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 "${@:2}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
for rep in 1 2 3; do
  run base_$rep X=1
  run kib16_$rep HISPMV_PLAN_CORESIDE_KIB=16
  run kib24_$rep HISPMV_PLAN_CORESIDE_KIB=24
  run kib44_$rep HISPMV_PLAN_CORESIDE_KIB=44
done
run uniform_base X=1 --standin uniform
run uniform_kib24 HISPMV_PLAN_CORESIDE_KIB=24 --standin uniform
run uniform_kib44 HISPMV_PLAN_CORESIDE_KIB=44 --standin uniform
