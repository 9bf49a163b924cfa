#!/bin/bash
# round 4: A/B on ONE box -- the kernels before the stray slots (libhispmv_prev.so: hispmv_kernels.hip of commit 734f568 linked with
# today's host code) against HEAD, three alternating rounds of the driver's command
out=gpurun_out/r4n; mkdir -p $out
for r in 1 2 3; do
  for v in head prev; do
    if [ $v = prev ]; then export HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_prev.so; else unset HISPMV_LIB; fi
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $out/${v}_$r.log 2>&1
    echo "$v $r: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/${v}_$r.log | tr '\n' ' ')"
  done
done
