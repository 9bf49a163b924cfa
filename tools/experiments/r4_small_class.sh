#!/bin/bash
# round 4 (verdict item 7): knobs of the small-matrix class measured INSIDE the step of the set (the step is the sum of its
# kernels' CU-time: r4_step_subsets.sh), not alone
out=gpurun_out/r4r; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
run base X=1
run lines48 HISPMV_TTS_MAX_LINES=48
run lines64 HISPMV_TTS_MAX_LINES=64
run min256k HISPMV_TTS_MIN_NNZ=262144
run min256k_lines64 HISPMV_TTS_MIN_NNZ=262144 HISPMV_TTS_MAX_LINES=64
run min64k_lines64 HISPMV_TTS_MIN_NNZ=65536 HISPMV_TTS_MAX_LINES=64
run floor48k HISPMV_TTS_FLOOR=49152
run floor96k HISPMV_TTS_FLOOR=98304
run floor12k HISPMV_TTS_FLOOR=12288
run floor48k_lines64 HISPMV_TTS_FLOOR=49152 HISPMV_TTS_MAX_LINES=64
run floor96k_lines64_min256k HISPMV_TTS_FLOOR=98304 HISPMV_TTS_MAX_LINES=64 HISPMV_TTS_MIN_NNZ=262144
run small HISPMV_TTS_SMALL=1
run base2 X=1
