#!/bin/bash
# round 4: the launch-order knobs again with plain launches as the default (they were last settled under graph replay), one box
out=gpurun_out/r4ai; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --per-matrix-reps 0 --steps 300 --warmup 100 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
run base_1 X=1
run lanes_rr HISPMV_BATCH_LANES=rr
run small_first HISPMV_BATCH_ORDER=small_first
run small_first_rr HISPMV_BATCH_ORDER=small_first HISPMV_BATCH_LANES=rr
run streams3 HISPMV_BATCH_STREAMS=3
run streams1 HISPMV_BATCH_STREAMS=1
run no_fused_tail HISPMV_NO_FUSED_TAIL=1
run base_2 X=1
run div2_40 HISPMV_PLAN_RESIDENT_DIV=2,40
run div2_80 HISPMV_PLAN_RESIDENT_DIV=2,80
run no_pin HISPMV_NO_XCD_PIN=1
run lines40 HISPMV_TTS_MAX_LINES=40
run floor16k HISPMV_TTS_FLOOR=16384
run floor32k HISPMV_TTS_FLOOR=32768
run base_3 X=1
