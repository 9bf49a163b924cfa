#!/bin/bash
# round 4: CU-masked launch lanes (HISPMV_CU_SPLIT=<hexA>:<hexB>, 32-bit patterns repeated over the CUs): the slice grids (HBM-bound)
# on the CUs of pattern A, the tile streams (cache-bound) on the CUs of pattern B, both at once.
out=gpurun_out/r4f; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
run base X=1
run nograph HISPMV_BATCH_GRAPH=0
run split_128_128 HISPMV_CU_SPLIT=55555555:aaaaaaaa
run split_96_160 HISPMV_CU_SPLIT=25252525:dadadada
run split_64_192 HISPMV_CU_SPLIT=11111111:eeeeeeee
run split_160_96 HISPMV_CU_SPLIT=dadadada:25252525
run split_128_all HISPMV_CU_SPLIT=55555555:ffffffff
run split_96_all HISPMV_CU_SPLIT=25252525:ffffffff
run split_all_all HISPMV_CU_SPLIT=ffffffff:ffffffff
