#!/bin/bash
# round 4: longer resident groups (HISPMV_PLAN_RESIDENT_DIV=d[,g]): every workgroup of a slice stream stages its x window before it
# streams, and nothing hides that phase with one 1024-thread workgroup per CU -- in a batch call a matrix need not cover every CU
out=gpurun_out/r4aa; mkdir -p $out
run() { tag=$1; shift; env "$1" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 "${@:2}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
run base X=1
run div2 HISPMV_PLAN_RESIDENT_DIV=2
run div4 HISPMV_PLAN_RESIDENT_DIV=4
run div2_below80 HISPMV_PLAN_RESIDENT_DIV=2,80
run div4_below80 HISPMV_PLAN_RESIDENT_DIV=4,80
run div2_below40 HISPMV_PLAN_RESIDENT_DIV=2,40
run div4_below40 HISPMV_PLAN_RESIDENT_DIV=4,40
run base2 X=1
BIG=PFlow_742,mouse_gene,TSOPF_RS_b2383,Si41Ge41H72,crankseg_2,nd6k,thread
run slices_base X=1 --matrices $BIG
run slices_div2_below80 HISPMV_PLAN_RESIDENT_DIV=2,80 --matrices $BIG
run slices_div4_below80 HISPMV_PLAN_RESIDENT_DIV=4,80 --matrices $BIG
