#!/bin/bash
# round 4: resident groups of fewer than 40 slices made twice as long (HISPMV_PLAN_RESIDENT_DIV=2,40): nd6k and thread alone, the step of
# both families, and the tile floor of 32 K elements next to it
out=gpurun_out/r4aj; mkdir -p $out
one() { tag=$1; shift; env "$@" python3 bench.py --matrices nd6k,thread,crystk03,ford2 --no-cpu-baseline --no-extras --steps 100 --warmup 50 --per-matrix-reps 10 --details $out/$tag.json > $out/$tag.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/$tag.json"))
print("$tag:", [(r["name"], r["us"], r["us_back_to_back"], r["plan"]) for r in d["per_matrix"]])
PY
}
run() { tag=$1; shift; env "$1" python3 bench.py --no-cpu-baseline --no-extras --per-matrix-reps 0 --steps 300 --warmup 100 "${@:2}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
one alone_base X=1
one alone_div2_40 HISPMV_PLAN_RESIDENT_DIV=2,40
for rep in 1 2; do
run set_base_$rep X=1
run set_div2_40_$rep HISPMV_PLAN_RESIDENT_DIV=2,40
run set_div2_32_$rep HISPMV_PLAN_RESIDENT_DIV=2,32
run set_div2_48_$rep HISPMV_PLAN_RESIDENT_DIV=2,48
done
run uniform_base X=1 --standin uniform
run uniform_div2_40 HISPMV_PLAN_RESIDENT_DIV=2,40 --standin uniform
run model_base X=1 --workload model
run model_div2_40 HISPMV_PLAN_RESIDENT_DIV=2,40 --workload model
