#!/bin/bash
# round 4: the stray split (tile_kind 3) IN THE STEP: the set with its eight mesh-origin matrices perturbed (HISPMV_BENCH_VARIANT), split on / off
out=gpurun_out/r4i; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 3 --details $out/$tag.json > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"
  python3 - <<PY
import json
print("   ", [(r["name"], r["us"], r["plan"]) for r in json.load(open("$out/$tag.json"))["per_matrix"] if r["name"] in ("PFlow_742","TSOPF_RS_b2383","Si41Ge41H72","crankseg_2","nd6k","thread")])
PY
}
for v in stray2 stray5 stray10; do
run ${v}_nosplit HISPMV_BENCH_VARIANT=$v HISPMV_STRAY_SPLIT=0
run ${v}_split HISPMV_BENCH_VARIANT=$v HISPMV_STRAY_SPLIT=1
done
