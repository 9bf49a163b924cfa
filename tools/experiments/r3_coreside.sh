#!/bin/bash
# round 3: can the tile streams (cache-bound) and the slice streams (HBM-bound) share every CU?  8-wavefront tiles (paired geometry, 78 KiB)
# next to 512-thread slice workgroups (planner configuration 2: resident, window <= 48 KiB, two per CU)
out=gpurun_out/r3s; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 3 --details $out/$tag.json > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"
  python3 - <<PY
import json
print("   ", [(r["name"], r["us"], r["plan"]) for r in json.load(open("$out/$tag.json"))["per_matrix"][:8]])
PY
}
run base X=1
run plan2 HISPMV_PLAN=2
run plan2_paired HISPMV_PLAN=2 HISPMV_TTS_GEOMETRY=paired
run plan1_paired HISPMV_PLAN=1 HISPMV_TTS_GEOMETRY=paired
run plan3_paired HISPMV_PLAN=3 HISPMV_TTS_GEOMETRY=paired
