#!/bin/bash
# round 4: SELECTIVE co-residency.  Only the slice matrices whose natural window is small (<= 44 KiB) take
# 512-thread two-per-CU plans (HISPMV_PLAN_CORESIDE_KIB), PFlow_742 / mouse_gene keep their 1024-thread plans; the tile streams take
# the paired geometry (8-wavefront tiles, 78 KiB); three launch lanes so that the 512-thread slice grid starts next to the tiles.
out=gpurun_out/r4c; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 3 --details $out/$tag.json > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"
  python3 - <<PY
import json
print("   ", [(r["name"], r["us"], r["plan"]) for r in json.load(open("$out/$tag.json"))["per_matrix"][:8]])
PY
}
run base X=1
run sel44 HISPMV_PLAN_CORESIDE_KIB=44
run sel44_paired HISPMV_PLAN_CORESIDE_KIB=44 HISPMV_TTS_GEOMETRY=paired
run sel44_paired_3lanes HISPMV_PLAN_CORESIDE_KIB=44 HISPMV_TTS_GEOMETRY=paired HISPMV_BATCH_STREAMS=3
run sel44_paired_3lanes_smallfirst HISPMV_PLAN_CORESIDE_KIB=44 HISPMV_TTS_GEOMETRY=paired HISPMV_BATCH_STREAMS=3 HISPMV_BATCH_ORDER=small_first
run sel24_paired_3lanes HISPMV_PLAN_CORESIDE_KIB=24 HISPMV_TTS_GEOMETRY=paired HISPMV_BATCH_STREAMS=3
run paired_3lanes HISPMV_TTS_GEOMETRY=paired HISPMV_BATCH_STREAMS=3
