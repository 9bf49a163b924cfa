#!/bin/bash
# round 3: the half-LDS tile geometry for the small tile streams only (their own launch), byte-weighted lanes
out=gpurun_out/r3r; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-verify --no-extras --steps 300 --warmup 100 --per-matrix-reps 3 --details $out/$tag.json > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $out/$tag.log | tr '\n' ' ')"
  python3 - <<PY
import json
print("   ", [(r["name"], r["us"], r["plan"]) for r in json.load(open("$out/$tag.json"))["per_matrix"] if r["name"] in ("nxp1","analytics","boyd2","language")])
PY
}
run base X=1
run small HISPMV_TTS_SMALL=1
run small_lpt HISPMV_TTS_SMALL=1 HISPMV_BATCH_LANES=lpt
run lpt HISPMV_BATCH_LANES=lpt
