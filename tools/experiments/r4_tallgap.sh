#!/bin/bash
# round 4: the tall tile geometry with GAP-CODED row ends (HISPMV_TTS_GEOMETRY=tallgap: rows absent from a block own no slot) against
# the standard and the tall (zero-filled staging) geometries: soc-Pokec alone, the power-law workload, and the step of the set
out=gpurun_out/r4p; mkdir -p $out
one() { tag=$1; shift; env "$@" python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --steps 100 --warmup 50 --per-matrix-reps 10 --details $out/$tag.json > $out/$tag.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/$tag.json")); r=d["per_matrix"][0]
print("$tag:", r["us"], "us alone,", r["us_back_to_back"], "back to back; plan", r["plan"], "y_checked", d["summary"]["y_checked"])
PY
}
one pokec_standard X=1
one pokec_tall HISPMV_TTS_GEOMETRY=tall
one pokec_tallgap HISPMV_TTS_GEOMETRY=tallgap
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 "${WL[@]}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
WL=(--workload powerlaw)
run powerlaw_standard X=1
run powerlaw_tallgap HISPMV_TTS_GEOMETRY=tallgap
WL=()
run set_standard X=1
run set_tallgap HISPMV_TTS_GEOMETRY=tallgap
