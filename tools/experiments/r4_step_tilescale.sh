#!/bin/bash
# round 4: the price of a tile in the step kernel's queue (the step of the set ends with small tiles: are they priced too low?)
out=gpurun_out/r4step; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 $EXTRA > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$tag:", d["ms_per_step"], "ms", d["roofline"]["frac"], "y_checked", d.get("y_checked"), d["batch_call"]["items"])
else: print("$tag: no line"); print(open("$out/$tag.log").read()[-1500:])
PY
}
run ts_10 X=1
run ts_125 HISPMV_STEP_TILE_SCALE=1.25
run ts_15 HISPMV_STEP_TILE_SCALE=1.5
run ts_08 HISPMV_STEP_TILE_SCALE=0.8
run ts_10b X=1
run ts_c256_4 HISPMV_STEP_COST256=4
run ts_c256_16 HISPMV_STEP_COST256=16
EXTRA="--standin uniform"
run tsu_10 X=1
run tsu_125 HISPMV_STEP_TILE_SCALE=1.25
run tsu_15 HISPMV_STEP_TILE_SCALE=1.5
