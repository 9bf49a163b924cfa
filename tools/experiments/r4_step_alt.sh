#!/bin/bash
# round 4: the head of the step kernel's queue: cycles of t long tiles and s slice items (HISPMV_STEP_ORDER=<16*t+s>; default 1:1),
# against longest first (all long tiles first) and 1:2 / 1:3 (alt2 / alt3)
out=gpurun_out/r4step; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 $EXTRA > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$tag:", d["ms_per_step"], "ms", d["roofline"]["frac"], "y_checked", d.get("y_checked"))
else: print("$tag: no line"); print(open("$out/$tag.log").read()[-1500:])
PY
}
run b11 X=1
run b21 HISPMV_STEP_ORDER=33
run b32 HISPMV_STEP_ORDER=50
run b31 HISPMV_STEP_ORDER=49
run b43 HISPMV_STEP_ORDER=67
run b11b X=1
run b21b HISPMV_STEP_ORDER=33
EXTRA="--standin uniform"
run bu11 X=1
run bu21 HISPMV_STEP_ORDER=33
run bu32 HISPMV_STEP_ORDER=50
