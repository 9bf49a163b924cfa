#!/bin/bash
# round 4: where the four-group items of the 256-thread plans sit in the step kernel's queue (their price per slice), on one box
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
run e_c2 HISPMV_STEP_COST256=2
run e_c45 X=1
run e_c8 HISPMV_STEP_COST256=8
run e_c20 HISPMV_STEP_COST256=20
run e_c2b HISPMV_STEP_COST256=2
run e_grids HISPMV_STEP_KERNEL=0
EXTRA="--standin uniform"
run eu_c2 HISPMV_STEP_COST256=2
run eu_c45 X=1
run eu_c8 HISPMV_STEP_COST256=8
EXTRA=
export HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_wgtrace.so
timeout -k 10 200 python3 tools/wg_timeline.py --out $out/wg_c45.json > $out/wg_c45.log 2>&1; echo "wg rc=$?"
python3 - <<PY
import json
d=json.load(open("$out/wg_c45.json"))["steps"][-1]
print(d["span_us"], d["cu_busy_frac"], d["gaps"]["sum_per_cu_us"], d["end_of_step"])
for k,v in d["per_kind"].items(): print("   ",k,v)
PY
