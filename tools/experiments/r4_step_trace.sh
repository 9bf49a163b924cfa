#!/bin/bash
# round 4: the step kernel under its queue orders against the grids on two lanes, ON ONE BOX: bench lines, then per-CU occupancy
# (tools/wg_timeline.py, diagnostic library)
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
run d_grids HISPMV_STEP_KERNEL=0
run d_alt X=1
run d_lpt HISPMV_STEP_ORDER=lpt
run d_grid HISPMV_STEP_ORDER=grid
run d_grids2 HISPMV_STEP_KERNEL=0
run d_alt2 X=1
EXTRA="--standin uniform"
run du_grids HISPMV_STEP_KERNEL=0
run du_alt X=1
run du_lpt HISPMV_STEP_ORDER=lpt
EXTRA=
export HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_wgtrace.so
for o in alt; do
  export HISPMV_STEP_KERNEL=1
  timeout -k 10 200 python3 tools/wg_timeline.py --out $out/wg_$o.json > $out/wg_$o.log 2>&1; echo "wg $o rc=$?"
  python3 - <<PY
import json
d=json.load(open("$out/wg_$o.json"))["steps"][-1]
print("$o", d["span_us"], d["cu_busy_frac"], d["gaps"]["sum_per_cu_us"], d["end_of_step"])
for k,v in d["per_kind"].items(): print("   ",k,v)
for e in d["per_entry"]:
    if e["kind"]!="slices_256t": print("      ",e["kind"],e["entry"],e["workgroups"],e["mean_us"],e["first_start_us"],e["last_end_us"])
PY
done
