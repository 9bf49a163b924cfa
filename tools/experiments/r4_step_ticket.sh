#!/bin/bash
# round 4: the next ticket drawn right behind the top-of-item barrier (its round trip under the descriptor loads) against drawn where it
# is consumed; libraries of the two commits, interleaved on one box.  Then the per-CU occupancy of the grids and the step kernel on this box.
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
P=$PWD/hispmv_amd/lib/libhispmv_prevticket.so
run t_prev HISPMV_LIB=$P
run t_new X=1
run t_prev2 HISPMV_LIB=$P
run t_new2 X=1
run t_grids HISPMV_STEP_KERNEL=0
EXTRA="--standin uniform"
run tu_prev HISPMV_LIB=$P
run tu_new X=1
EXTRA=
export HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_wgtrace.so
for o in off on; do
  if [ $o = off ]; then export HISPMV_STEP_KERNEL=0; else export HISPMV_STEP_KERNEL=1; fi
  timeout -k 10 200 python3 tools/wg_timeline.py --out $out/wgt_$o.json > $out/wgt_$o.log 2>&1; echo "wg $o rc=$?"
  python3 - <<PY
import json
d=json.load(open("$out/wgt_$o.json"))["steps"][-1]
print("$o", d["span_us"], d["cu_busy_frac"], d["gaps"]["sum_per_cu_us"], d["end_of_step"])
for k,v in d["per_kind"].items(): print("   ",k,v)
for e in d["per_entry"]:
    if e["kind"]!="slices_256t": print("      ",e["kind"],e["entry"],e["workgroups"],e["mean_us"],e["first_start_us"],e["last_end_us"])
PY
done
