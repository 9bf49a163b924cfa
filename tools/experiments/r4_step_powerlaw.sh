#!/bin/bash
# round 4: the powerlaw workload (two tile streams) under the step kernel against the tile-stream grid, on one box
out=gpurun_out/r4step; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --workload powerlaw --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$tag:", d["ms_per_step"], "ms", d["roofline"]["frac"], "y_checked", d.get("y_checked"), d.get("batch_call"))
else: print("$tag: no line"); print(open("$out/$tag.log").read()[-1500:])
PY
}
run p_grids HISPMV_STEP_KERNEL=0
run p_step X=1
run p_step_grid HISPMV_STEP_ORDER=grid
run p_grids2 HISPMV_STEP_KERNEL=0
export HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_wgtrace.so
for o in off on; do
  if [ $o = off ]; then export HISPMV_STEP_KERNEL=0; else export HISPMV_STEP_KERNEL=1; fi
  timeout -k 10 200 python3 tools/wg_timeline.py --powerlaw --out $out/wgp_$o.json > $out/wgp_$o.log 2>&1; echo "wg $o rc=$?"
  python3 - <<PY
import json
d=json.load(open("$out/wgp_$o.json"))["steps"][-1]
print("$o", d["span_us"], d["cu_busy_frac"], d["gaps"], d["end_of_step"], d["workgroups_per_cu"])
for k,v in d["per_kind"].items(): print("   ",k,v)
for e in d["per_entry"]: print("      ",e)
PY
done
