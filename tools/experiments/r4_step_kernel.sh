#!/bin/bash
# round 4: the step kernel (one persistent workgroup per CU drawing slice groups and tiles from a queue) against the grids on two lanes,
# and the queue orders; then the per-CU occupancy of both (tools/wg_timeline.py, diagnostic library).
out=gpurun_out/r4step; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 $EXTRA > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$tag:", d["ms_per_step"], "ms", d["roofline"]["frac"], "y_checked", d.get("y_checked"), d.get("y_check",{}).get("max_backward_error"))
else: print("$tag: no line"); print(open("$out/$tag.log").read()[-1500:])
PY
}
run step X=1
run grids HISPMV_STEP_KERNEL=0
run step_lpt HISPMV_STEP_ORDER=lpt
run step_grid HISPMV_STEP_ORDER=grid
run step_2 X=1
EXTRA=--standin-uniform
run u_step X=1
run u_grids HISPMV_STEP_KERNEL=0
export HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_wgtrace.so
timeout -k 10 200 python3 tools/wg_timeline.py --out $out/wg_step.json > $out/wg_step.log 2>&1; echo "wg step rc=$?"; tail -45 $out/wg_step.log
