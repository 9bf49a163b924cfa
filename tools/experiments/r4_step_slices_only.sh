#!/bin/bash
# round 4: is the slice path slower inside the step kernel (40 - 68 SGPR spill reloads per slice iteration that the multi-matrix kernel
# does not have)?  Five slice matrices alone, same groups (no batch layout), grids against the step kernel in grid order.
out=gpurun_out/r4step; mkdir -p $out
M=PFlow_742,mouse_gene,TSOPF_RS_b2383,Si41Ge41H72,crankseg_2
run() { tag=$1; shift; env "$@" python3 bench.py --matrices $M --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$tag:", d["ms_per_step"], "ms", d["roofline"]["frac"], "y_checked", d.get("y_checked"), d["batch_call"])
else: print("$tag: no line"); print(open("$out/$tag.log").read()[-1500:])
PY
}
export HISPMV_BATCH_LAYOUT=0
run so_grids HISPMV_STEP_KERNEL=0
run so_grids1 HISPMV_STEP_KERNEL=0 HISPMV_BATCH_STREAMS=1
run so_step_grid HISPMV_STEP_ORDER=grid
run so_step X=1
run so_grids_b HISPMV_STEP_KERNEL=0
run so_step_b X=1
unset HISPMV_BATCH_LAYOUT
run so_step_long X=1
