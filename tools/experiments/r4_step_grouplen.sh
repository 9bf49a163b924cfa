#!/bin/bash
# round 4: longer slice groups under the step kernel (an item's start-up -- descriptor chain, first slice from HBM, window staging --
# costs 3 - 4 us whatever its length): the batch layout for resident groups below 40 / 80 / 200 slices, planned for n_cus / 2, 3, 4 workgroups
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
run h_80_2 HISPMV_BATCH_GROUP_BELOW=80
run h_80_3 HISPMV_BATCH_GROUP_BELOW=80 HISPMV_BATCH_GROUP_DIV=3
run h_80_4 HISPMV_BATCH_GROUP_BELOW=80 HISPMV_BATCH_GROUP_DIV=4
run h_200_3 HISPMV_BATCH_GROUP_BELOW=200 HISPMV_BATCH_GROUP_DIV=3
run h_200_4 HISPMV_BATCH_GROUP_BELOW=200 HISPMV_BATCH_GROUP_DIV=4
run h_80_2b HISPMV_BATCH_GROUP_BELOW=80
EXTRA="--standin uniform"
run hu_80_2 HISPMV_BATCH_GROUP_BELOW=80
run hu_80_3 HISPMV_BATCH_GROUP_BELOW=80 HISPMV_BATCH_GROUP_DIV=3
run hu_80_4 HISPMV_BATCH_GROUP_BELOW=80 HISPMV_BATCH_GROUP_DIV=4
run hu_200_4 HISPMV_BATCH_GROUP_BELOW=200 HISPMV_BATCH_GROUP_DIV=4
