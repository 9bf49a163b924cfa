#!/bin/bash
# round 4: should the SMALLEST slice matrices keep their short groups under the step kernel (many 5 - 8 us items to end the queue with)?
# HISPMV_BATCH_MIN_SLICES: matrices with fewer slices get no long-group layout (nd6k 6.7 K, thread 4.3 K, crankseg_2 13.8 K, Si41Ge41H72 14.7 K, TSOPF 15.8 K)
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
run ms_512 X=1
run ms_5k HISPMV_BATCH_MIN_SLICES=5000
run ms_8k HISPMV_BATCH_MIN_SLICES=8192
run ms_512b X=1
run ms_8kb HISPMV_BATCH_MIN_SLICES=8192
run ms_8k_c4 HISPMV_BATCH_MIN_SLICES=8192 HISPMV_STEP_COST256=4
EXTRA="--standin uniform"
run msu_512 X=1
run msu_5k HISPMV_BATCH_MIN_SLICES=5000
run msu_8k HISPMV_BATCH_MIN_SLICES=8192
