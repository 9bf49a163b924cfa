#!/bin/bash
# round 4: even longer groups for the five short-group matrices (batch layout planned for n_cus / 4, 6, 8 workgroups)
out=gpurun_out/r4step; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_step_kernel.py -x -q > $out/div_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/div_pytest.log
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 $EXTRA > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$tag:", d["ms_per_step"], "ms", d["roofline"]["frac"], "y_checked", d.get("y_checked"), d["batch_call"]["items"])
else: print("$tag: no line"); print(open("$out/$tag.log").read()[-1500:])
PY
}
run d4 X=1
run d6 HISPMV_BATCH_GROUP_DIV=6
run d8 HISPMV_BATCH_GROUP_DIV=8
run d4b X=1
run d6b HISPMV_BATCH_GROUP_DIV=6
run d4_200 HISPMV_BATCH_GROUP_BELOW=200
run d8_200 HISPMV_BATCH_GROUP_BELOW=200 HISPMV_BATCH_GROUP_DIV=8
EXTRA="--standin uniform"
run du4 X=1
run du6 HISPMV_BATCH_GROUP_DIV=6
run du8 HISPMV_BATCH_GROUP_DIV=8
