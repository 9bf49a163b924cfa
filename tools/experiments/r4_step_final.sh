#!/bin/bash
# round 4: defaults after the step kernel (batch layout below 80 slices, up to four times as long): tests that touch it, then bench lines
out=gpurun_out/r4step; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_step_kernel.py tests/test_gpu_bench_set.py -x -q > $out/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/final_pytest.log
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 $EXTRA > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$tag:", d["ms_per_step"], "ms", d["roofline"]["frac"], "y_checked", d.get("y_checked"), d["batch_call"], d["host"])
else: print("$tag: no line"); print(open("$out/$tag.log").read()[-1500:])
PY
}
run f_def X=1
run f_grids HISPMV_STEP_KERNEL=0
run f_grids_old HISPMV_STEP_KERNEL=0 HISPMV_BATCH_GROUP_BELOW=40 HISPMV_BATCH_GROUP_DIV=2
run f_def2 X=1
EXTRA="--standin uniform"
run fu_def X=1
run fu_grids HISPMV_STEP_KERNEL=0
