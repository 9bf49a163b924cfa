#!/bin/bash
# round 3: the 1024 x 8192 d=0.25 layer of apps/model_test.py as a slice stream with its whole x (32 KiB) in one LDS window
out=gpurun_out/r3zb; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --workload model --no-cpu-baseline --per-matrix-reps 5 --details $out/$tag.json > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$tag model:", d["ms_per_step"], d["roofline"]["frac"], d["y_checked"], [(r["name"][:6], r["kernel_us"]) for r in d["linear_batch8"]["layers"]], [(r["name"][:6], r["us"], r["us_back_to_back"], r["plan"]) for r in json.load(open("$out/$tag.json"))["per_matrix"]])
PY
}
run base X=1
run slices_auto HISPMV_FORMAT=slices
run slices_res8 HISPMV_FORMAT=slices HISPMV_PLAN=2 HISPMV_PLAN_MIN_RESIDENT=4
run slices_res8_1024 HISPMV_FORMAT=slices HISPMV_PLAN=5 HISPMV_PLAN_MIN_RESIDENT=4
