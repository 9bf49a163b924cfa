#!/bin/bash
out=gpurun_out/r3n; mkdir -p $out
for v in xlds noxlds; do
  e=""; [ $v = noxlds ] && e="HISPMV_TTS_NO_XLDS=1"
  env $e python3 bench.py --workload model --no-cpu-baseline --per-matrix-reps 5 --details $out/model_$v.json > $out/model_$v.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/model_$v.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$v model:", d["ms_per_step"], d["roofline"]["frac"], d["y_checked"])
print("   ", d["linear_batch8"]["layers"][2])
print("   ", [(r["name"], r["us"], r["us_back_to_back"]) for r in json.load(open("$out/model_$v.json"))["per_matrix"]][2])
PY
done
