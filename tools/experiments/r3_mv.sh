#!/bin/bash
# round 3: multi-vector tile-stream launch + 6 K-element tiles for small x; regression of the set
out=gpurun_out/r3k; mkdir -p $out
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dist_full.py > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.log
python3 bench.py --workload model --no-cpu-baseline --details $out/model.json > $out/model.log 2>&1
python3 - <<PY
import json
l=[x for x in open("$out/model.log") if x.startswith("{")][-1]; d=json.loads(l)
print("model:", d["ms_per_step"], d["roofline"]["frac"], d["y_checked"])
for r in d["linear_batch8"]["layers"]: print("  ", r)
for r in json.load(open("$out/model.json"))["per_matrix"]: print("  ", r["name"], r["us"], r["us_back_to_back"], r["plan"])
PY
python3 bench.py --no-cpu-baseline --no-verify --steps 300 --warmup 100 --per-matrix-reps 0 > $out/set.log 2>&1
echo "set: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*' $out/set.log | tr '\n' ' ')"
