#!/bin/bash
out=gpurun_out/r3zc; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.log
python3 bench.py --workload model --no-cpu-baseline --per-matrix-reps 5 --details $out/model.json > $out/model.log 2>&1
python3 - <<PY
import json
l=[x for x in open("$out/model.log") if x.startswith("{")][-1]; d=json.loads(l)
print("model:", d["ms_per_step"], d["roofline"]["frac"], d["y_checked"], [(r["name"][:6], r["kernel_us"], r["call_us_with_pcie"]) for r in d["linear_batch8"]["layers"]], [(r["name"][:6], r["us"], r["us_back_to_back"], r["plan"]) for r in json.load(open("$out/model.json"))["per_matrix"]])
PY
python3 bench.py --no-cpu-baseline --no-verify --steps 300 --warmup 100 --per-matrix-reps 0 > $out/set.log 2>&1
echo "set: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*' $out/set.log | tr '\n' ' ')"
