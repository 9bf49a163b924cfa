#!/bin/bash
# round 3: the four segmented scans of a slice as straight-line code (independent DPP chains) when every step has a row end
out=gpurun_out/r3o; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_set.py -x -q > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.log
python3 bench.py --no-cpu-baseline --no-verify --steps 300 --warmup 100 --per-matrix-reps 5 --details $out/set.json > $out/set.log 2>&1
echo "set: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*' $out/set.log | tr '\n' ' ')"
python3 - <<PY
import json
for r in json.load(open("$out/set.json"))["per_matrix"][:9]: print("  ", r["name"], r["us"], r["us_back_to_back"], r["plan"])
PY
