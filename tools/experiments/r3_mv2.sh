#!/bin/bash
out=gpurun_out/r3m; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_tts.py tests/test_gpu_bench_set.py tests/test_gpu_parity.py tests/test_gpu_apps.py -x -q -k "tile_stream or model_test or golden or apps or multi_matrix" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.log
python3 bench.py --workload model --no-cpu-baseline --details $out/model.json > $out/model.log 2>&1
python3 - <<PY
import json
l=[x for x in open("$out/model.log") if x.startswith("{")][-1]; d=json.loads(l)
print("model:", d["ms_per_step"], d["roofline"]["frac"], d["y_checked"])
for r in d["linear_batch8"]["layers"]: print("  ", r)
for r in json.load(open("$out/model.json"))["per_matrix"]: print("  ", r["name"], r["us"], r["us_back_to_back"], r["plan"])
PY
HISPMV_TTS_NO_XLDS=1 python3 bench.py --workload model --no-cpu-baseline --no-extras --per-matrix-reps 5 --details $out/model_noxlds.json > $out/model_noxlds.log 2>&1
echo "no xlds: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $out/model_noxlds.log | tr '\n' ' ')"
