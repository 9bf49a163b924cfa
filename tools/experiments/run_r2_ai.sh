#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2ai; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_tts.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python3 bench.py --workload powerlaw --no-cpu-baseline --details $O/details_powerlaw.json > $O/bench_powerlaw.log 2>&1
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2ai/details_powerlaw.json"))
print("powerlaw", d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"]: print(f'{r["name"]:28s} {r["us"]:8.1f} us (b2b {r.get("us_back_to_back",0):6.1f}) {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
HISPMV_FORMAT=tts timeout -k 10 300 python3 bench.py --workload powerlaw --no-cpu-baseline --details $O/details_powerlaw_tts.json > $O/bench_powerlaw_tts.log 2>&1
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2ai/details_powerlaw_tts.json"))
print("powerlaw forced tts", d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"]: print(f'{r["name"]:28s} {r["us"]:8.1f} us (b2b {r.get("us_back_to_back",0):6.1f}) {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
