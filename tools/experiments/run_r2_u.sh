#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2u; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_tts.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for fam in uniform structured; do
timeout -k 10 300 python3 bench.py --standin $fam --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details_$fam.json > $O/bench_$fam.log 2>&1; echo "$fam rc=$?"
python3 - $fam <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2u/details_{sys.argv[1]}.json"))
print(sys.argv[1], d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"]:
    if "/1ct/0%c" in r["plan"] and "1024t" in r["plan"]: print(f'{r["name"]:16s} {r["source"]:20s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
done
timeout -k 10 300 python3 bench.py --workload powerlaw --no-cpu-baseline --details $O/details_powerlaw.json > $O/bench_powerlaw.log 2>&1
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r2u/details_powerlaw.json"))
print("powerlaw", d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"]: print(f'{r["name"]:28s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
