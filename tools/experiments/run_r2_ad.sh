#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2ad; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for fam in structured uniform; do
timeout -k 10 300 python3 bench.py --standin $fam --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details_$fam.json > $O/bench_$fam.log 2>&1; echo "$fam rc=$?"
python3 - $fam <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2ad/details_{sys.argv[1]}.json"))
print(sys.argv[1], d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"]: print(f'{r["name"]:16s} {r["us"]:8.1f} us (b2b {r.get("us_back_to_back", 0):6.1f}) {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
done
