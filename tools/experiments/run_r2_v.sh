#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2v; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for fam in structured uniform; do
timeout -k 10 300 python3 bench.py --standin $fam --steps 20 --warmup 3 --no-cpu-baseline --no-extras --details $O/details_$fam.json > $O/bench_$fam.log 2>&1; echo "$fam rc=$?"
python3 - $fam <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2v/details_{sys.argv[1]}.json"))
print(sys.argv[1], d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"][:12]: print(f'{r["name"]:16s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
done
for w in powerlaw model dense; do
timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --details $O/details_$w.json > $O/bench_$w.log 2>&1
python3 - $w <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2v/details_{sys.argv[1]}.json"))
print(sys.argv[1], d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"])
for r in d["per_matrix"]: print(f'{r["name"]:28s} {r["us"]:8.1f} us {r["alg_gbs"]:8.1f} GB/s {r["plan"]}')
PY
done
