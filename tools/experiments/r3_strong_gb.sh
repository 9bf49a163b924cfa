#!/bin/bash
out=gpurun_out/r3u; mkdir -p $out
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ) > $out/n1.log 2>&1; echo "n1 rc $?"
python3 - <<PY
import json
l=[x for x in open("$out/n1.log") if x.startswith("{")][-1]; d=json.loads(l)
print("main", d["ms_per_step"], d["roofline"]["frac"], d["y_checked"], "strong", d["strong_scaling"])
PY
grep real $out/n1.log
HISPMV_BENCH_REHEARSAL=1 timeout -k 10 800 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $out/reh2.log 2>&1; echo "rehearsal rc $?"
python3 - <<PY
import json
l=[x for x in open("$out/reh2.log") if x.startswith("{")][-1]; d=json.loads(l)
print("rehearsal main", d["ms_per_step"], d["y_checked"], "strong", d["strong_scaling"])
PY
