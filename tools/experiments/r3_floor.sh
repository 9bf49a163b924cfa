#!/bin/bash
# round 3: smaller tiles for the small tile streams (more, shorter latency chains behind soc-Pokec's wave)
out=gpurun_out/r3v; mkdir -p $out
for f in 24576 12288 8192 6144; do
  HISPMV_TTS_FLOOR=$f python3 bench.py --no-cpu-baseline --no-extras --no-verify --steps 300 --warmup 100 --per-matrix-reps 3 --details $out/floor_$f.json > $out/floor_$f.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/floor_$f.json"))
print("floor $f: step ms", d["summary"]["ms_per_step"], d["summary"]["roofline"]["frac"], [(r["name"], r["us"]) for r in d["per_matrix"] if r["name"] in ("nxp1","analytics","boyd2","language")])
PY
done
