#!/bin/bash
# round 3: the standard tile geometry without filler words (zero-filled staging): 5.4 % fewer gathered elements on soc-Pokec
out=gpurun_out/r3t; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_tts.py -x -q -k zerofill > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $out/pytest.log
for g in standard zerofill; do
  HISPMV_TTS_GEOMETRY=$g python3 bench.py --matrices soc-Pokec,nxp1,analytics,boyd2,language --no-cpu-baseline --no-extras --steps 200 --warmup 50 --per-matrix-reps 10 --details $out/tts_$g.json > $out/tts_$g.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/tts_$g.json"))
print("$g: batch step ms", d["summary"]["ms_per_step"], [(r["name"], r["us"]) for r in d["per_matrix"]])
PY
  HISPMV_TTS_GEOMETRY=$g python3 bench.py --no-cpu-baseline --no-extras --no-verify --steps 300 --warmup 100 --per-matrix-reps 0 > $out/set_$g.log 2>&1
  echo "set $g: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $out/set_$g.log | tr '\n' ' ')"
done
