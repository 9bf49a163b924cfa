#!/bin/bash
# round 3: the paired tile-stream geometry (two 8-wavefront workgroups per CU over two column parts, zero-filled staging)
out=gpurun_out/r3d; mkdir -p $out
python3 -m pytest tests/test_gpu_tts.py -x -q -k "paired" > $out/pytest_tts.log 2>&1; echo "pytest tts rc $?"; tail -3 $out/pytest_tts.log
for g in standard paired; do
  HISPMV_TTS_GEOMETRY=$g python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --steps 100 --warmup 50 \
     --per-matrix-reps 10 --details $out/pokec_$g.json > $out/pokec_$g.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/pokec_$g.json"))
r=d["per_matrix"][0]
print("$g:", r["name"], r["us"], "us alone", r["us_back_to_back"], "us back to back; batch step ms", d["summary"]["ms_per_step"], r["plan"], flush=True)
PY
done
HISPMV_TTS_GEOMETRY=paired HISPMV_NO_XCD_PIN=1 python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --steps 100 --warmup 50 --per-matrix-reps 0 > $out/pokec_paired_nopin.log 2>&1; echo nopin; grep -o '"ms_per_step": [0-9.]*' $out/pokec_paired_nopin.log | head -1
