#!/bin/bash
# round 3: the tall tile-stream geometry (two XCD-pinned column parts of 16 K-row tiles, zero-filled staging)
out=gpurun_out/r3b; mkdir -p $out
python3 -m pytest tests/test_gpu_tts.py -x -q > $out/pytest_tts.log 2>&1; echo "pytest tts rc $?"; tail -3 $out/pytest_tts.log
for g in standard tall; do
  HISPMV_TTS_GEOMETRY=$g python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --steps 100 --warmup 50 \
     --per-matrix-reps 10 --details $out/pokec_$g.json > $out/pokec_$g.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/pokec_$g.json"))
r=d["per_matrix"][0]
print("$g:", r["name"], r["us"], "us alone", r["us_back_to_back"], "us back to back; batch step ms", d["summary"]["ms_per_step"], r["plan"], flush=True)
PY
done
HISPMV_TTS_GEOMETRY=tall HISPMV_NO_XCD_PIN=1 python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --steps 100 --warmup 50 --per-matrix-reps 0 > $out/pokec_tall_nopin.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $out/pokec_tall_nopin.log | head -1
for g in standard auto; do
  HISPMV_TTS_GEOMETRY=$g python3 bench.py --no-cpu-baseline --steps 300 --warmup 100 --per-matrix-reps 0 > $out/set_$g.log 2>&1
  echo "set $g:"; grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*' $out/set_$g.log | tr '\n' ' '; echo
done
HISPMV_TTS_GEOMETRY=auto python3 bench.py --workload powerlaw --no-cpu-baseline --steps 200 --warmup 50 --details $out/powerlaw_auto.json > $out/powerlaw_auto.log 2>&1
echo "powerlaw auto:"; grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $out/powerlaw_auto.log | tr '\n' ' '; echo
