#!/bin/bash
# NOT REPRODUCIBLE AT HEAD: the library no longer reads HISPMV_TTS_AUX (the code of this variant was removed after the measurement recorded in
# profiles/r3_experiments/tts_soc_pokec.json).  Kept as the record of what was run; refuses to run so that it cannot silently measure the default.
echo "$0: HISPMV_TTS_AUX is not read by libhispmv.so any more -- this experiment is not reproducible at HEAD (see profiles/r3_experiments/)" >&2; exit 2
# round 3: cache policy of the tile stream's x gathers (buffer_load aux bits: 1 = sc0, 2 = nt, 16 = sc1) on soc-Pokec
set -e
out=gpurun_out/r3a; mkdir -p $out
for a in 0 1 2 16 17 3 18 19; do
  HISPMV_TTS_AUX=$a python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --steps 50 --warmup 20 \
     --per-matrix-reps 10 --details $out/aux_$a.json > $out/aux_$a.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/aux_$a.json"))
r=d["per_matrix"][0]
print("aux $a:", r["name"], r["us"], "us alone", r["us_back_to_back"], "us back to back; batch step", d["summary"]["ms_per_step"], flush=True)
PY
done
