#!/bin/bash
# round 4: where does the tile stream's time go?  soc-Pokec alone with (1) no x gathers (phase A without the cache traffic of x),
# (2) no row-order pass (phase B), against the kernel -- libraries built with -DHISPMV_TTS_EXPERIMENT=1|2 (results are wrong, only the
# time counts: --no-verify); also the tall and paired geometries.
out=gpurun_out/r4d; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --no-verify --steps 100 --warmup 50 --per-matrix-reps 10 --details $out/$tag.json > $out/$tag.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/$tag.json")); r=d["per_matrix"][0]
print("$tag:", r["us"], "us alone,", r["us_back_to_back"], "back to back; plan", r["plan"])
PY
}
run kernel X=1
run no_gathers HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_exp1.so
run no_phase_b HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_exp2.so
run tall HISPMV_TTS_GEOMETRY=tall
run tall_no_gathers HISPMV_TTS_GEOMETRY=tall HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_exp1.so
run tall_no_phase_b HISPMV_TTS_GEOMETRY=tall HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_exp2.so
