#!/bin/bash
# round 4: column parts of a tile stream sized for an XCD's L2 (x of soc-Pokec: 6.5 MB against 4 MB of L2; tools/line_gather_bench
# with tables of 1 - 13 MB: a gather of 20.7 lines costs HALF when x is L2-resident) with tile shapes between the standard and
# the tall geometry: HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=rows,slots,tiles per part,zero fill[,parts]
out=gpurun_out/r4v; mkdir -p $out
one() { tag=$1; shift; env "$@" python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --steps 100 --warmup 50 --per-matrix-reps 10 --details $out/$tag.json > $out/$tag.log 2>&1
  python3 - <<PY
import json
try:
    d=json.load(open("$out/$tag.json")); r=d["per_matrix"][0]
    print("$tag:", r["us"], "us alone,", r["us_back_to_back"], "back to back; plan", r["plan"], "y_checked", d["summary"]["y_checked"])
except Exception as e: print("$tag: failed", e)
PY
}
one standard X=1
one tall HISPMV_TTS_GEOMETRY=tall
one A_8k_28k_256_fillers HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=8192,28672,256,0
one B_8k_28k_256_zerofill HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=8192,28672,256,1
one C_12k_27k_128_fillers HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=12288,27648,128,0
one D_12k_27k_128_zerofill HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=12288,27648,128,1
one E_16k_23k_128_fillers HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=16384,23552,128,0
one F_4parts_8k_28k_128_fillers HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=8192,28672,128,0,4
one G_8k_28k_128_fillers HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=8192,28672,128,0
one H_10k_28k_160_fillers HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=10240,28672,160,0
