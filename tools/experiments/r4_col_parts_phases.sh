#!/bin/bash
# round 4: the column-part tile shapes of r4_col_parts.sh split into phases (libraries built with -DHISPMV_TTS_EXPERIMENT=1: no x
# gathers, =2: no row-order pass), with and without the XCD pinning of the parts
out=gpurun_out/r4w; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --no-verify --steps 100 --warmup 50 --per-matrix-reps 10 --details $out/$tag.json > $out/$tag.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$out/$tag.json")); r=d["per_matrix"][0]
print("$tag:", r["us"], "us alone,", r["us_back_to_back"], "back to back; plan", r["plan"])
PY
}
E1=HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_exp1.so
E2=HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_exp2.so
A="HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=8192,28672,256,0"
run standard X=1
run standard_no_gathers $E1
run A $A
run A_no_gathers $A $E1
run A_no_phase_b $A $E2
run A_unpinned $A HISPMV_NO_XCD_PIN=1
run A_unpinned_no_gathers $A HISPMV_NO_XCD_PIN=1 $E1
run tall HISPMV_TTS_GEOMETRY=tall
run tall_unpinned HISPMV_TTS_GEOMETRY=tall HISPMV_NO_XCD_PIN=1
