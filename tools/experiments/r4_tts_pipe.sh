#!/bin/bash
# (variant = tools/experiments/r4_tts_pipe.patch; libhispmv_pipe4.so built from it, libhispmv_nopipe.so = HEAD; the recorded run also had a build with phase B not unrolled)
# round 4: the x gathers of a wavefront's first slice of block b+1 issued BEFORE its row-order pass of block b (HISPMV_TTS_PIPE, tile
# stream kernel): libraries built with -DHISPMV_TTS_PIPE=1 -DHISPMV_TTS_B_UNROLL=4|1 and -DHISPMV_TTS_PIPE=0, same box
out=gpurun_out/r4y; mkdir -p $out
one() { tag=$1; shift; env "$@" python3 bench.py --matrices soc-Pokec --no-cpu-baseline --no-extras --steps 100 --warmup 50 --per-matrix-reps 10 --details $out/$tag.json > $out/$tag.log 2>&1
  python3 - <<PY
import json
try:
    d=json.load(open("$out/$tag.json")); r=d["per_matrix"][0]
    print("$tag:", r["us"], "us alone,", r["us_back_to_back"], "back to back; plan", r["plan"], "y_checked", d["summary"]["y_checked"])
except Exception as e: print("$tag: failed", e)
PY
}
run() { tag=$1; shift; env "$1" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 "${@:2}" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
L=$PWD/hispmv_amd/lib
timeout -k 10 300 python3 -m pytest tests/test_gpu_tts.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for v in nopipe pipe4; do one pokec_$v HISPMV_LIB=$L/libhispmv_$v.so; done
for v in nopipe pipe4; do run set_$v HISPMV_LIB=$L/libhispmv_$v.so; done
for v in nopipe pipe4; do run powerlaw_$v HISPMV_LIB=$L/libhispmv_$v.so --workload powerlaw; done
for v in nopipe pipe4; do run uniform_$v HISPMV_LIB=$L/libhispmv_$v.so --standin uniform; done
