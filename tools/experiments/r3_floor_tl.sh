#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3w; mkdir -p $out
for f in 24576 12288; do
  HISPMV_TTS_FLOOR=$f rocprofv3 --kernel-trace --output-format csv -d $out/f$f -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --preheat 0 --steps 30 --warmup 100 --per-matrix-reps 0 > $out/f$f.log 2>&1
  echo "== floor $f (graph replay)"; python3 tools/trace_timeline.py $out/f$f 2
done
