#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3g; mkdir -p $out
for v in base small_first; do
  o=""; [ $v = small_first ] && o="HISPMV_BATCH_ORDER=small_first"
  env $o HISPMV_BATCH_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d $out/$v -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --preheat 0 --steps 30 --warmup 100 --per-matrix-reps 0 > $out/$v.log 2>&1
  echo "== $v"; python3 tools/trace_timeline.py $out/$v 2
done
