#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3j; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-verify --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $out/$tag.log | tr '\n' ' ')"; }
run graph X=1
run plain HISPMV_BATCH_GRAPH=0
run graph2 X=1
run plain2 HISPMV_BATCH_GRAPH=0
HISPMV_BATCH_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d $out/plain_tr -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --preheat 0 --steps 30 --warmup 100 --per-matrix-reps 0 > $out/plain_tr.log 2>&1
echo "== plain launches"; python3 tools/trace_timeline.py $out/plain_tr 2
