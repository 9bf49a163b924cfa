#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3h; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/graph -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --preheat 0 --steps 30 --warmup 100 --per-matrix-reps 0 > $out/graph.log 2>&1
echo "== graph replay (default)"; python3 tools/trace_timeline.py $out/graph 2
