#!/bin/bash
# kernel sequence of one bench step: tools/run_trace.sh <tag> [bench args]
set -u
export TMPDIR=/tmp
TAG=${1:-t}; shift || true
O=gpurun_out/trace_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --per-matrix-reps 0 "$@" > $O/bench.log 2>&1 || echo "trace failed"
python3 tools/trace_seq.py $O/tr | tail -40
