#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r1 [extra bench.py args]
# 1. --kernel-trace --stats          -> per-kernel durations (must agree with bench.py's HIP-event time)
# 2. --pmc FETCH_SIZE                -> HBM read traffic   (separate passes: TCC has 4 slots, FETCH_SIZE
# 3. --pmc WRITE_SIZE                -> HBM write traffic   costs 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md)
# Counters are never combined with tracing domains other than --kernel-trace.
set -u
TAG=${1:-r1}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --preheat 0 --no-verify --no-cpu-baseline --no-extras --per-matrix-reps 0 $*"   # bench.py defaults (--launch batch); passes over the set are read from its JSON line
# (the trace pass runs long enough for the chip to reach its steady clocks, like bench.py's default: 200 warm-up steps)
TRACE_ARGS="--steps 30 --warmup 200 --preheat 0 --no-verify --no-cpu-baseline --no-extras --per-matrix-reps 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $TRACE_ARGS > "$OUT/trace.log" 2>&1 || echo "trace pass failed"
# 1b. the same step on ONE stream, plain launches (no side stream, no graph): kernel durations that do not overlap, from which the
#     per-kernel roofline fraction is recomputed (bench.py's JSON line in the log carries the algorithmic bytes of every grid)
HISPMV_BATCH_STREAMS=1 HISPMV_BATCH_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_single" -- python3 bench.py $TRACE_ARGS > "$OUT/trace_single.log" 2>&1 || echo "single-stream trace pass failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS > "$OUT/fetch.log" 2>&1 || echo "fetch pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS > "$OUT/write.log" 2>&1 || echo "write pass failed"
find "$OUT" -name '*.csv' | head -20
