#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -4 $O/pytest_all.log
./tools/profile_round.sh r2 > $O/profile.log 2>&1; echo "profile rc=$?"
for w in dense model powerlaw; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2_$w/trace -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --per-matrix-reps 0 > gpurun_out/prof_r2_$w/trace.log 2>&1 || echo "trace $w failed"
done
timeout -k 10 500 python3 bench.py --details $O/details_default.json > $O/bench_default.log 2>&1; echo "default rc=$?"; tail -1 $O/bench_default.log | cut -c1-600
