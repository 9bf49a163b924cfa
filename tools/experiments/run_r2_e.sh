#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench_default.log 2>&1; echo "default rc=$?"; tail -1 $O/bench_default.log | cut -c1-3000
for w in powerlaw dense model; do
  timeout -k 10 400 python3 bench.py --workload $w --details $O/details_$w.json > $O/bench_$w.log 2>&1; echo "$w rc=$?"; tail -1 $O/bench_$w.log | cut -c1-1500
done
HISPMV_BENCH_REHEARSAL=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_reh2.log 2>&1; echo "rehearsal weak rc=$?"; tail -1 $O/bench_reh2.log | cut -c1-1200
HISPMV_BENCH_REHEARSAL=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 3 --warmup 1 --scaling strong --no-extras > $O/bench_reh2s.log 2>&1; echo "rehearsal strong rc=$?"; tail -1 $O/bench_reh2s.log | cut -c1-1200
