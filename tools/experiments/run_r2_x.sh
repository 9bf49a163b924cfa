#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
./tools/counters.sh r2x_pflow PFlow_742 structured
./tools/counters.sh r2x_si41 Si41Ge41H72 structured
for w in model set; do
timeout -k 10 300 python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/$w.log 2>&1
echo "$w $(grep -o '"ms_per_step": [0-9.]*' $O/$w.log | head -1)"
done
