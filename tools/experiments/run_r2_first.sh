#!/bin/bash
# round 2, first GPU call: new parity tests on the current kernels, the accumulator-tile lab, a baseline bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
( timeout -k 10 300 ./tools/tile_lab all > $O/tile_lab.log 2>&1; echo "tile_lab rc=$?" ) 
tail -40 $O/tile_lab.log
timeout -k 10 800 python3 -m pytest tests/test_gpu_bench_set.py tests/test_gpu_parity.py -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?"
tail -3 $O/bench.log
