#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_set.py tests/test_gpu_dist.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --details $O/details.json > $O/bench.log 2>&1; echo "bench rc=$?"
tail -2 $O/bench.log | cut -c1-900
