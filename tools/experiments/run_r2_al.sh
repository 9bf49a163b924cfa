#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2al; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_tts.py tests/test_gpu_bench_set.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $O/b.log 2>&1; echo "set $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1)"
