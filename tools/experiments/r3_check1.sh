#!/bin/bash
# round 3 checkpoint: whole -m gpu suite, then the driver-style and the default bench lines
out=gpurun_out/r3e; mkdir -p $out
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q --durations=15 > $out/pytest_all.log 2>&1; echo "pytest rc $?"; tail -25 $out/pytest_all.log
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.log 2>&1; echo "bench driver-style rc $?"; tail -1 $out/bench_driver.log | cut -c1-900
