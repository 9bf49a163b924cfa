#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2q; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
HISPMV_BENCH_REHEARSAL=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_reh2.log 2>&1; echo "rehearsal rc=$?"; tail -1 $O/bench_reh2.log | cut -c1-400
