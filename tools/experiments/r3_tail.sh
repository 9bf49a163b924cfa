#!/bin/bash
# round 3: fused tail launch (fix-up + merge) and byte-weighted lanes with the heaviest on the caller's stream
export TMPDIR=/tmp
out=gpurun_out/r3i; mkdir -p $out
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dist_full.py > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.log
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-verify --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*' $out/$tag.log | tr '\n' ' ')"; }
run new X=1
run nofuse HISPMV_NO_FUSED_TAIL=1
rocprofv3 --kernel-trace --output-format csv -d $out/graph -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --preheat 0 --steps 30 --warmup 100 --per-matrix-reps 0 > $out/graph.log 2>&1
echo "== graph replay (default)"; python3 tools/trace_timeline.py $out/graph 2
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/driver.log 2>&1; echo "driver-style: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*\|"y_checked": [a-z]*' $out/driver.log | tr '\n' ' ')"
