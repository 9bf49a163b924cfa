#!/bin/bash
# NOT REPRODUCIBLE AT HEAD: the library no longer reads HISPMV_BATCH_TAIL (the code of this variant was removed after the measurement recorded in
# profiles/r3_experiments/step_structure.json).  Kept as the record of what was run; refuses to run so that it cannot silently measure the default.
echo "$0: HISPMV_BATCH_TAIL is not read by libhispmv.so any more -- this experiment is not reproducible at HEAD (see profiles/r3_experiments/)" >&2; exit 2
export TMPDIR=/tmp
out=gpurun_out/r3zd; mkdir -p $out
HISPMV_BATCH_TAIL=per_lane timeout -k 10 900 python3 -m pytest tests/test_gpu_bench_set.py tests/test_gpu_parity.py tests/test_gpu_tts.py -x -q > $out/pytest.log 2>&1; echo "pytest (per_lane) rc $?"; tail -2 $out/pytest.log
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-verify --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*' $out/$tag.log | tr '\n' ' ')"; }
for r in 1 2 3; do
run joined_$r X=1
run per_lane_$r HISPMV_BATCH_TAIL=per_lane
done
HISPMV_BATCH_TAIL=per_lane rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --preheat 0 --steps 30 --warmup 100 --per-matrix-reps 0 > $out/tr.log 2>&1
echo "== per-lane tails (graph replay)"; python3 tools/trace_timeline.py $out/tr 1
