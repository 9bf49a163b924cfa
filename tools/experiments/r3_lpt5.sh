#!/bin/bash
# NOT REPRODUCIBLE AT HEAD: the library no longer reads HISPMV_BATCH_LANES=lpt (the code of this variant was removed after the measurement recorded in
# profiles/r3_experiments/step_structure.json).  Kept as the record of what was run; refuses to run so that it cannot silently measure the default.
echo "$0: HISPMV_BATCH_LANES=lpt is not read by libhispmv.so any more -- this experiment is not reproducible at HEAD (see profiles/r3_experiments/)" >&2; exit 2
out=gpurun_out/r3za; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-verify --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*' $out/$tag.log | tr '\n' ' ')"; }
for r in 1 2 3; do
run rr_$r HISPMV_BATCH_LANES=rr
run heavy_$r X=1
done
timeout -k 10 900 python3 -m pytest tests/test_gpu_bench_set.py tests/test_gpu_parity.py -x -q > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $out/pytest.log
