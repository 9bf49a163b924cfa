#!/bin/bash
# NOT REPRODUCIBLE AT HEAD: the library no longer reads HISPMV_BATCH_LANES=lpt (the code of this variant was removed after the measurement recorded in
# profiles/r3_experiments/step_structure.json).  Kept as the record of what was run; refuses to run so that it cannot silently measure the default.
echo "$0: HISPMV_BATCH_LANES=lpt is not read by libhispmv.so any more -- this experiment is not reproducible at HEAD (see profiles/r3_experiments/)" >&2; exit 2
export TMPDIR=/tmp
out=gpurun_out/r3x; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-verify --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $out/$tag.log | tr '\n' ' ')"; }
run base X=1
run lpt HISPMV_BATCH_LANES=lpt
run base2 X=1
run lpt2 HISPMV_BATCH_LANES=lpt
HISPMV_BATCH_LANES=lpt rocprofv3 --kernel-trace --output-format csv -d $out/lpt_tr -- python3 bench.py --no-cpu-baseline --no-verify --no-extras --preheat 0 --steps 30 --warmup 100 --per-matrix-reps 0 > $out/lpt_tr.log 2>&1
echo "== lpt lanes, heavy lane enqueued first (graph replay)"; python3 tools/trace_timeline.py $out/lpt_tr 2
