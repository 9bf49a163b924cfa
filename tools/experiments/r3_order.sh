#!/bin/bash
# round 3: launch order / stream assignment of the batch step (co-residency of 256-thread slice workgroups with 1024-thread ones)
out=gpurun_out/r3f; mkdir -p $out
python3 -m pytest tests/test_gpu_dist_full.py -x -q > $out/pytest_dist_full.log 2>&1; echo "pytest dist_full rc $?"; tail -3 $out/pytest_dist_full.log
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-verify --steps 300 --warmup 100 --per-matrix-reps 0 > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"roofline_frac": [0-9.]*' $out/$tag.log | tr '\n' ' ')"; }
run base X=1
run small_first HISPMV_BATCH_ORDER=small_first
run small_first_3 HISPMV_BATCH_ORDER=small_first HISPMV_BATCH_STREAMS=3
run base_3 HISPMV_BATCH_STREAMS=3
run tts4m HISPMV_TTS_MIN_NNZ=4000000
run tts4m_small_first HISPMV_TTS_MIN_NNZ=4000000 HISPMV_BATCH_ORDER=small_first
run tts4m_small_first_3 HISPMV_TTS_MIN_NNZ=4000000 HISPMV_BATCH_ORDER=small_first HISPMV_BATCH_STREAMS=3
