#!/bin/bash
out=gpurun_out/r3q; mkdir -p $out
for v in graph eager; do
  g=1; [ $v = eager ] && g=0
  HISPMV_BENCH_STEP_GRAPH=$g HISPMV_BENCH_FORCE_DIST=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $out/force_dist_$v.log 2>&1; echo "force dist $v rc $?"
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*\|"rank_step": "[a-z ]*"' $out/force_dist_$v.log | tr '\n' ' ')"; grep "step graph not used" $out/force_dist_$v.log | head -2
done
python3 bench.py --no-cpu-baseline --no-extras --per-matrix-reps 0 > $out/plain.log 2>&1
echo "plain: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $out/plain.log | tr '\n' ' ')"
