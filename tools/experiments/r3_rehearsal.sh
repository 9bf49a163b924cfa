#!/bin/bash
# round 3: the N > 1 code path of bench.py on the one-GPU box: 2 ranks on GPU 0 over gloo (numbers mean nothing), and the one-rank
# RCCL path (HISPMV_BENCH_FORCE_DIST=1) next to the plain one-rank step
out=gpurun_out/r3p; mkdir -p $out
HISPMV_BENCH_REHEARSAL=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $out/rehearsal_weak.log 2>&1; echo "rehearsal weak rc $?"; tail -1 $out/rehearsal_weak.log | cut -c1-300
HISPMV_BENCH_REHEARSAL=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 10 --warmup 3 --scaling strong --no-cpu-baseline > $out/rehearsal_strong.log 2>&1; echo "rehearsal strong rc $?"; tail -1 $out/rehearsal_strong.log | cut -c1-300
HISPMV_BENCH_FORCE_DIST=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --no-extras --per-matrix-reps 0 > $out/force_dist.log 2>&1; echo "force dist rc $?"
python3 bench.py --no-cpu-baseline --no-extras --per-matrix-reps 0 > $out/plain.log 2>&1
for f in force_dist plain rehearsal_weak rehearsal_strong; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*\|"backend": "[a-z]*"\|"ranks_seen": [0-9]*' $out/$f.log | tr '\n' ' ')"; done
