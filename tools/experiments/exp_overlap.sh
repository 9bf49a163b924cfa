#!/bin/bash
# experiment: do an HBM-bound and an L2-request-bound kernel overlap when the big one leaves half a CU free?
O=gpurun_out/exp_overlap; mkdir -p $O
B="--launch streams --no-cpu-baseline --steps 20 --warmup 3 --per-matrix-reps 0"
for M in "PFlow_742,soc-Pokec" "PFlow_742,soc-Pokec,mouse_gene,TSOPF_RS_b2383,Si41Ge41H72,crankseg_2,ASIC_680k,nxp1,analytics"; do
  for P in auto 3 1; do
    for S in 1 2 4; do
      if [ $P = auto ]; then unset HISPMV_PLAN; else export HISPMV_PLAN=$P; fi
      python3 bench.py $B --matrices $M --streams $S > $O/run.log 2>&1 || { echo FAILED; tail -3 $O/run.log; exit 1; }
      python3 - "$M" $P $S <<'PY'
import json,sys
l=[x for x in open("gpurun_out/exp_overlap/run.log") if x.startswith("{")][-1]
d=json.loads(l); print(f"{len(sys.argv[1].split(',')):2d} matrices plan {sys.argv[2]:4s} streams {sys.argv[3]}: {d['ms_per_step']*1e3:7.1f} us/step")
PY
    done
  done
done
