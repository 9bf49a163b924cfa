#!/bin/bash
# round 4: prototype of the x-sweep tile (tools/xsweep_lab.hip) on a soc-Pokec-like matrix, flag-synchronised producers / consumers:
# consumer waves _ ring windows x window floats x column parts; the whole kernel, then each side alone
out=gpurun_out/r4b; mkdir -p $out
for cfg in "12_r6 3072 2" "12_r4 2048 4" "8_r6 3072 2"; do
  set -- $cfg
  echo "=== consumer waves _ ring $1 W $2 parts $3"
  timeout -k 10 100 ./tools/xsweep_lab_$1 $2 $3 2>&1 | grep -v "^row "
done
