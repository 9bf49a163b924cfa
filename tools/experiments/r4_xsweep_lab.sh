#!/bin/bash
# round 4: prototype of the x-sweep tile (tools/xsweep_lab.hip) on a soc-Pokec-like matrix: consumer waves x window x parts x slots
out=gpurun_out/r4b; mkdir -p $out
for cfg in "8 8192 2 15360" "8 8192 2 12288" "12 8192 2 15360" "6 8192 2 15360" "8 4096 4 26000" "12 4096 4 26000" "8 4096 2 15360"; do
  set -- $cfg
  echo "=== consumer waves $1 W $2 parts $3 slots $4"
  timeout -k 10 120 ./tools/xsweep_lab_$1 $2 $3 $4 2>&1 | grep -v "tile 5\|^row "
done
