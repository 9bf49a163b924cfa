#!/bin/bash
# round 4: prototype of the x-sweep tile (tools/xsweep_lab.hip) on a soc-Pokec-like matrix: consumer waves / window prefetch depth x window x parts
out=gpurun_out/r4b; mkdir -p $out
for cfg in "12_pd6 3072 4" "8_pd6 3072 4" "12_pd3 8192 2" "12_pd6 6144 2" "8_pd6 6144 2" "12_pd6 4096 2" "12_pd6 3072 8" "12_pd6 4096 8"; do
  set -- $cfg
  echo "=== consumer waves / prefetch $1 W $2 parts $3"
  timeout -k 10 120 ./tools/xsweep_lab_$1 $2 $3 0 2>&1 | grep -v "tile 5\|^row "
done
