#!/bin/bash
# round 4: gathers and an HBM stream in ONE workgroup (tools/mix_bench.hip): do they overlap on a CU?  x of 6.5 MB (soc-Pokec:
# larger than an XCD's L2) and x of 3 / 1.6 MB (L2-resident)
out=gpurun_out/r4t; mkdir -p $out
timeout -k 10 200 ./tools/mix_bench 512 680 2>&1 | tee $out/mix_512_680.log
timeout -k 10 200 ./tools/mix_bench 512 680 3072 2>&1 | tee $out/mix_512_680_3m.log
timeout -k 10 200 ./tools/mix_bench 512 680 1600 2>&1 | tee $out/mix_512_680_1m6.log
timeout -k 10 200 ./tools/mix_bench 340 680 3072 2>&1 | tee $out/mix_340_680_3m.log
