#!/bin/bash
# round 4: is the cost of a line in a gather the vector cache's fill rate or L2 MISS traffic?  tools/line_gather_bench with
# tables of x from 1 MB (inside an XCD's 4 MB L2) to 13 MB, and tools/mix_bench (gathers + stream) at 1.5 MB
out=gpurun_out/r4u; mkdir -p $out
for mb in 1 2 3 4 6 13; do
  echo "=== table $mb MB"; timeout -k 10 120 ./tools/line_gather_bench $((mb * 1048576 + (mb == 6 ? 524288 : 0))) 2>&1 | grep -v "line loads"
done
