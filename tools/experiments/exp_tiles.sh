#!/bin/bash
# experiment: column-tile budget for soc-Pokec (x = 6.5 MB)
O=gpurun_out/exp_tiles; mkdir -p $O
for T in 1500000 2200000 3300000 4194304 7000000; do
  HISPMV_COL_TILE_BYTES=$T python3 bench.py --matrices soc-Pokec --launch streams --no-cpu-baseline --streams 1 --steps 10 --warmup 2 --per-matrix-reps 10 --details $O/r.json > $O/r.log 2>&1 || { tail -3 $O/r.log; exit 1; }
  python3 - $T <<'PY'
import json,sys
d=json.load(open("gpurun_out/exp_tiles/r.json"))
print("tile bytes", sys.argv[1], [(r["name"], r["us"], r["plan"]) for r in d["per_matrix"]])
PY
done
