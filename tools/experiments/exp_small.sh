#!/bin/bash
# experiment: the 12 small/mid matrices under different carry modes and plans
set -u
O=gpurun_out/exp_small; mkdir -p $O
SM=ASIC_680k,nxp1,analytics,boyd2,language,crystk03,trans5,ford2,lowThrust_7,c-52,hangGlider_3,poli_large,thread,nd6k
M="--matrices $SM --launch streams --no-cpu-baseline --streams 1 --steps 20 --warmup 3 --per-matrix-reps 20"
python3 bench.py $M --details $O/auto.json > $O/auto.log 2>&1 && \
HISPMV_CARRY=fixup python3 bench.py $M --details $O/fixup.json > $O/fixup.log 2>&1 && \
HISPMV_CARRY=lookback python3 bench.py $M --details $O/lookback.json > $O/lookback.log 2>&1 && \
HISPMV_PLAN=global python3 bench.py $M --details $O/global.json > $O/global.log 2>&1 && \
HISPMV_PLAN=0 python3 bench.py $M --details $O/plan0.json > $O/plan0.log 2>&1 && \
HISPMV_PLAN=1 python3 bench.py $M --details $O/plan1.json > $O/plan1.log 2>&1
python3 - <<'PY'
import json
tags=["auto","fixup","lookback","global","plan0","plan1"]
D={}
for t in tags:
    try: D[t]=json.load(open(f"gpurun_out/exp_small/{t}.json"))
    except Exception as e: print(t,"missing",e)
names=[r["name"] for r in D["auto"]["per_matrix"]]
print(f'{"":16s}'+"".join(f"{t:>10s}" for t in D))
for i,n in enumerate(names):
    print(f"{n:16s}"+"".join(f'{D[t]["per_matrix"][i]["us"]:10.1f}' for t in D)+"  "+D["auto"]["per_matrix"][i]["plan"])
print(f'{"step us":16s}'+"".join(f'{D[t]["summary"]["ms_per_step"]*1e3:10.1f}' for t in D))
PY
