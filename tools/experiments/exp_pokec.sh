#!/bin/bash
# experiment: soc-Pokec / mouse_gene / PFlow_742 alone, plus L2 hit counters of soc-Pokec
# (the non-temporal stream loads this script first compared against are now the only variant)
set -u
export TMPDIR=/tmp
O=gpurun_out/exp_pokec; mkdir -p $O
M="--matrices soc-Pokec,mouse_gene,PFlow_742 --no-cpu-baseline --launch streams --streams 1 --steps 10 --warmup 2"
python3 bench.py $M --details $O/base.json > $O/base.log 2>&1 && \
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- python3 bench.py --matrices soc-Pokec --no-cpu-baseline --streams 1 --steps 3 --warmup 1 --per-matrix-reps 0 > $O/tcc.log 2>&1
python3 - <<'PY'
import json,glob,csv,collections
for t in ("base",):
    d=json.load(open(f"gpurun_out/exp_pokec/{t}.json"))
    print(t, [(r["name"], round(r["us"],1)) for r in d["per_matrix"]] if "us" in d["per_matrix"][0] else d["per_matrix"])
for f in glob.glob("gpurun_out/exp_pokec/tcc/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Kernel_Name"][:60]]+=1
    for k,v in acc.items(): print(k, dict(v), n[k])
PY
