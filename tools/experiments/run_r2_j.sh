#!/bin/bash
# counters of the tile-stream kernel on soc-Pokec alone
set -u
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
ARGS="--matrices soc-Pokec --steps 3 --warmup 1 --no-cpu-baseline --no-extras --per-matrix-reps 0"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 bench.py $ARGS > $O/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/r2j/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    if "tts" in k or "slices" in k:
        print(k, {c: round(x / max(1, n[(k, c)])) for c, x in v.items()})
PY
