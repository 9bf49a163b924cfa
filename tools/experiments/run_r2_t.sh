#!/bin/bash
# counters of the tile-stream kernel where its gathers are nearly free (PFlow_742, unstructured band: 1.6 lines per gather)
set -u
export TMPDIR=/tmp
O=gpurun_out/r2t; mkdir -p $O
ARGS="--standin uniform --matrices PFlow_742 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --per-matrix-reps 0"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 bench.py $ARGS > $O/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r2t/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for f in glob.glob("gpurun_out/r2t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    if "tts" in k:
        print(k, "dur us", sorted(dur[k])[len(dur[k]) // 2], {c: round(x / max(1, n[(k, c)])) for c, x in v.items()})
PY
