#!/bin/bash
# SQ/TCC counters of the hispmv kernels on one matrix of the benchmark set, launched alone:
#   tools/counters.sh <tag> <matrix> [structured|uniform]     -> gpurun_out/<tag>/counters.json
# (rocprofv3 --pmc passes of <= 4 counters, each with --kernel-trace only; see MI355X_MICROARCH.md)
set -u
export TMPDIR=/tmp
TAG=$1; MAT=$2; FAM=${3:-structured}
O=gpurun_out/$TAG; mkdir -p $O
ARGS="--matrices $MAT --standin $FAM --steps 3 --warmup 1 --no-cpu-baseline --no-extras --per-matrix-reps 0"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 bench.py $ARGS > $O/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - $O $MAT $FAM <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, v in acc.items():
    if "hispmv" in k and ("slices" in k or "tts" in k):
        out[k] = {"median_us_under_pmc": sorted(dur[k])[len(dur[k]) // 2], **{c: round(x / max(1, n[(k, c)])) for c, x in v.items()}}
json.dump({"matrix": sys.argv[2], "family": sys.argv[3], "kernels": out}, open(O + "/counters.json", "w"), indent=1)
print(json.dumps(out))
PY
