#!/bin/bash
# round 3: what bounds the tile stream on soc-Pokec -- TCP / SQ counters for the standard and the tall geometry
# (a pass with TA_* counters hung rocprofv3 on the box -- incomplete dispatch --: none here)
set -u
export TMPDIR=/tmp
out=gpurun_out/r3c; mkdir -p $out
rocprofv3 -L > $out/counters_list.txt 2>&1
ARGS="--matrices soc-Pokec --steps 3 --warmup 1 --no-cpu-baseline --no-extras --per-matrix-reps 0"
for g in standard tall; do
  O=$out/$g; mkdir -p $O
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    HISPMV_TTS_GEOMETRY=$g rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 bench.py $ARGS > $O/$tag.log 2>&1 || echo "pass $g $tag failed"
  done
  python3 - $O <<'PY'
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
json.dump(out, open(O + "/counters.json", "w"), indent=1)
print(O, json.dumps(out))
PY
done
