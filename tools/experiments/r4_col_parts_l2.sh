#!/bin/bash
# round 4: L2 hits and misses of the tile-stream kernel on soc-Pokec, standard geometry against two column parts pinned to XCD
# halves (r4_col_parts.sh shape A) and the same parts unpinned: does an XCD's L2 hold its half of x?
export TMPDIR=/tmp
out=gpurun_out/r4x; mkdir -p $out
ARGS="--matrices soc-Pokec --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --per-matrix-reps 0"
pass() { tag=$1; shift
  env "$@" true
  ( export "$@"; rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/$tag -- python3 bench.py $ARGS > $out/$tag.log 2>&1 ) || echo "pass $tag failed"
  python3 - $out/$tag $tag <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    if "tts" in k: print(sys.argv[2], k[-40:], {c: round(x / max(1, n[(k, c)])) for c, x in v.items()})
PY
}
pass standard X=1
pass A_pinned HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=8192,28672,256,0
pass A_unpinned HISPMV_TTS_GEOMETRY=tall HISPMV_TTS_TALL_SHAPE=8192,28672,256,0 HISPMV_NO_XCD_PIN=1
pass tall HISPMV_TTS_GEOMETRY=tall
