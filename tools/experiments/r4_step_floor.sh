#!/bin/bash
# round 4: the tile floor of the small tile streams (HISPMV_TTS_FLOOR, elements per row tile) under the step kernel -- in round 3 smaller
# tiles were faster alone and slower in the step of grids; then the per-CU occupancy of the final defaults, both families
out=gpurun_out/r4step; mkdir -p $out
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 $EXTRA > $out/$tag.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$tag:", d["ms_per_step"], "ms", d["roofline"]["frac"], "y_checked", d.get("y_checked"), d["batch_call"]["items"])
else: print("$tag: no line"); print(open("$out/$tag.log").read()[-1500:])
PY
}
run fl_def X=1
run fl_12k HISPMV_TTS_FLOOR=12288
run fl_48k HISPMV_TTS_FLOOR=49152
run fl_96k HISPMV_TTS_FLOOR=98304
run fl_def2 X=1
EXTRA="--standin uniform"
run flu_def X=1
run flu_48k HISPMV_TTS_FLOOR=49152
EXTRA=
export HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_wgtrace.so
for f in structured uniform; do U=; [ $f = uniform ] && U=--uniform
  timeout -k 10 200 python3 tools/wg_timeline.py $U --out $out/wg_final_$f.json > $out/wg_final_$f.log 2>&1
  HISPMV_STEP_KERNEL=0 timeout -k 10 200 python3 tools/wg_timeline.py $U --out $out/wg_final_grids_$f.json > $out/wg_final_grids_$f.log 2>&1
  python3 - <<PY
import json
for t in ("wg_final_$f","wg_final_grids_$f"):
    d=json.load(open("$out/"+t+".json"))["steps"][-1]
    print(t, d["span_us"], d["cu_busy_frac"], d["gaps"]["sum_per_cu_us"], d["end_of_step"]["mean_idle_before_end_us"], d["workgroups"])
    for k,v in d["per_kind"].items(): print("   ",k,v)
PY
done
