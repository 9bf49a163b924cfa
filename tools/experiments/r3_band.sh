#!/bin/bash
# round 3: band tiles -- wide-band matrices cut along the diagonal (the pessimistic family: PFlow_742 / Si41Ge41H72 as unstructured bands)
out=gpurun_out/r3ze; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "wide_band or column" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.log
for b in 0 1; do
  HISPMV_BAND_TILES=$b python3 bench.py --standin uniform --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 5 --details $out/uni_$b.json > $out/uni_$b.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/uni_$b.log") if x.startswith("{")][-1]; d=json.loads(l)
print("band tiles $b: uniform set", d["ms_per_step"], d["roofline"]["frac"], d["y_checked"], [(r["name"], r["us"], r["plan"]) for r in json.load(open("$out/uni_$b.json"))["per_matrix"] if r["name"] in ("PFlow_742","Si41Ge41H72","crankseg_2","nd6k","thread","TSOPF_RS_b2383")])
PY
done
python3 bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 100 --per-matrix-reps 0 > $out/structured.log 2>&1; echo "structured: $(grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"y_checked": [a-z]*' $out/structured.log | tr '\n' ' ')"
