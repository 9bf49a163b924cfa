#!/bin/bash
# round 4: the host-vector path (run_kernel / linear from host buffers): y written straight into pinned memory against the copy back
out=gpurun_out/r4ab; mkdir -p $out
for v in direct copy direct copy; do
  HISPMV_HOST_Y=$v python3 bench.py --workload model --no-cpu-baseline --steps 100 --warmup 50 --per-matrix-reps 0 > $out/model_$v.log 2>&1
  python3 - $out/model_$v.log $v <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[2], "one vector:", [(l["name"], l["kernel_us"], l["call_us_with_pcie"]) for l in d["host_vector_call"]["layers"]])
        print(sys.argv[2], "8 vectors: ", [(l["name"], l["kernel_us"], l["call_us_with_pcie"]) for l in d["linear_batch8"]["layers"]])
PY
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_apps.py tests/test_gpu_tts.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
