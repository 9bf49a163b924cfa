#!/bin/bash
# The evidence of a round on the GPU box (gpurun merges gpurun_out/ back), in two calls (a call is limited to 20 minutes):
#   tools/evidence_pass.sh r3 tests     full -m gpu suite + the default `python bench.py` line with its per-matrix table + the driver-style line
#   tools/evidence_pass.sh r3 profiles  the rocprofv3 passes of the benchmark step (tools/profile_round.sh) + kernel traces and bench lines of the other workloads
# Afterwards, in the build container: tools/refresh_evidence.py <tag> copies the summaries into profiles/ and regenerates the
# results block of DESIGN.md.
set -u
export TMPDIR=/tmp
TAG=${1:-r3}; WHAT=${2:-tests}
O=gpurun_out/${TAG}m; mkdir -p $O gpurun_out/prof_${TAG}_dense gpurun_out/prof_${TAG}_model gpurun_out/prof_${TAG}_powerlaw
if [ "$WHAT" = tests ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -14 $O/pytest_all.log
  python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
  timeout -k 10 400 python3 bench.py --details $O/details_default.json > $O/bench_default.log 2>&1; echo "default rc=$?"; tail -1 $O/bench_default.log | cut -c1-400
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.log 2>&1; echo "driver-style rc=$?"; tail -1 $O/bench_driver_style.log | cut -c1-400
else
  ./tools/profile_round.sh $TAG > $O/profile.log 2>&1; echo "profile rc=$?"
  for w in dense model powerlaw; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$w/trace -- python3 bench.py --workload $w --steps 5 --warmup 1 --preheat 0 --no-verify --no-cpu-baseline --per-matrix-reps 0 > gpurun_out/prof_${TAG}_$w/trace.log 2>&1 || echo "trace $w failed"
    timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --details $O/details_$w.json > $O/bench_$w.log 2>&1 || echo "bench $w failed"
    echo "$w: $(tail -1 $O/bench_$w.log | cut -c1-300)"
  done
fi
