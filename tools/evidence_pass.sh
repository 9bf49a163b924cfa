#!/bin/bash
# The evidence of a round, in one call on the GPU box (gpurun merges gpurun_out/ back):  tools/evidence_pass.sh [r2]
#   full -m gpu suite, the rocprofv3 passes of the benchmark step (tools/profile_round.sh), kernel traces of the other
#   workloads, and the default `python bench.py` line with its per-matrix table.  Afterwards, in the build container:
#   tools/refresh_evidence.py <tag>  copies the summaries into profiles/ and regenerates the results block of DESIGN.md.
set -u
export TMPDIR=/tmp
TAG=${1:-r2}
O=gpurun_out/${TAG}m; mkdir -p $O gpurun_out/prof_${TAG}_dense gpurun_out/prof_${TAG}_model gpurun_out/prof_${TAG}_powerlaw
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -4 $O/pytest_all.log
./tools/profile_round.sh $TAG > $O/profile.log 2>&1; echo "profile rc=$?"
for w in dense model powerlaw; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$w/trace -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --per-matrix-reps 0 > gpurun_out/prof_${TAG}_$w/trace.log 2>&1 || echo "trace $w failed"
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --details $O/details_$w.json > $O/bench_$w.log 2>&1 || echo "bench $w failed"
done
timeout -k 10 500 python3 bench.py --details $O/details_default.json > $O/bench_default.log 2>&1; echo "default rc=$?"; tail -1 $O/bench_default.log | cut -c1-600
