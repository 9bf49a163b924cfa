#!/bin/bash
set -u
export TMPDIR=/tmp
HISPMV_BATCH_STREAMS=1 ./tools/run_trace.sh z1 | tail -8
./tools/run_trace.sh z2 | tail -8
./tools/counters.sh r2z_pflow PFlow_742 structured | cut -c1-900
