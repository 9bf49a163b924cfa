#!/bin/bash
# AddressSanitizer + UBSan over the host side of the library (reader, COO->CSR, slice stream, launch planner, device
# layout, transposed tile stream, format / tiling choice, the order of the step kernel's queue).
# GPU sanitizers are not available on the test pool; the device side is covered by the bit-exact parity tests.
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/hispmv_sanitize_host
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -std=c++17 -Ihispmv_amd/csrc \
    tools/sanitize/host_main.cpp hispmv_amd/csrc/hispmv_prep.cpp hispmv_amd/csrc/hispmv_plan.cpp hispmv_amd/csrc/hispmv_tts.cpp hispmv_amd/csrc/hispmv_choose.cpp -o "$OUT"
HISPMV_MTX_CHUNK_BYTES=48 OMP_NUM_THREADS=4 ASAN_OPTIONS=detect_leaks=0 "$OUT" tests/golden/*.mtx
rm -f "$OUT"
