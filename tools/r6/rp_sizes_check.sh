#!/bin/bash
# gpurun -- tools/r6/rp_sizes_check.sh <tag>: the per-row routes' tests, a sweep, and per-row timings by row length on the same box.
set -u
cd "${GRAFT_REPO_ROOT:?GPU box only}"
O="gpurun_out/${1:-r6rps}"; mkdir -p "$O"
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q -k "rowpos or per_row or unsorted or perm" > "$O/tests.log" 2>&1; echo "tests exit $?"; tail -n 3 "$O/tests.log"
timeout 400 python3 tools/r6/fuzz_rowpos.py 200 21 > "$O/fuzz.log" 2>&1; tail -n 4 "$O/fuzz.log" | cut -c1-300
timeout 300 python3 tools/r6/presort_sizes.py > "$O/sizes.log" 2>&1; tail -n 14 "$O/sizes.log"
