#!/bin/bash
# gpurun -- tools/r6/rp_backward_check.sh <tag>: the per-row-position routes' tests, then the per-row timings (tools/r4/perrow_time.py,
# tools/r6/perm_forward_probe.py) on the same box.
set -u
cd "${GRAFT_REPO_ROOT:?GPU box only}"
O="gpurun_out/${1:-r6rp}"; mkdir -p "$O"
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q -k "rowpos or per_row or unsorted or perm" > "$O/tests.log" 2>&1; echo "tests exit $?" | tee -a "$O/tests.log"
tail -5 "$O/tests.log"
timeout 300 python3 tools/r4/perrow_time.py > "$O/perrow.log" 2>&1; tail -30 "$O/perrow.log"
timeout 300 python3 tools/r6/perm_forward_probe.py > "$O/permfwd.log" 2>&1; tail -30 "$O/permfwd.log"
