#!/bin/bash
# gpurun -- tools/r6/campaign.sh <tag> [seconds per sweep = 300]: the randomised sweeps on the final library, one after the other (fresh seeds, not the suite's).
set -u
cd "${GRAFT_REPO_ROOT:?GPU box only}"
O="gpurun_out/${1:-r6camp}"; S="${2:-300}"; mkdir -p "$O"
python3 tools/fuzz_gpu.py "$S" 601 > "$O/fuzz_gpu.log" 2>&1; tail -n 1 "$O/fuzz_gpu.log"
python3 tools/r6/fuzz_rowpos.py "$S" 602 > "$O/fuzz_rowpos.log" 2>&1; tail -n 1 "$O/fuzz_rowpos.log"
python3 tools/fuzz_stft.py "$S" 603 > "$O/fuzz_stft.log" 2>&1; tail -n 1 "$O/fuzz_stft.log"
python3 tools/fuzz_module.py "$S" 604 > "$O/fuzz_module.log" 2>&1; tail -n 1 "$O/fuzz_module.log"
python3 tools/r5/fuzz_mss.py "$S" 605 > "$O/fuzz_mss.log" 2>&1; tail -n 1 "$O/fuzz_mss.log"
