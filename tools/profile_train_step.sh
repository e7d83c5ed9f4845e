#!/bin/bash
# rocprofv3 kernel-trace stats of tools/bench_train_step.py (config 5: STFT producer + SOT forward/backward).
# Usage: tools/profile_train_step.sh <tag>   -> gpurun_out/prof_train_<tag>/summary.txt
set -u
TAG=${1:-r1}
OUT=gpurun_out/prof_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_train_step.py > $OUT/step.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/step.log
