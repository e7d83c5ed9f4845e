#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of tools/bench_matrix.py (all shapes, forward and backward).
# Usage: tools/profile_shapes.sh <tag>   -> gpurun_out/prof_shapes_<tag>/summary.txt
set -u
TAG=${1:-r1}
OUT=gpurun_out/prof_shapes_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_matrix.py > $OUT/matrix.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/matrix.log
