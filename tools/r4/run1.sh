#!/bin/bash
# round 4, GPU call 1: parity of the staged gradient stores + fresh-position path, A/B timing, traffic counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "paper_row_lengths or config5 or full_row_backward or runtime_length or training" > gpurun_out/r4a/pytest1.log 2>&1; echo "pytest1 rc=$?" 
tail -3 gpurun_out/r4a/pytest1.log
python -m pytest tests/test_gpu_edge.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r4a/pytest2.log 2>&1; echo "pytest2 rc=$?"
tail -3 gpurun_out/r4a/pytest2.log
for shape in "16384 1025" "4096 1025" "16384 257" "8192 513" "4096 2049" "32768 129"; do
  set -- $shape
  echo "== lg B=$1 N=$2" >> gpurun_out/r4a/ab.txt
  AB_B=$1 AB_N=$2 AB_FLAGS=15 AB_P=2.0 AB_CALL=lg AB_SETS=3 python tools/ab_probe.py old new >> gpurun_out/r4a/ab.txt 2>&1
done
echo "== bwdxy B=16384 N=1025" >> gpurun_out/r4a/ab.txt
AB_B=16384 AB_N=1025 AB_FLAGS=15 AB_P=2.0 AB_CALL=bwdxy AB_SETS=3 python tools/ab_probe.py old new >> gpurun_out/r4a/ab.txt 2>&1
cat gpurun_out/r4a/ab.txt
tools/r4/traffic_pass.sh r4a/traffic old:15:2.0:lg:16384:1025 new:15:2.0:lg:16384:1025 new:15:2.0:bwdxy:16384:1025 old:15:2.0:lg:65536:257 new:15:2.0:lg:65536:257
