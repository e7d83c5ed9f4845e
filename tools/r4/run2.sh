#!/bin/bash
# round 4, GPU call 2: column fetch (chunk sums from registers) A/B + parity, RT / 513 traffic, bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_row or config2 or config4 or p1_same_grid or fixtures or golden or dyadic or streams" > gpurun_out/r4b/pytest1.log 2>&1; echo "pytest1 rc=$?"
tail -3 gpurun_out/r4b/pytest1.log
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_edge.py -x -q -m gpu > gpurun_out/r4b/pytest2.log 2>&1; echo "pytest2 rc=$?"
tail -3 gpurun_out/r4b/pytest2.log
ab() { # tag B N flags p call
  echo "== $1 B=$2 N=$3 flags=$4 p=$5 call=$6" >> gpurun_out/r4b/ab.txt
  AB_B=$2 AB_N=$3 AB_FLAGS=$4 AB_P=$5 AB_CALL=$6 python tools/ab_probe.py f4 cf cfnosq >> gpurun_out/r4b/ab.txt 2>&1
}
ab "merge-free p1" 8192 2048 8 1.0 fwd
ab "paper fwd" 8192 2048 15 2.0 fwd
ab "merge p1" 8192 2048 136 1.0 fwd
ab "training form" 8192 2048 15 2.0 lg
ab "both grads" 8192 2048 15 2.0 bwdxy
ab "paper fwd 512" 32768 512 15 2.0 fwd
ab "training 512" 32768 512 15 2.0 lg
ab "paper fwd 1024" 16384 1024 15 2.0 fwd
cat gpurun_out/r4b/ab.txt
tools/r4/traffic_pass.sh r4b/traffic -:15:2.0:lg:8192:2000 -:15:2.0:lg:8192:513 -:15:2.0:lg:8192:2048 -:8:1.0:fwd:8192:2048
