#!/bin/bash
# round 4, GPU call 4: column fetch for the one-sided spectra (NX > 0): parity + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r4d/pytest1.log 2>&1; echo "pytest1 rc=$?"
tail -n 3 gpurun_out/r4d/pytest1.log
ab() { # tag B N flags p call
  echo "== $1 B=$2 N=$3 flags=$4 p=$5 call=$6" >> gpurun_out/r4d/ab.txt
  AB_B=$2 AB_N=$3 AB_FLAGS=$4 AB_P=$5 AB_CALL=$6 AB_SETS=3 python tools/ab_probe.py nonx cfnx >> gpurun_out/r4d/ab.txt 2>&1
}
ab "training 1025" 16384 1025 15 2.0 lg
ab "training 1025 small" 4096 1025 15 2.0 lg
ab "training 1025 step" 1024 1025 15 2.0 lg
ab "paper fwd 1025 one-wave" 16384 1025 15 2.0 fwd
ab "paper fwd 1025 two-wave" 4096 1025 15 2.0 fwd
ab "p1 area 1025" 16384 1025 8 1.0 fwd
ab "training 257" 16384 257 15 2.0 lg
ab "paper fwd 257" 16384 257 15 2.0 fwd
ab "training 513" 8192 513 15 2.0 lg
ab "training 2049" 4096 2049 15 2.0 lg
ab "paper fwd 2049" 4096 2049 15 2.0 fwd
ab "both grads 1025" 16384 1025 15 2.0 bwdxy
cat gpurun_out/r4d/ab.txt
