#!/bin/bash
# generic (SOT_FLAG_NO_SPECIALIZE = 32) kernels of the 2048-point geometry with / without the register-step caps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ae; mkdir -p $O; : > $O/ab_generic.txt
for spec in "fwd 47 2.0" "fwd 40 1.0" "lg 47 2.0" "bwdxy 47 2.0" "bwdxy 40 1.0"; do
  set -- $spec
  echo "== $1 flags $2 p $3" >> $O/ab_generic.txt
  AB_CALL=$1 AB_FLAGS=$2 AB_P=$3 AB_B=8192 AB_N=2048 AB_SETS=4 python tools/ab_probe.py gencap gennocap >> $O/ab_generic.txt 2>&1
done
cat $O/ab_generic.txt
