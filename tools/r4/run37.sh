#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ah; mkdir -p $O; : > $O/f17.txt
for spec in "fwd 15 2.0" "fwd 8 1.0"; do
  set -- $spec
  echo "== $1 flags $2 p $3 16384x1025" >> $O/f17.txt
  AB_CALL=$1 AB_FLAGS=$2 AB_P=$3 AB_B=16384 AB_N=1025 AB_SETS=4 python tools/ab_probe.py f17a f17b >> $O/f17.txt 2>&1
done
cat $O/f17.txt
