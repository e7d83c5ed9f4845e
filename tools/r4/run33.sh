#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ad; mkdir -p $O
AB_CALL=fwd AB_FLAGS=15 AB_P=2.0 AB_B=4096 AB_N=4000 AB_SETS=4 python tools/ab_probe.py rtcap rtnocap > $O/ab_rt4000.txt 2>&1
AB_CALL=fwd AB_FLAGS=8 AB_P=1.0 AB_B=4096 AB_N=3500 AB_SETS=4 python tools/ab_probe.py rtcap rtnocap >> $O/ab_rt4000.txt 2>&1
cat $O/ab_rt4000.txt
