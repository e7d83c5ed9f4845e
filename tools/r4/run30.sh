#!/bin/bash
# opt-in merge-free training form on 1025-bin rows: where does the time go?  (diagnostic variants, tools/build_variants.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4aa; mkdir -p $O
AB_CALL=lg AB_FLAGS=264 AB_P=1.0 AB_B=16384 AB_N=1025 AB_SETS=4 python tools/ab_probe.py at1 at1m4 > $O/ab_1025.txt 2>&1
AB_CALL=lg AB_FLAGS=264 AB_P=1.0 AB_B=4096 AB_N=2049 AB_SETS=4 python tools/ab_probe.py at1 at1m4 >> $O/ab_1025.txt 2>&1
AB_CALL=lg AB_FLAGS=264 AB_P=1.0 AB_B=4096 AB_N=1025 AB_SETS=4 python tools/ab_probe.py at1 at1m4 >> $O/ab_1025.txt 2>&1
cat $O/ab_1025.txt
