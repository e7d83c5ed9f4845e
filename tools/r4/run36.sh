#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ag; mkdir -p $O; : > $O/rp.txt
for v in rp3 rp4 rp3 rp4; do echo "== $v" >> $O/rp.txt; SOT_LIB_PATH=$PWD/tools/ablate_libs/$v.so python tools/r4/perrow_time.py 2>&1 | grep -v amdgpu | head -2 >> $O/rp.txt; done
cat $O/rp.txt
