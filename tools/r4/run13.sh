#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
for v in slots fwdonly; do
  SOT_LIB_PATH=$PWD/tools/ablate_libs/$v.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "config5_full_size" 2>&1 | grep -E "config 5 @256|passed|failed" | sed "s/^/$v: /"
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "config5_full_size" 2>&1 | grep -E "config 5 @256|passed|failed" | sed "s/^/product: /"
