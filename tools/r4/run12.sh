#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
SOT_LIB_PATH=$PWD/tools/ablate_libs/swz.so python -m pytest tests/test_stft_producer.py -x -q -m gpu > gpurun_out/r4l/pytest_swz.log 2>&1; echo "pytest swz rc=$?"
tail -n 3 gpurun_out/r4l/pytest_swz.log
python tools/r4/stft_chain.py noswz swz > gpurun_out/r4l/chain.txt 2>&1
cat gpurun_out/r4l/chain.txt
