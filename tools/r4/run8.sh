#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
SOT_LIB_PATH=$PWD/tools/ablate_libs/wave2d.so python -m pytest tests/test_stft_producer.py -x -q -m gpu > gpurun_out/r4h/pytest_wave2d.log 2>&1; echo "pytest wave2d rc=$?"
tail -n 3 gpurun_out/r4h/pytest_wave2d.log
python tools/ab_stft.py base wave2c wave2d > gpurun_out/r4h/ab_stft.txt 2>&1
cat gpurun_out/r4h/ab_stft.txt
