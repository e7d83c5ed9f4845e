#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
SOT_LIB_PATH=$PWD/tools/ablate_libs/wave2c.so python -m pytest tests/test_stft_producer.py -x -q -m gpu > gpurun_out/r4g/pytest_wave2c.log 2>&1; echo "pytest wave2c rc=$?"
tail -n 3 gpurun_out/r4g/pytest_wave2c.log
SOT_LIB_PATH=$PWD/tools/ablate_libs/base3.so python -m pytest tests/test_stft_producer.py -x -q -m gpu > gpurun_out/r4g/pytest_base3.log 2>&1; echo "pytest base3 (asm cmul in the slot kernels) rc=$?"
tail -n 3 gpurun_out/r4g/pytest_base3.log
python tools/ab_stft.py base base3 wave2c > gpurun_out/r4g/ab_stft.txt 2>&1
cat gpurun_out/r4g/ab_stft.txt
