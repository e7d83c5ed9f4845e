#!/bin/bash
# round 4, GPU call 6: the one-wave-per-frame STFT forward kernel rebuilt around occupancy (wave2): parity with the variant library + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
SOT_LIB_PATH=$PWD/tools/ablate_libs/wave2.so python -m pytest tests/test_stft_producer.py -x -q -m gpu > gpurun_out/r4f/pytest_wave2.log 2>&1; echo "pytest wave2 rc=$?"
tail -n 4 gpurun_out/r4f/pytest_wave2.log
python tools/ab_stft.py base wave0 wave2 > gpurun_out/r4f/ab_stft.txt 2>&1
cat gpurun_out/r4f/ab_stft.txt
