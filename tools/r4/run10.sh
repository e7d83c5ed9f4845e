#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
python tools/r4/stft_sizes.py base wave2all > gpurun_out/r4j/stft_sizes.txt 2>&1
cat gpurun_out/r4j/stft_sizes.txt
