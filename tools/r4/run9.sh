#!/bin/bash
cd $GRAFT_REPO_ROOT
export SOT_LIB_PATH=$PWD/tools/ablate_libs/wave2d.so
tools/pmc_stft.sh > /dev/null 2>&1
mkdir -p gpurun_out/r4i; cp gpurun_out/pmc_stft.txt gpurun_out/r4i/pmc_stft_wave2d.txt
cat gpurun_out/pmc_stft.txt
