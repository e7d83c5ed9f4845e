#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4t; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_stft_producer.py -x -q -m gpu > $O/pytest_stft.log 2>&1; tail -n 8 $O/pytest_stft.log
