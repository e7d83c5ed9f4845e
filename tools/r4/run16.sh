#!/bin/bash
# position-gradient kernel: its tests + the existing position-gradient tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
timeout 900 python -m pytest tests/test_gpu_edge.py -x -q -k "position" > gpurun_out/r4o/pytest_posgrad.log 2>&1; tail -n 15 gpurun_out/r4o/pytest_posgrad.log
