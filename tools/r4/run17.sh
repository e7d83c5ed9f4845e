#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
python tools/r4/posgrad_debug.py > gpurun_out/r4o/debug.txt 2>&1; tail -n 30 gpurun_out/r4o/debug.txt
