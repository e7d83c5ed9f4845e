#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; rm -rf $O; mkdir -p $O
python tools/fuzz_module.py 150 777 > $O/fuzz_module.txt 2>&1; tail -n 4 $O/fuzz_module.txt | cut -c1-400
python tools/fuzz_gpu.py 120 778 > $O/fuzz_gpu.txt 2>&1; tail -n 2 $O/fuzz_gpu.txt | cut -c1-400
