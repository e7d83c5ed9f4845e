#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4af; rm -rf $O; mkdir -p $O
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; tail -n 3 $O/bench.err
