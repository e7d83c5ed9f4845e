#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
python tools/r4/sort_probe.py > $O/sort.txt 2>&1; tail -n 1 $O/sort.txt
python tools/r4/sort_probe.py 1024 8192 >> $O/sort.txt 2>&1; tail -n 1 $O/sort.txt
python tools/r4/sort_probe.py 8192 300 >> $O/sort.txt 2>&1; tail -n 1 $O/sort.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; tail -n 6 $O/pytest_all.log
