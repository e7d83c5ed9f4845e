#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ab; mkdir -p $O
python tools/r4/perrow_time.py > $O/perrow.txt 2>&1; cat $O/perrow.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_fuzz.py -x -q -m gpu -k "row or position or sort or fuzz or unsorted" > $O/pytest_rowpos.log 2>&1; tail -n 4 $O/pytest_rowpos.log
