#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ai; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; tail -n 3 $O/pytest_all.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
