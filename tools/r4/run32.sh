#!/bin/bash
# final state of round 4: full GPU test suite + the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ac; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; tail -n 3 $O/pytest_all.log | cut -c1-300
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; tail -n 2 $O/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
