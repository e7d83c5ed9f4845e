#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4w; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_parity.py::test_merge_free_training_form_p1 > $O/pytest_all.log 2>&1; tail -n 40 $O/pytest_all.log | cut -c1-300
