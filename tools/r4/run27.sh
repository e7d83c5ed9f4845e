#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "merge_free_training_form or tie_free_gradient or loss_and_grad_in_one_pass" > $O/pytest_sel.log 2>&1; tail -n 6 $O/pytest_sel.log | cut -c1-300
