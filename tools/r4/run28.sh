#!/bin/bash
# round 4 final (second session): full GPU test suite, the bench line, rocprofv3 --kernel-trace --stats + separate PMC passes of the bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4y; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; tail -n 4 $O/pytest_all.log | cut -c1-300
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -n 3 $O/bench.err
rm -rf gpurun_out/prof_r4b
tools/profile_round.sh r4b > gpurun_out/prof_r4b_run.log 2>&1
tail -n 40 gpurun_out/prof_r4b/summary.txt
for spec in "p1:sot_area_full_kernel<256, 8, 1, false, 0>:" "cutoff:sot_forward_full_kernel<256, 8, 1, 2, true, true, 0>:_paper_mode" "cutoff:sot_backward_full_kernel<256, 8, 2, 2, true, true, 0, false, true, 1>:_training_form"; do
  IFS=: read mode kern suffix <<< "$spec"
  python3 tools/make_traffic_json.py gpurun_out/prof_r4b "B=8192,N=2048,$mode" gpurun_out/prof_r4b/r4b_hbm_traffic$suffix.json "$kern"
done
ls gpurun_out/prof_r4b | head -30
