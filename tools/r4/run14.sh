#!/bin/bash
# round 4 final profile: rocprofv3 --kernel-trace --stats + separate PMC passes of the bench command
cd $GRAFT_REPO_ROOT
tools/profile_round.sh r4 > gpurun_out/prof_r4_run.log 2>&1
tail -n 60 gpurun_out/prof_r4/summary.txt
for spec in "p1:sot_area_full_kernel<256, 8, 1, false, 0>:" "cutoff:sot_forward_full_kernel<256, 8, 1, 2, true, true, 0>:_paper_mode" "cutoff:sot_backward_full_kernel<256, 8, 2, 2, true, true, 0, false, true, 1>:_training_form"; do
  IFS=: read mode kern suffix <<< "$spec"
  python3 tools/make_traffic_json.py gpurun_out/prof_r4 "B=8192,N=2048,$mode" gpurun_out/prof_r4/r4_hbm_traffic$suffix.json "$kern"
done
