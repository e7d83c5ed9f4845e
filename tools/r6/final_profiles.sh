#!/bin/bash
# gpurun -- tools/r6/final_profiles.sh <tag>: the round's tracked evidence in one box -- rocprofv3 stats + PMC of the bench (tools/profile_round.sh), the paper's
# loss step kernel by kernel (one node / module by module / MSS / SOT slice), and the counters of the per-row-position kernels.
set -u
cd "${GRAFT_REPO_ROOT:?GPU box only}"
TAG="${1:-r6d}"
tools/profile_round.sh "$TAG" > "gpurun_out/profile_round_$TAG.log" 2>&1; tail -n 5 "gpurun_out/profile_round_$TAG.log"
tools/gpu_call.sh paper_step_profile "step_$TAG" 64 256 > /dev/null 2>&1; tail -n 3 "gpurun_out/step_$TAG/summary.txt"
tools/pmc_passes.sh "pmc_rp_$TAG" "full_kernel" tools/r6/perm_forward_probe.py > /dev/null 2>&1; tail -n 30 "gpurun_out/pmc_rp_$TAG/pmc.txt"
tools/pmc_passes.sh "pmc_rpb_$TAG" "_kernel" tools/r4/perrow_time.py > /dev/null 2>&1; tail -n 5 "gpurun_out/pmc_rpb_$TAG/pmc.txt"
tools/gpu_call.sh kstats "kstats_perrow_$TAG" tools/r4/perrow_time.py | head -n 20
