#!/bin/bash
# rocprofv3 kernel-trace stats + SQ counter passes of the config-5 step (STFT producer + SOT forward/backward).
# Usage: tools/profile_stft.sh <tag>   -> gpurun_out/prof_stft_<tag>/summary.txt
set -u
TAG=${1:-r1}
OUT=gpurun_out/prof_stft_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_train_step.py > $OUT/stats.log 2>&1
pass() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/bench_train_step.py > $OUT/$name.log 2>&1
}
pass pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass pmc_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL
pass pmc_grbm GRBM_GUI_ACTIVE
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
