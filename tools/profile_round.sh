#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile_round.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r1}; shift || true
case " $* " in *" --gpus "*) echo "profile_round.sh profiles ONE process: bench.py --gpus N > 1 starts its ranks as grandchildren of rocprofv3, which would then profile the launcher only" >&2; exit 2;; esac
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-extras $*"
# kernel-trace stats of the WHOLE bench command (extras included: per-row positions, segmented sort, position gradients, the paper's loss
# step and MSSLoss, config 5 ...), then of the headline-only command the PMC passes below use (its steady-state averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_extras -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline $* > $OUT/stats_extras.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
pass() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1
}
pass pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass pmc_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_grbm GRBM_GUI_ACTIVE
# the paper's loss step (trainer.py:183-245) at 256 clips: SQ / traffic counters of the fused MSS kernels and the STFT kernels
probe() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/r5/paper_step_probe.py 256 20 full > $OUT/$name.log 2>&1
}
probe pmc_step_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
probe pmc_step_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH
probe pmc_step_fetch FETCH_SIZE
probe pmc_step_write WRITE_SIZE
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
