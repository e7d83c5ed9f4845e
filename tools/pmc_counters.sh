#!/bin/bash
# SQ counters per row of library variants: tools/pmc_counters.sh <variant>[:flags[:p[:call[:B[:N]]]]] ...  -> gpurun_out/pmc_counters.txt
export TMPDIR=/tmp
OUT=gpurun_out/pmc_counters; rm -rf $OUT; mkdir -p $OUT; : > gpurun_out/pmc_counters.txt
for spec in "$@"; do
  IFS=: read v flags p call B N <<< "$spec"
  for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
              "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F32"; do
    tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${v}_${flags}_${B:-8192}_$tag -- python3 tools/pmc_variant.py $v ${flags:-8} ${p:-1.0} ${call:-fwd} ${B:-8192} ${N:-2048} > $OUT/$v.log 2>&1
  done
  python3 - $OUT "$v" "${flags:-8}" "${B:-8192}" >> gpurun_out/pmc_counters.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + f"/{sys.argv[2]}_{sys.argv[3]}_{sys.argv[4]}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "sot_" in k and "prepare" not in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(sys.argv[2], sys.argv[3], k[:70], {c: round(sum(v) / len(v) / float(sys.argv[4]), 1) for c, v in sorted(d.items())}, "(per row)")
PY
done
cat gpurun_out/pmc_counters.txt
